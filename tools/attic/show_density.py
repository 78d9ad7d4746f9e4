import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j['value'], j['ms_per_step'])
for l in j['density_sweep']['levels']: print('   %-36s %.4f ms  %8.0f f/s  mid %3d slow %d' % (l['stream'], l['ms_per_step'], l['frames_per_s'], l['frames_mid_tier'], l['frames_slow_path']))
d = j['density_sweep'].get('deep_schedule')
if d and 'levels' in d:
    print('   deep schedule (%d batches in flight, %d sparse streams):' % (d['batches_in_flight'], d['sparse_streams']))
    for l in d['levels']: print('      %-33s %.4f ms  %8.0f f/s  mid %3d slow %d' % (l['stream'], l['ms_per_step'], l['frames_per_s'], l['frames_mid_tier'], l['frames_slow_path']))
elif d:
    print('   deep schedule:', d)
