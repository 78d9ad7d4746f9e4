# dev tool: k_binary efficiency vs frames per launch (fixed ramp/tail cost?)
for n in 64 128 256 512 1024; do timeout -k 10 200 python bench.py --frames $n --steps 20 --cpu-frames 0 --no-extras --streams 1 > gpurun_out/abf.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abf.log').read().strip().splitlines()[-1]); r=j['roofline']; print($n, 'k1 ms', r['avg_launch_ms'], 'GB/s', r['achieved'], 'per-frame us', round(1e3*r['avg_launch_ms']/$n,4), 'fps', j['value'])"; done
