for m in pipeline alternate; do for s in 2 3 4; do
  timeout -k 10 120 python bench.py --steps 30 --cpu-frames 0 --no-extras --streams $s --mode $m > gpurun_out/bp.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/bp.log').read().strip().splitlines()[-1]); print('$m','streams',$s, j['value'], j['ms_per_step'], j['config']['armours_gathered'])"
done; done
