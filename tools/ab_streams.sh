for s in 1 2 3 4; do
  timeout -k 10 120 python bench.py --steps 20 --cpu-frames 0 --no-extras --streams $s > gpurun_out/bp.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/bp.log').read().strip().splitlines()[-1]); print('streams',$s, j['value'], j['ms_per_step'], j['stage_ms'])"
done
