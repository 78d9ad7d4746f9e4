for bpc in 3 4 5 6; do for s in 2 3; do
  RMCV_K1_BPC=$bpc timeout -k 10 120 python bench.py --steps 30 --cpu-frames 0 --no-extras --streams $s > gpurun_out/bp.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/bp.log').read().strip().splitlines()[-1]); print('bpc',$bpc,'streams',$s, j['value'], j['ms_per_step'])"
done; done
