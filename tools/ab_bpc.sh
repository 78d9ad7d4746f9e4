timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binary or full_path or geometry or full_size" 2>&1 | tail -2
for bpc in 8 6 5 4; do for s in 1 2 3; do
  RMCV_K1_BPC=$bpc timeout -k 10 120 python bench.py --steps 20 --cpu-frames 0 --no-extras --streams $s > gpurun_out/bp.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/bp.log').read().strip().splitlines()[-1]); print('bpc',$bpc,'streams',$s, j['value'], j['ms_per_step'], j['stage_ms']['binary'])"
done; done
