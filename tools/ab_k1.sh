timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binary or full_path or geometry" 2>&1 | tail -1
RMCV_K1_LOADV=0 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binary or full_path or geometry" 2>&1 | tail -1
for lv in 0 1; do for v in librmcv_hip var_neither; do
  RMCV_K1_LOADV=$lv RMCV_LIB_PATH=$PWD/rmcv_amd/lib/$v.so timeout -k 10 100 python bench.py --steps 10 --cpu-frames 0 --streams 1 > gpurun_out/ab.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]); print('LOADV$lv $v', j['stage_ms']['binary'], j['roofline']['achieved'], j['c2_binary_only'])"
done; done
