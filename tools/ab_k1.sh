for v in librmcv_hip var_sr32 var_sr128 var_ntl var_nts var_ntls; do
  RMCV_LIB_PATH=$PWD/rmcv_amd/lib/$v.so RMCV_K1_LOADV=0 python bench.py --steps 10 --cpu-frames 0 > gpurun_out/ab_$v.log 2>&1
  python3 -c "
import json
j=json.loads(open('gpurun_out/ab_$v.log').read().strip().splitlines()[-1]); print('$v', j['value'], j['stage_ms']['binary'], j['roofline']['achieved'], j['c2_binary_only'])"
done
