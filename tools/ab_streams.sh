# dev tool: stream count / priority / HW-queue / variant-library experiments for bench.py's double-buffered schedule
run() { echo "== NS=$NS $*"; env "$@" timeout -k 10 120 python bench.py --steps 80 --cpu-frames 0 --no-extras --streams $NS > gpurun_out/abs.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abs.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'])"; }
for rep in 1 2; do
NS=3 run A=1
NS=3 run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_lds.so
NS=3 run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_prio.so
NS=2 run A=1
NS=3 run RMCV_K1_BPC=3
NS=3 run RMCV_K1_BPC=1
done
