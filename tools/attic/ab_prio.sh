# dev tool: wave issue priorities (s_setprio) of the sparse and the pixel kernel against each other, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2 3; do
echo "tree: sparse 3, pixel 0"; run A=1
echo "sparse 0, pixel 0"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_sp0.so
echo "sparse 0, pixel 2"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_sp0k2.so
echo "sparse 3, pixel 2"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_k2.so
done
