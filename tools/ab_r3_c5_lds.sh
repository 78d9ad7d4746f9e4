cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2 3; do
echo "tree (row tables for 2048 rows: 88.0 KB)"; run A=1
echo "row tables for 1280 rows (81.8 KB: fits beside four pixel workgroups)"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_maxh1280.so
done
