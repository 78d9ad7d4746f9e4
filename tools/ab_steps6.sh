# dev tool: schedule knobs of the pipelined bench after the coalesced k_binary (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS="--steps 20"; echo "default (groups 2, 2 pixel streams)"; run A=1
ARGS="--steps 20"; echo "pixel groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20"; echo "pixel groups 4"; run RMCV_PIXEL_GROUPS=4
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 4"; run RMCV_PIXEL_GROUPS=4
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 3 waves 8"; run RMCV_PIXEL_GROUPS=3 RMCV_SPARSE_WAVES=8
ARGS="--steps 20"; echo "groups 3 waves 8"; run RMCV_PIXEL_GROUPS=3 RMCV_SPARSE_WAVES=8
ARGS="--steps 20 --mode alternate --streams 3"; echo "alternate 3 streams groups 2"; run A=1
ARGS="--steps 20 --mode alternate --streams 3"; echo "alternate 3 streams groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20 --mode alternate --streams 2"; echo "alternate 2 streams groups 3"; run RMCV_PIXEL_GROUPS=3
done
