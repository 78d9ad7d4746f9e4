# dev tool: same-box A/B of environment settings (usage: bash tools/ab_env.sh "ENV1=a ENV2=b,ENV3=c ..." "<bench args>"; a comma joins
# variables of one setting)
run() { env "$@" timeout -k 10 120 python bench.py --steps 60 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abe.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abe.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'].get('fused_sparse'), j['roofline']['avg_launch_ms'])"; }
ARGS="${2:---streams 3}"
for rep in 1 2 3; do for E in $1; do echo "== $ARGS $E"; run ${E//,/ }; done; done
