# dev tool: same-box A/B of builds of librmcv_hip.so (usage: bash tools/ab_lib.sh "<variant.so> ..." "<bench args>;<bench args>")
run() { env "$@" timeout -k 10 120 python bench.py --steps 60 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abl.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abl.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'].get('fused_sparse'))"; }
IFS=';' read -ra SETS <<< "${2:---streams 3;--streams 1}"
for ARGS in "${SETS[@]}"; do for rep in 1 2; do
echo "== $ARGS base"; run A=1
for V in $1; do echo "== $ARGS $V"; run RMCV_LIB_PATH=$PWD/$V; done
done; done
