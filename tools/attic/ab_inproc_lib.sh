#!/bin/bash
# dev tool: this build against a variant build, alternating regions inside ONE process, in BOTH roles (the second set of contexts of a
# process runs 1-2 % faster than the first whatever the build: tools/ab_inproc.sh lib:<a copy of the same .so> reads 0.98-0.99):
#   bash tools/ab_inproc_lib.sh rmcv_amd/lib/var_<name>.so [bench args]     ->  variant / default, both ways, and their geometric mean
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
var=$1; shift
one() { env "$@" RMCV_BENCH_AB=lib:$OTHER timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $BARGS > gpurun_out/abr3/abl.log 2>gpurun_out/abr3/abl.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/abl.log').read().strip().splitlines()[-1]); print(j['ab']['b_over_a'])" || { tail -3 gpurun_out/abr3/abl.err; echo nan; }; }
BARGS="$*"
OTHER=$var; r1=$(one A=1)                          # primary = default build, second = variant: r1 = variant / default
OTHER=rmcv_amd/lib/var_same.so; r2=$(one RMCV_LIB_PATH=$var)   # primary = variant, second = (a copy of) the default build: r2 = default / variant
python3 -c "
import math
r1, r2 = float('$r1'), float('$r2')
print('$var $BARGS: variant/default %.4f (variant as the second set), %.4f (as the first) -> %.4f' % (r1, 1 / r2, math.sqrt(r1 / r2)))"
