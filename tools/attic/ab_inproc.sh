#!/bin/bash
# dev tool: in-process alternating A/B of a context option in bench.py's own loop (RMCV_BENCH_AB, see bench.py):  bash tools/ab_inproc.sh <opt:a:b[:pairs]> [bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
spec=$1; shift
RMCV_BENCH_AB=$spec timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras "$@" > gpurun_out/abr3/ab.log 2>gpurun_out/abr3/ab.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/ab.log').read().strip().splitlines()[-1]); a=j['ab']
print('$spec $*: A=%d mean %.4f median %.4f | B=%d mean %.4f median %.4f | B/A %.4f' % (a['a']['value'], a['a']['mean_ms'], a['a']['median_ms'], a['b']['value'], a['b']['mean_ms'], a['b']['median_ms'], a['b_over_a']))
print('   A', a['a']['each']); print('   B', a['b']['each'])" || tail -3 gpurun_out/abr3/ab.err
