#!/bin/bash
# dev tool: the density sweep with and without RMCV_OPT_DENSE_DEFER (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
for rep in 1 2; do for d in 0 1; do
echo "== defer $d"
RMCV_DENSE_DEFER=$d RMCV_BENCH_SWEEP_LEVELS=plain,dense1,dense2,dense3,dense4,one timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras --density-sweep > gpurun_out/abr3/df.log 2>gpurun_out/abr3/df.err && python tools/show_density.py gpurun_out/abr3/df.log
done; done 2>&1 | tee gpurun_out/abr3/defer_ab.txt
