# usage (on the GPU box): bash tools/chain_beside.sh <tag>  -- the per-frame chain from C beside a process that holds what bench.py holds
cd $GRAFT_REPO_ROOT
tag=${1:-beside}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc_dev -Lrmcv_amd/lib/dev -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib/dev || exit 1
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
{
python tools/hold_gpu.py 40 3 & hp=$!
sleep 25
for i in 1 2; do
  echo "== BESIDE 3 full-size pipelines: image export (RMCV_IMAGE_EXPORT=1) + polling waits, run $i"; RMCV_IMAGE_EXPORT=1 $out/fc || exit 1
  echo "== BESIDE: runtime's pageable copy (the default), polling waits, run $i"; $out/fc || exit 1
  echo "== BESIDE: runtime's copy AND runtime's waits (round 4's chain; dev build), run $i"; RMCV_WAIT_RUNTIME=1 $out/fc_dev || exit 1
done
wait $hp
} > $out/chain_beside.txt 2>&1
grep "^==\|^runtime_pageable\|^registered\|hold_gpu\|extract_color on" $out/chain_beside.txt | cut -c1-330
