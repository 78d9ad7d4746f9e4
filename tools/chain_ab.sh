# usage (on the GPU box): bash tools/chain_ab.sh <tag>
# The per-frame drop-in chain from a C host (tools/frame_chain.c), process after process on one box: where rmcv_extract_color's host
# time goes (rmcv_ctx_frame_timing) -- alone on the GPU, and beside another process that holds a pipeline's worth of queues on it
# (tools/hold_gpu.py), with the library's own image export (default) and with the runtime's pageable copy (the default); and --
# dev build, RMCV_WAIT_RUNTIME=1 (make EXTRA=-DRMCV_DEV_KNOBS OUT=../lib/dev/librmcv_hip.so OBJDIR=../lib/dev/obj) -- with the HIP
# runtime's own waits instead of the library's polling.
cd $GRAFT_REPO_ROOT
tag=${1:-chain}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc_dev -Lrmcv_amd/lib/dev -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib/dev || exit 1
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
{
for i in 1 2; do
  echo "== ALONE: library's image export (RMCV_IMAGE_EXPORT=1) + polling waits, run $i"; RMCV_IMAGE_EXPORT=1 $out/fc || exit 1
  echo "== ALONE: runtime's pageable copy (the default), run $i"; $out/fc || exit 1
  echo "== ALONE: runtime's copy AND runtime's waits (round 4's chain; dev build), run $i"; RMCV_WAIT_RUNTIME=1 $out/fc_dev || exit 1
done
python tools/hold_gpu.py 45 & hp=$!
sleep 12
for i in 1 2; do
  echo "== BESIDE an idle process with a pipeline: library's image export (RMCV_IMAGE_EXPORT=1) + polling waits, run $i"; RMCV_IMAGE_EXPORT=1 $out/fc || exit 1
  echo "== BESIDE: runtime's pageable copy (the default), run $i"; $out/fc || exit 1
  echo "== BESIDE: runtime's copy AND runtime's waits (round 4's chain; dev build), run $i"; RMCV_WAIT_RUNTIME=1 $out/fc_dev || exit 1
done
wait $hp
} > $out/chain_ab.txt 2>&1
grep "^==\|^runtime_pageable\|^registered" $out/chain_ab.txt
