# usage (on the GPU box): bash tools/chain_ab.sh <tag>
# The per-frame drop-in chain from a C host (tools/frame_chain.c), process after process on one box: where rmcv_extract_color's host
# time goes (rmcv_ctx_frame_timing), with the caller's buffers backed in three ways (FC_ALLOC), and -- dev build, RMCV_WAIT_RUNTIME=1
# (make EXTRA=-DRMCV_DEV_KNOBS OUT=../lib/dev/librmcv_hip.so OBJDIR=../lib/dev/obj) -- the HIP runtime's own waits instead of polling.
cd $GRAFT_REPO_ROOT
tag=${1:-chain}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc_dev -Lrmcv_amd/lib/dev -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib/dev || exit 1
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
{ cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo "<- numa nodes of the cards"; lscpu | grep -i "numa\|socket" ; cat /sys/kernel/mm/transparent_hugepage/enabled; } > $out/chain_ab.txt 2>&1
for i in 1 2 3; do
  for a in malloc huge nohuge; do
    echo "== polling waits (product build), FC_ALLOC=$a, run $i"; FC_ALLOC=$a $out/fc || exit 1
  done
  echo "== runtime waits (hipStreamSynchronize, dev build), run $i"; RMCV_WAIT_RUNTIME=1 $out/fc_dev || exit 1
done >> $out/chain_ab.txt 2>&1
cat $out/chain_ab.txt
