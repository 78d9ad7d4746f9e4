# usage (on the GPU box): bash tools/chain_after_exit.sh <tag> -- the per-frame chain from C right behind the exit of a process that held
# tens of GB of GPU memory (tools/hold_gpu.py 0 3), every few seconds for a minute: the runtime's pageable copies (RMCV_FRAME_UPLOAD=0
# RMCV_IMAGE_EXPORT=0: round 4's chain) against the defaults (the library leaves them while they are slow)
cd $GRAFT_REPO_ROOT
tag=${1:-afterexit}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
{
python tools/hold_gpu.py 0 3
t0=$(date +%s)
for i in 1 2 3 4 5 6; do
  echo "== $(( $(date +%s) - t0 )) s after the exit: runtime's copies only"; RMCV_FRAME_UPLOAD=0 RMCV_IMAGE_EXPORT=0 $out/fc
  echo "== $(( $(date +%s) - t0 )) s after the exit: defaults (auto)"; $out/fc
  sleep 4
done
} > $out/after_exit.txt 2>&1
grep "^==\|^default_upload\|^registered\|chains on" $out/after_exit.txt | cut -c1-420
