# usage (on the GPU box): bash tools/export_shapes.sh <tag> -- the export kernel's shapes (chunks x workgroups), the per-frame chain from C on its own paths
cd $GRAFT_REPO_ROOT
tag=${1:-exps}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
for cfg in "16 16" "8 16" "4 16" "2 16" "8 32" "4 32" "8 8" "4 8" "1 16"; do
  set -- $cfg
  echo "== chunks $1 groups $2"; RMCV_IMAGE_EXPORT=1 RMCV_FRAME_UPLOAD=0 RMCV_IMG_CHUNKS=$1 RMCV_IMG_GROUPS=$2 $out/fc | grep "default_upload\|extract_color on" | head -2 | cut -c1-330
done > $out/export_shapes.txt 2>&1
echo "== runtime copy"; RMCV_IMAGE_EXPORT=0 RMCV_FRAME_UPLOAD=0 $out/fc | grep "default_upload" >> $out/export_shapes.txt
cat $out/export_shapes.txt
