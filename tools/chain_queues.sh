cd $GRAFT_REPO_ROOT; out=gpurun_out/r5e; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
for q in 4 12 4 12; do
  echo "== GPU_MAX_HW_QUEUES=$q, runtime copy"; GPU_MAX_HW_QUEUES=$q $out/fc
  echo "== GPU_MAX_HW_QUEUES=$q, image export"; GPU_MAX_HW_QUEUES=$q RMCV_IMAGE_EXPORT=1 $out/fc
done > $out/q.txt 2>&1
grep "^==\|^runtime_pageable\|^registered\|extract_color on" $out/q.txt | cut -c1-330
