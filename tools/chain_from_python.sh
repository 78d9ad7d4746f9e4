# usage (on the GPU box): bash tools/chain_from_python.sh <tag> -- the per-frame chain from C, started from bash and from Python processes
# that have imported nothing / numpy / torch (does the child inherit something that slows the runtime's copies down?)
cd $GRAFT_REPO_ROOT
tag=${1:-frompy}; out=gpurun_out/$tag; mkdir -p $out
gcc -O2 -Iinclude tools/frame_chain.c -o $out/fc -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib || exit 1
{
echo "== from bash"; $out/fc
echo "== from python (nothing imported)"; python -c "import subprocess,sys; subprocess.run([sys.argv[1]])" $out/fc
echo "== from python (numpy imported)"; python -c "import numpy, subprocess,sys; subprocess.run([sys.argv[1]])" $out/fc
echo "== from python (torch imported)"; python -c "import torch, subprocess,sys; subprocess.run([sys.argv[1]])" $out/fc
echo "== from python (torch imported, capture_output)"; python -c "import torch, subprocess,sys; print(subprocess.run([sys.argv[1]], capture_output=True, text=True).stdout)" $out/fc
echo "== from python (bench.py's imports and env)"; python -c "
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')
sys.path.insert(0, '.')
import bench
print(bench.c_host_chain(1280, 1024))"
echo "== from bash again"; $out/fc
} > $out/frompy.txt 2>&1
grep "^==\|^cpu\|^runtime_pageable\|^registered\|^{" $out/frompy.txt | cut -c1-400
