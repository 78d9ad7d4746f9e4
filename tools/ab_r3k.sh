# dev tool (round 3): k_binary alone with the plane stores written through (sc1, the build) against plain (variant pl0), then the bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
{
for rep in 1 2; do
for g in 2 3; do
echo "== tree (plane stores sc1) groups $g"; python tools/k1_bench.py $g
echo "== pl0 (plain plane stores) groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pl0.so python tools/k1_bench.py $g
done
done
} > gpurun_out/abr3/k1.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abr3/k1.txt
