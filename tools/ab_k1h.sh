cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do for g in 2 3 4; do
echo "== per-lane groups $g"; RMCV_K1_COAL=0 python tools/k1_bench.py $g
echo "== coalesced groups $g"; python tools/k1_bench.py $g
done; done
echo "== per-lane 1920 groups 2"; RMCV_K1_COAL=0 python tools/k1_bench.py 2 2 1920 1200
echo "== coalesced 1920 groups 2"; python tools/k1_bench.py 2 2 1920 1200
echo "== coalesced morph 1"; python tools/k1_bench.py 2 1
} > gpurun_out/abk1/out_h.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_h.txt
