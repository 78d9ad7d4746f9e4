cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do for g in 2 3 4 6; do
echo "== new groups $g"; python tools/k1_bench.py $g
done; done
echo "== new groups 4 1920x1200"; python tools/k1_bench.py 4 2 1920 1200
echo "== new groups 2 1920x1200"; python tools/k1_bench.py 2 2 1920 1200
echo "== new groups 4 morph 1"; python tools/k1_bench.py 4 1
echo "== new groups 4 morph 0"; python tools/k1_bench.py 4 0
} > gpurun_out/abk1/out_e.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_e.txt
