# dev tool: same-box A/B of k_binary builds/knobs with tools/k1_bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 4; do
echo "== base groups $g"; python tools/k1_bench.py $g
echo "== LOADV=1 (coalesced through LDS) groups $g"; RMCV_K1_LOADV=1 python tools/k1_bench.py $g
echo "== nt loads, strided, groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so python tools/k1_bench.py $g
echo "== nt loads + LOADV=1 groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py $g
done
done
echo "== nt loads + LOADV=1 groups 1"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 1
echo "== nt loads + LOADV=1 groups 3"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 3
echo "== nt loads + LOADV=1 groups 6"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 6
} > gpurun_out/abk1/out.txt 2>&1
grep -E "^==|k_binary|checksum" gpurun_out/abk1/out.txt
