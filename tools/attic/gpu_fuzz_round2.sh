# longer fuzz runs with seeds the test-suite does not use (5 processes in parallel: the oracle side is single-threaded)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/fuzz
timeout -k 10 800 python tools/fuzz_path.py 70 101 0 > gpurun_out/fuzz/path0.txt 2>&1 &
timeout -k 10 800 python tools/fuzz_path.py 50 102 1 > gpurun_out/fuzz/path1.txt 2>&1 &
timeout -k 10 800 python tools/fuzz_contours.py 2500 103 > gpurun_out/fuzz/contours.txt 2>&1 &
timeout -k 10 800 python tools/fuzz_legacy.py 500 104 > gpurun_out/fuzz/legacy.txt 2>&1 &
timeout -k 10 800 python tools/fuzz_hull.py 600 105 > gpurun_out/fuzz/hull.txt 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 30; echo "still fuzzing: $(jobs -r | wc -l) running"; done
wait
for f in gpurun_out/fuzz/*.txt; do echo "== $f"; grep -v amdgpu.ids $f | tail -2; done
