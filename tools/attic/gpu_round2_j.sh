# final-state record of round 2 (coalesced pixel kernel): the driver's command twice, C5, on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h/bench1.json 2> gpurun_out/r3h/bench1.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h/bench2.json 2> gpurun_out/r3h/bench2.err
python bench.py --workload c5 --steps 20 --warmup 5 > gpurun_out/r3h/bench_c5.json 2> gpurun_out/r3h/bench_c5.err
python tools/print_bench.py gpurun_out/r3h/bench1.json gpurun_out/r3h/bench2.json gpurun_out/r3h/bench_c5.json
python tools/sparse_time.py
