# the whole GPU suite on the box, log under gpurun_out/<tag>/pytest.txt
cd $GRAFT_REPO_ROOT
tag=${1:-t}; mkdir -p gpurun_out/$tag
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 ${@:2} > gpurun_out/$tag/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/$tag/pytest.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/$tag/pytest.txt | tail -15
