# the whole GPU suite on the box, log under gpurun_out/<tag>/pytest.txt
cd $GRAFT_REPO_ROOT
tag=${1:-t}; mkdir -p gpurun_out/$tag
# --capture=sys: Python-level capture only -- what the HIP runtime itself writes to fd 2 (a queue error, a memory fault, the text of an
# abort) goes straight into the log instead of dying with pytest's capture buffer (rounds 2 and 4 each lost one that way)
timeout -k 10 1000 python -m pytest tests -m gpu -q --capture=sys --durations=15 ${@:2} > gpurun_out/$tag/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/$tag/pytest.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/$tag/pytest.txt | tail -15
