mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q -k "rccl or edge or golden or two_ranks or sharded" > gpurun_out/r2a/test_sel.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/test_sel.log
grep -E "passed|failed|rc=|Error" gpurun_out/r2a/test_sel.log | tail -5
