R=$GRAFT_REPO_ROOT
cd $R
for r in 1 2; do timeout 120 python tools/time_passes.py 4000000 256 10 2>&1 | grep -E "llk|smooth|extrapolate" | head -3; done
timeout 120 python tools/time_passes.py 4000000 200 10 2>&1 | grep -E "llk|smooth|extrapolate" | head -3
timeout 900 python -m pytest tests -m gpu -x -q -k "output_rows or output_passes or extrapolate or smooth or eight_wave" 2>&1 | tail -3
