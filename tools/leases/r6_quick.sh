R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "wp_digits or generic or config4 or split" 2>&1 | tail -3
timeout 300 python bench.py --config 4 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg4', j['ms_per_step'])"
