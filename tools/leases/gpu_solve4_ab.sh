#!/bin/bash
# A/B of the batched blocked solver (ppca_solve4.hip) against the one-sample-per-wave form, plus the parity tests it serves
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s4; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generic_pipeline or config4 or cfg4 or mixture" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for rep in 1 2; do
for v in 1 0; do
  for s in "256 20" "256 32" "256 48"; do
    set -- $s
    PPCA_SOLVE4=$v timeout 300 python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $O/b_s4${v}_d$1_k$2_$rep.json 2>> $O/err.log
  done
  PPCA_SOLVE4=$v timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $O/b_s4${v}_cfg4_$rep.json 2>> $O/err.log
done
done
grep -o '"ms_per_step": [0-9.]*' $O/b_*.json > $O/summary.txt
