# Round 5: the streaming int8 GEMM (one workgroup per CU walks its tiles): parity of the split pipeline, then config 4 with PPCA_I8GEMM_STREAM = 1 / 0.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5stream}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generic or split or config4 or k16 or cliff" 2>&1 | tail -3 | tee $OUT/parity.log
run() { timeout 300 python bench.py "$@" --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), 'ms per EM iteration')"; }
for rep in 1 2; do
  echo -n "stream config 4: "; run --config 4
  echo -n "PPCA_I8GEMM_STREAM=0 config 4: "; PPCA_I8GEMM_STREAM=0 run --config 4
done 2>&1 | tee $OUT/ab.log
for s in "256 20" "256 32" "256 48" "512 10"; do set -- $s
  echo -n "stream d=$1 k=$2: "; run --n 2000000 --d $1 --k $2
  echo -n "PPCA_I8GEMM_STREAM=0 d=$1 k=$2: "; PPCA_I8GEMM_STREAM=0 run --n 2000000 --d $1 --k $2
done 2>&1 | tee -a $OUT/ab.log
