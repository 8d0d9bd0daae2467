# Round 5: the adopted ring loop of the int8 GEMM: parity of the split pipeline, config 4 with the Gram product on 256- and 128-row tiles, neighbours.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ringf}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generic or split or config4 or k16 or cliff" 2>&1 | tail -3 | tee $OUT/parity.log
run() { python bench.py "$@" --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), 'ms per EM iteration')"; }
for rep in 1 2; do
  echo -n "main config 4: "; run --config 4
  echo -n "main config 4, PPCA_I8GEMM_GRAM_TM=128: "; PPCA_I8GEMM_GRAM_TM=128 run --config 4
  echo -n "ring0 config 4: "; PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_ring0.so run --config 4
done 2>&1 | tee $OUT/ab.log
for s in "256 20" "256 32" "512 10" "300 10"; do set -- $s
  echo -n "main d=$1 k=$2: "; run --n 2000000 --d $1 --k $2
  echo -n "ring0 d=$1 k=$2: "; PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_ring0.so run --n 2000000 --d $1 --k $2
done 2>&1 | tee -a $OUT/ab.log
