# Round 5: bench --gram fp64 / --outliers 10 / llk through the fp64 engine, PASS_G444 on (the tree's library) / off (libppca_hip_g444off.so, a full build).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5g444ab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in main g444off; do
  LIB=$PWD/ppca_rs_amd/libppca_hip.so; [ $L != main ] && LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so
  PPCA_HIP_LIB=$LIB timeout 300 python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L --gram fp64', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms', j['roofline']['kernel'])"
  PPCA_HIP_LIB=$LIB timeout 300 python bench.py --outliers 10 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L --outliers 10', round(j['value'],2), 'it/s', j['roofline'].get('fallback',{}).get('second_stage_ms_per_step'))"
done; done 2>&1 | tee $OUT/ab.log
