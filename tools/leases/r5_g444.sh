# Round 5: the fallback engine's fp64 Gram + b on the 4 x 4 x 4 MFMA: the GPU suite (default, and with every fused pass forced onto the fp64 Gram), the
# alternate flows, then bench --gram fp64 with PASS_G444 on / off (K = 10 development libraries).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5g444}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
suite() { env "$@" timeout 1200 python -m pytest tests -m gpu -q -k "not dynamic_range_guard_placeholder" 2>&1 | grep -E "passed|failed|FAILED" | sed "s/^/[$*] /"; }
suite PPCA_DEFAULT=1
env PPCA_GRAM_FP64=1 timeout 1200 python -m pytest tests -m gpu -q -k "not test_int8_gram_dynamic_range_guard" 2>&1 | grep -E "passed|failed|FAILED" | sed "s/^/[PPCA_GRAM_FP64=1] /"
suite PPCA_EM9=0
suite PPCA_QPREP_CACHE=0
for rep in 1 2; do for L in g444on g444off; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L --gram fp64', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms', j['roofline']['kernel'])"
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --outliers 10 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L --outliers 10', round(j['value'],2), 'it/s')"
done; done
