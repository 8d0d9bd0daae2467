# Round-3 baseline: GPU tests, headline, cliff shapes around the fused kernel, fp64-Gram fallback engine, phase table.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
PPCA_GRAM_FP64=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_fp64gram.json 2> $OUT/bench_n10m_fp64gram.err
for s in "200 16" "256 11" "256 16" "300 10" "512 10" "256 10" "200 10" "128 8"; do
  set -- $s
  timeout 600 python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
done
python tools/devbuild.py --timing > $OUT/devbuild.log 2>&1
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so timeout 300 python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
tail -4 $OUT/timing.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3a/*.json')):
    try:
        j = json.load(open(f)); r = j['roofline']
        print(os.path.basename(f), round(j['value'], 2), 'it/s', round(j['ms_per_step'], 3), 'ms frac', round(r['frac'], 3), r['kernel'][:50])
    except Exception as e:
        print(f, 'ERR', e)
PY
