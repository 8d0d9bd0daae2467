# Round 6, second lease: em9's gathered / weighted passes with their row list and weights prefetched as vectors; config 5 again, its
# kernel statistics and PMC traffic; the multi-component tests.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6b
mkdir -p $OUT $R/profiles/r06
cd $R
COMMIT=$(cat $R/tools/commit_stamp.txt 2>/dev/null)
timeout 900 python -m pytest tests -m gpu -x -q -k "multi_component or mixture_against_oracle or different_state_sizes or gathered or steady_state and eight_wave" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python tools/time_weighted.py 10000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/weighted_n10m.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg5 -- python3 $R/bench.py --config 5 --no-cpu > $OUT/kt_cfg5_bench.json 2> $OUT/kt_cfg5.err
export PMC_N=1000000 PMC_STEPS=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc5_fetch -- python3 $R/tools/pmc_mix.py > $OUT/pmc5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc5_write -- python3 $R/tools/pmc_mix.py > $OUT/pmc5_write.log 2>&1
python3 $R/tools/make_traffic_all.py $OUT/pmc5_fetch $OUT/pmc5_write 1000000 256 2 $OUT/traffic_cfg5.json "$COMMIT" "one steady-state mixture EM iteration, K = 8, d = 256, k = 10, 30 % masked (BASELINE config 5's shape at N = 1 M)" > $OUT/traffic_cfg5.log 2>&1
cd $R
python - <<PY
import json
for f in ("bench_cfg5", "bench_n10m"):
    try:
        j = json.load(open("$OUT/%s.json" % f))
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
cat $OUT/weighted_n10m.log
grep -E "hbm_bytes_per_sample|llk8|em9|posteriors|select|reduce" $OUT/traffic_cfg5.json
f=$(ls -t $OUT/kt_cfg5/*/*kernel_stats.csv | head -1); head -12 $f
