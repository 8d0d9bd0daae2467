R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6fz
mkdir -p $OUT
cd $R
(for seed in 101 202 303; do PPCA_GEN_WPRED_STATS=1 timeout 900 python tools/fuzz_generic.py $seed 60 2>&1 | grep -v "of 0 (col" | tail -4; done) > $OUT/fuzz_generic.log 2>&1
tail -3 $OUT/fuzz_generic.log
timeout 600 python tools/soak_generic.py > $OUT/soak_generic.log 2>&1; tail -3 $OUT/soak_generic.log
