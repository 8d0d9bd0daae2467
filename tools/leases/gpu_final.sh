# Final measurements of the round: everything lands under gpurun_out/r2f and is copied into profiles/r02 afterwards.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/kt_bench.json 2> $OUT/kt.err
export PMC_N=10000000
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_write.log 2>&1
export PMC_N=1000000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_mfma.log 2>&1
# config 4 (generic pipeline) and the mixture (config 5 shape on one GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu > $OUT/kt_cfg4_bench.json 2> $OUT/kt_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mix -- python3 $GRAFT_REPO_ROOT/tools/time_mix.py 5000000 256 10 8 12 > $OUT/kt_mix.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --config 4 --steps 5 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python tools/time_mix.py 5000000 256 10 8 16 > $OUT/mix_cfg5.log 2>&1
python tools/time_passes.py 4000000 256 10 > $OUT/passes.log 2>&1
ls -la $OUT
