OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3p4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/kt_cfg4_bench.json 2> $OUT/kt_cfg4.err
f=$(ls -t $(find $OUT/kt_cfg4 -name "*kernel_stats.csv") | head -1); head -16 $f | cut -d, -f1-4 | cut -c1-130
export PMC_N=175000
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/p1.log 2>&1
tail -2 $OUT/p1.log
