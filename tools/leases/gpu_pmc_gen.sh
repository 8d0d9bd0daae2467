OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcgen
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PMC_N=175000
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/p3.log 2>&1
tail -2 $OUT/p1.log $OUT/p2.log $OUT/p3.log
