# Round 5: what the int8 GEMM's K loop waits for -- address translation, L1, LDS and issue counters of its two launches at config 4's shape.
#   bash tools/leases/r5_pmc_i8.sh <out-dir> <lib-suffix>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5pmci8}
mkdir -p $OUT
LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip.so
[ -n "$2" ] && [ "$2" != main ] && LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip_$2.so
export PPCA_HIP_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
export PMC_N=175000 PMC_STEPS=1
i=0
while read -r C; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/p$i.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_agg.py $OUT/p$i --filter=i8gemm
done <<'LIST' 2>&1 | tee $OUT/summary.log
TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_PENDING_STALL_CYCLES
TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES
SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES
SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS
GRBM_GUI_ACTIVE TCP_GATE_EN1 TCP_GATE_EN2 TCP_TAGRAM0_REQ
LIST
