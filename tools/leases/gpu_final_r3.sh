# Final measurements of round 3: everything lands under gpurun_out/r3f and is copied into profiles/r03 by
# tools/leases/collect_profiles_r3.sh afterwards.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_fp64gram.json 2> $OUT/bench_n10m_fp64gram.err
PPCA_EM8=0 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_em4.json 2> $OUT/bench_n10m_em4.err
python bench.py --config 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python bench.py --config 4 --steps 5 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
for s in "200 16" "256 11" "256 13" "256 16" "300 10" "512 10" "256 10" "200 10" "256 20"; do
  set -- $s
  python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
done
python tools/time_passes.py 4000000 256 10 > $OUT/passes.log 2>&1
python tools/time_transfer.py > $OUT/transfer.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/kt_bench.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fp64gram -- python3 $GRAFT_REPO_ROOT/bench.py --gram fp64 --steps 6 --warmup 2 --no-cpu > $OUT/kt_fp64gram_bench.json 2> $OUT/kt_fp64gram.err
export PMC_N=10000000
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_write.log 2>&1
export PMC_N=1000000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_inst -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_inst.log 2>&1
# config 4 (generic pipeline): kernel stats, counters of the int8 GEMM / solver / skinny kernels, HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu > $OUT/kt_cfg4_bench.json 2> $OUT/kt_cfg4.err
export PMC_N=175000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_cfg4 -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/pmc_cfg4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cfg4_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/pmc_cfg4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cfg4_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/pmc_cfg4_write.log 2>&1
# the mixture (config 5 on one GPU) and the shape one step outside the fused kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu > $OUT/kt_cfg5_bench.json 2> $OUT/kt_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d256_k11 -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d 256 --k 11 --steps 3 --warmup 1 --no-cpu > $OUT/kt_d256_k11_bench.json 2> $OUT/kt_d256_k11.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d200_k16 -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d 200 --k 16 --steps 3 --warmup 1 --no-cpu > $OUT/kt_d200_k16_bench.json 2> $OUT/kt_d200_k16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d512_k10 -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d 512 --k 10 --steps 3 --warmup 1 --no-cpu > $OUT/kt_d512_k10_bench.json 2> $OUT/kt_d512_k10.err
cd $GRAFT_REPO_ROOT
python tools/devbuild.py --timing --name=devt > $OUT/devbuild.log 2>&1
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_devt.so python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_devt.so python tools/time_passes.py 4000000 256 10 > $OUT/timing_passes.log 2>&1
python tools/devbuild16.py --timing > $OUT/devbuild16.log 2>&1
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev16.so python bench.py --n 2000000 --d 200 --k 16 --steps 3 --warmup 1 --no-cpu > $OUT/timing16.json 2> $OUT/timing16.err
python tools/devbuild.py --name=devllk > /dev/null 2>&1; python tools/devbuild.py -DLLK2_DIAG_RESIDENT --name=devllkres > /dev/null 2>&1
bash tools/leases/gpu_llk_resident.sh > $OUT/llk2_resident.log 2>&1
python tools/em16_check.py > $OUT/em16_check.log 2>&1
ls -la $OUT
