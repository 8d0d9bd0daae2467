# Final measurements of round 6: everything lands under gpurun_out/r6f2 (copied into profiles/r06 by tools/leases/collect_profiles_r6.sh).
# The PMC passes and the MFMA loops run FIRST and write profiles/r06/{traffic*.json, mfma_peak.json} on the box, so that the bench lines
# that cite them are produced against the files they cite (same commit: tools/commit_stamp.txt).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6f2
mkdir -p $OUT $R/profiles/r06
cd /tmp && export TMPDIR=/tmp
COMMIT=$(cat $R/tools/commit_stamp.txt 2>/dev/null)
[ -x $R/tools/mfma_peak/mfma_peak ] || hipcc --offload-arch=gfx950 -O3 -w $R/tools/mfma_peak/mfma_peak.hip -o $R/tools/mfma_peak/mfma_peak
$R/tools/mfma_peak/mfma_peak "$COMMIT" > $OUT/mfma_peak.txt 2>&1
tail -1 $OUT/mfma_peak.txt > $R/profiles/r06/mfma_peak.json; cp $R/profiles/r06/mfma_peak.json $OUT/mfma_peak.json
# ---- headline kernel: traffic (N = 10 M) and instruction / MFMA counters (N = 1 M)
export PMC_N=10000000
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/pmc_run.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/pmc_run.py > $OUT/pmc_write.log 2>&1
python3 $R/tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write 10000000 $R/profiles/r06/traffic.json "$COMMIT" > $OUT/traffic.log 2>&1
cp $R/profiles/r06/traffic.json $OUT/traffic.json
export PMC_N=1000000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/tools/pmc_run.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_inst -- python3 $R/tools/pmc_run.py > $OUT/pmc_inst.log 2>&1
# ---- config 5: one steady-state mixture iteration, every kernel (N = 1 M rows of config 5's shape)
export PMC_N=1000000 PMC_STEPS=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc5_fetch -- python3 $R/tools/pmc_mix.py > $OUT/pmc5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc5_write -- python3 $R/tools/pmc_mix.py > $OUT/pmc5_write.log 2>&1
python3 $R/tools/make_traffic_all.py $OUT/pmc5_fetch $OUT/pmc5_write 1000000 256 2 $R/profiles/r06/traffic_cfg5.json "$COMMIT" "one steady-state mixture EM iteration, K = 8, d = 256, k = 10, 30 % masked (BASELINE config 5's shape at N = 1 M)" > $OUT/traffic_cfg5.log 2>&1
cp $R/profiles/r06/traffic_cfg5.json $OUT/traffic_cfg5.json
# ---- config 4: every kernel of the split pipeline (175 000 rows of config 4's shape: two chunks)
export PMC_N=175000 PMC_STEPS=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc4_fetch -- python3 $R/tools/pmc_generic.py > $OUT/pmc4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4_write -- python3 $R/tools/pmc_generic.py > $OUT/pmc4_write.log 2>&1
python3 $R/tools/make_traffic_all.py $OUT/pmc4_fetch $OUT/pmc4_write 175000 1024 1 $R/profiles/r06/traffic_cfg4.json "$COMMIT" "one EM pass of the split pipeline, d = 1024, k = 64, 50 % masked (BASELINE config 4's shape at N = 175 000)" > $OUT/traffic_cfg4.log 2>&1
cp $R/profiles/r06/traffic_cfg4.json $OUT/traffic_cfg4.json
cd $R
# ---- bench lines
python bench.py > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 1 > $OUT/bench_n10m_outliers1.json 2> $OUT/bench_n10m_outliers1.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 10 > $OUT/bench_n10m_outliers10.json 2> $OUT/bench_n10m_outliers10.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 1000 > $OUT/bench_n10m_outliers1000.json 2> $OUT/bench_n10m_outliers1000.err
PPCA_HEAVY_ROWS=0 python bench.py --steps 10 --warmup 2 --no-cpu --outliers 10 > $OUT/bench_n10m_outliers10_heavy0.json 2> $OUT/bench_n10m_outliers10_heavy0.err
python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_fp64gram.json 2> $OUT/bench_n10m_fp64gram.err
python bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu > $OUT/bench_n1250k.json 2> $OUT/bench_n1250k.err
python tools/time_weighted.py 10000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/weighted_n10m.log
python bench.py --config 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
PPCA_MIX_MULTI=0 python bench.py --config 5 --no-cpu > $OUT/bench_cfg5_component_by_component.json 2> $OUT/bench_cfg5_component_by_component.err
python bench.py --config 4 --steps 5 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python bench.py --config 2 --steps 50 --warmup 5 --no-cpu > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --config 1 --steps 50 --warmup 5 --no-cpu > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err
for s in "200 16" "256 11" "256 13" "256 16" "300 10" "512 10" "256 10" "200 10" "256 20" "256 32" "256 48" "256 65" "256 80" "256 100" "256 128"; do
  set -- $s
  python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
done
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes.log
PPCA_RECON8=0 python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes_recon8_off.log
python tools/time_passes.py 4000000 200 10 2>&1 | grep -v amdgpu.ids > $OUT/passes_d200_k10.log
python tools/time_passes.py 2000000 200 16 2>&1 | grep -v amdgpu.ids > $OUT/passes_d200_k16.log
python tools/check_additivity.py 10000000 2>&1 | grep -v amdgpu.ids > $OUT/additivity_n10m.log
python tools/outlier_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/outlier_probe.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > $OUT/kt_bench.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_n1250k -- python3 $R/bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu > $OUT/kt_n1250k_bench.json 2> $OUT/kt_n1250k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_out10 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --outliers 10 > $OUT/kt_out10_bench.json 2> $OUT/kt_out10.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg5 -- python3 $R/bench.py --config 5 --no-cpu > $OUT/kt_cfg5_bench.json 2> $OUT/kt_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -- python3 $R/bench.py --config 4 --steps 4 --warmup 1 --no-cpu > $OUT/kt_cfg4_bench.json 2> $OUT/kt_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_passes -- python3 $R/tools/time_passes.py 4000000 256 10 > $OUT/kt_passes.log 2> $OUT/kt_passes.err
cd $R
timeout 700 python tools/soak_em.py > $OUT/soak_em.log 2>&1
timeout 600 python tools/fuzz_gpu2.py > $OUT/fuzz2.log 2>&1
timeout 600 python tools/fuzz_gpu.py > $OUT/fuzz1.log 2>&1
ls -la $OUT
