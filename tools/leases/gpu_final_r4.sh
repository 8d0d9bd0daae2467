# Final measurements of round 4: everything lands under gpurun_out/r4f and is copied into profiles/r04 by
# tools/leases/collect_profiles_r4.sh afterwards.  The PMC passes run FIRST and write profiles/r04/traffic.json on the box, so
# that the bench line that cites it is produced against the file it cites (same commit: tools/commit_stamp.txt).
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $OUT $GRAFT_REPO_ROOT/profiles/r04
cd /tmp && export TMPDIR=/tmp
COMMIT=$(cat $GRAFT_REPO_ROOT/tools/commit_stamp.txt 2>/dev/null)
# the two harnesses are git-ignored binaries: built here when the tree arrived without them
[ -x $GRAFT_REPO_ROOT/tools/mfma_peak/mfma_peak ] || hipcc --offload-arch=gfx950 -O3 -w $GRAFT_REPO_ROOT/tools/mfma_peak/mfma_peak.hip -o $GRAFT_REPO_ROOT/tools/mfma_peak/mfma_peak
[ -x $GRAFT_REPO_ROOT/tools/s4bench/s4bench ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I $GRAFT_REPO_ROOT/ppca_rs_amd/csrc $GRAFT_REPO_ROOT/tools/s4bench/s4bench.hip -o $GRAFT_REPO_ROOT/tools/s4bench/s4bench
# what the part sustains on nothing but MFMAs (cited beside the nominal peak by bench.py)
$GRAFT_REPO_ROOT/tools/mfma_peak/mfma_peak "$COMMIT" > $OUT/mfma_peak.txt 2>&1
tail -1 $OUT/mfma_peak.txt > $GRAFT_REPO_ROOT/profiles/r04/mfma_peak.json
cp $GRAFT_REPO_ROOT/profiles/r04/mfma_peak.json $OUT/mfma_peak.json
export PMC_N=10000000
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write 10000000 $GRAFT_REPO_ROOT/profiles/r04/traffic.json "$COMMIT" > $OUT/traffic.log 2>&1
cp $GRAFT_REPO_ROOT/profiles/r04/traffic.json $OUT/traffic.json
export PMC_N=1000000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_inst -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > $OUT/pmc_inst.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_fp64gram.json 2> $OUT/bench_n10m_fp64gram.err
PPCA_EM8=0 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_em4.json 2> $OUT/bench_n10m_em4.err
PPCA_EM9=0 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_em8.json 2> $OUT/bench_n10m_em8.err
python tools/time_weighted.py 10000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/weighted_n10m.log
python bench.py --config 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
PPCA_LLK8=0 python bench.py --config 5 --no-cpu > $OUT/bench_cfg5_llk2.json 2> $OUT/bench_cfg5_llk2.err
python bench.py --config 4 --steps 5 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
for s in "200 16" "256 11" "256 13" "256 16" "300 10" "512 10" "256 10" "200 10" "256 20" "256 32" "256 48"; do
  set -- $s
  python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
done
# the batched blocked solver (ppca_solve4.hip) against the one-sample-per-wave form it replaces, same run
for rep in 1 2; do for v in 1 0; do
  for s in "256 20" "256 32" "256 48"; do
    set -- $s
    PPCA_SOLVE4=$v python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('PPCA_SOLVE4=$v d=$1 k=$2', round(j['ms_per_step'],2), 'ms per EM iteration')"
  done
  PPCA_SOLVE4=$v python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('PPCA_SOLVE4=$v config 4', round(j['ms_per_step'],2), 'ms per EM iteration')"
done; done > $OUT/solve4_ab.log 2>&1
for k in 20 32 48 64; do tools/s4bench/s4bench $k 1000000 | grep " ms"; done > $OUT/s4bench.log 2>&1
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes.log
PPCA_LLK8=0 python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids | grep -i llk > $OUT/passes_llk2.log
python tools/time_passes.py 2000000 200 16 2>&1 | grep -v amdgpu.ids > $OUT/passes_d200_k16.log
python tools/check_additivity.py 10000000 2>&1 | grep -v amdgpu.ids > $OUT/additivity_n10m.log
python tools/outlier_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/outlier_probe.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/kt_bench.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu > $OUT/kt_cfg5_bench.json 2> $OUT/kt_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu > $OUT/kt_cfg4_bench.json 2> $OUT/kt_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d200_k16 -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d 200 --k 16 --steps 3 --warmup 1 --no-cpu > $OUT/kt_d200_k16_bench.json 2> $OUT/kt_d200_k16.err
cd $GRAFT_REPO_ROOT
# diagnostic builds: per-wave phase table; one-role builds (what each role costs alone); the variants measured this round
python tools/devbuild.py --timing --name=devt > $OUT/devbuild.log 2>&1
PPCA_EM9=0 PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_devt.so python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
PPCA_EM9=1 PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_devt.so python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing_em9.json 2> $OUT/timing_em9.err
for v in "em9 -DNOTHING" "em9front -DE9_ONLY_FRONT" "em9back -DE9_ONLY_BACK" "base" "front -DE8_ONLY_FRONT" "back -DE8_ONLY_BACK" "shared -DE8_SHARED_FACTOR=1" "decoupled -DE8_DECOUPLED=1" "accint64 -DE8_ACC_F64=0" "noearly -DE8_EARLY_DIGDONE=0" "noerrb -DE8_NO_ERRB"; do
  set -- $v; name=$1; shift
  python tools/devbuild.py "$@" --name=v_$name > /dev/null 2>&1
done
for rep in 1 2; do for name in em9 em9front em9back base front back shared decoupled accint64 noearly noerrb; do
  E9=0; case $name in em9*) E9=1;; esac
  PPCA_EM9=$E9 PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_v_$name.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$name', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms per launch')"
done; done > $OUT/variants.log 2>&1
ls -la $OUT
