R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6wpred
mkdir -p $OUT
cd $R
(
for c in 0 24576 32768 49152 65536 76800; do
  if [ $c != 0 ]; then export PPCA_GEN_CHUNK=$c; fi
  echo "== chunk $c"; timeout 300 python bench.py --config 4 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg4', j['ms_per_step'])"
done
) 2>&1 | tee $OUT/chunks.log
