cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-dev16}; do
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$v.so timeout 300 python bench.py --n 2000000 --d ${D:-200} --k ${K:-16} --steps 3 --warmup 1 --no-cpu 2>&1 >/tmp/o.json | grep "em16 estep" | tail -1
python -c "
import json; j=json.load(open('/tmp/o.json')); print('$v', j['value'], 'it/s', j['ms_per_step'], 'ms')"
if [ -n "$CHECK" ]; then PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$v.so python tools/em16_check.py 2>&1 | grep "k=${K:-16} " | cut -c1-150; fi
done
