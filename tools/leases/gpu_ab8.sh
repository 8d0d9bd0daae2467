cd $GRAFT_REPO_ROOT
for v in ${VARIANTS}; do
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null >/tmp/o.json
python -c "
import json; j=json.load(open('/tmp/o.json')); print('$v', round(j['value'],2), 'it/s', round(j['ms_per_step'],3), 'ms', round(j['roofline']['frac'],4))"
done
