cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-11 13 14 14r 15 15r 16}; do
k=${v%r}
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev16k$v.so timeout 300 python bench.py --n 2000000 --d 256 --k $k --steps 3 --warmup 1 --no-cpu 2>/dev/null >/tmp/o.json
python -c "
import json; j=json.load(open('/tmp/o.json')); print('variant $v d=256', round(j['value'],1), 'it/s', round(j['ms_per_step'],3), 'ms')"
done
