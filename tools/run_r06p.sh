cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
for d in "0 24" "1 2" "1 3" "1 6" "1 12" "1 24" "0 24"; do
set -- $d
KPF_REDUCE_DEFER=$1 KPF_REDUCE_BATCH=$2 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>gpurun_out/r06p/b.err | tail -1 > gpurun_out/r06p/b.json
python - <<PY
import json
d=json.load(open('gpurun_out/r06p/b.json')); print('defer=$1 batch=$2', d['value'], d['ms_per_step'])
PY
done
