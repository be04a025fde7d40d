cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "wgrad or weight_grad or deferred or grouped or batched or dwconv" > gpurun_out/r06q/t1.log 2>&1; echo "rc $?" >> gpurun_out/r06q/t1.log; tail -15 gpurun_out/r06q/t1.log
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x > gpurun_out/r06q/t2.log 2>&1; echo "rc $?" >> gpurun_out/r06q/t2.log; tail -8 gpurun_out/r06q/t2.log
for d in "0 24" "1 3" "1 8" "1 24" "0 24"; do
set -- $d
KPF_REDUCE_DEFER=$1 KPF_REDUCE_BATCH=$2 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>gpurun_out/r06q/b.err | tail -1 > gpurun_out/r06q/b.json
python - <<PY
import json
d=json.load(open('gpurun_out/r06q/b.json')); print('defer=$1 batch=$2', d['value'], d['ms_per_step'])
PY
done
