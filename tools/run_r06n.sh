cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
timeout 600 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "batched_weight_gradient or deferred" > gpurun_out/r06n/t1.log 2>&1; echo "rc $?" >> gpurun_out/r06n/t1.log; tail -15 gpurun_out/r06n/t1.log
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x > gpurun_out/r06n/t2.log 2>&1; echo "rc $?" >> gpurun_out/r06n/t2.log; tail -8 gpurun_out/r06n/t2.log
for d in 1 0 1 0; do
KPF_REDUCE_DEFER=$d python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra 2>gpurun_out/r06n/b$d.err | tail -1 > gpurun_out/r06n/b$d.json
python - <<PY
import json
d=json.load(open('gpurun_out/r06n/b$d.json')); print('defer=$d', d['value'], d['ms_per_step'])
PY
done
