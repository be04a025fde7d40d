cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06r
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "wgrad or weight_grad or batched" > gpurun_out/r06r/t1.log 2>&1; echo "rc $?" >> gpurun_out/r06r/t1.log; tail -3 gpurun_out/r06r/t1.log
for d in 1 0 1 0; do
KPF_WG16S_XCD=$d python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>gpurun_out/r06r/b.err | tail -1 > gpurun_out/r06r/b.json
python - <<PY
import json
d=json.load(open('gpurun_out/r06r/b.json')); print('xcd=$d', d['value'], d['ms_per_step'])
PY
done
