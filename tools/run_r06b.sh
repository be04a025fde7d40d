set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "bert_stack21" -s > gpurun_out/r06b/stack_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06b/stack_tests.log
tail -c 2500 gpurun_out/r06b/stack_tests.log
timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q > gpurun_out/r06b/training_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06b/training_tests.log
tail -c 1500 gpurun_out/r06b/training_tests.log
for f in 1 0; do
KPF_TR_FUSED=$f timeout 600 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06b/train_bf16_fused$f.json 2> gpurun_out/r06b/train_bf16_fused$f.err
python -c "
import json,sys
d=json.load(open('gpurun_out/r06b/train_bf16_fused$f.json'))
print('fused=$f', d['value'], d['ms_per_step'])
"
done
