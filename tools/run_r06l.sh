set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
timeout 1200 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "bert_stack21 or grouped_desa" -s > gpurun_out/r06l/new_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06l/new_tests.log
grep -E "fused stack|passed|failed|Error|error" gpurun_out/r06l/new_tests.log | tail -12
timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q > gpurun_out/r06l/training_tests.log 2>&1
tail -2 gpurun_out/r06l/training_tests.log
python tools/trstack_stamps.py 32 2>&1 | tail -14
for m in auto f32; do
KPF_TR_MMA=$m timeout 600 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06l/train_bf16_mma_$m.json 2> gpurun_out/r06l/err.txt
python -c "
import json
d=json.load(open('gpurun_out/r06l/train_bf16_mma_$m.json'))
print('stack mma=$m', d['value'], d['ms_per_step'])
"
done
