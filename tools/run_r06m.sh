cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
timeout 1200 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "xattn_layer21 or bert_stack21 or grouped_desa" -s > gpurun_out/r06m/new_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06m/new_tests.log
grep -E "grouped DESA|fused stack|passed|failed|Error|error|assert" gpurun_out/r06m/new_tests.log | tail -12
timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q > gpurun_out/r06m/training_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06m/training_tests.log | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06m/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r06m/train128_bf16_replay_hist.txt 2>/dev/null
head -1 $GRAFT_REPO_ROOT/gpurun_out/r06m/train128_bf16_replay_hist.txt
grep -E "xattn|tr_stack" $GRAFT_REPO_ROOT/gpurun_out/r06m/train128_bf16_replay_hist.txt
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06m/train_bf16.json 2> gpurun_out/r06m/err.txt
python -c "
import json
d=json.load(open('gpurun_out/r06m/train_bf16.json'))
print('train128_bf16', d['value'], d['ms_per_step'])
"
for dbg in 0 1 2 3; do KPF_TRS_DBG=$dbg python tools/trstack_stamps.py 32 bf16 2>&1 | grep "all four" ; done
