set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
timeout 1200 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "group_max or grouped_desa or bert_stack21 or ball_group" -s > gpurun_out/r06e/new_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06e/new_tests.log
grep -E "grouped DESA|passed|failed|Error|error" gpurun_out/r06e/new_tests.log | tail -20
timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q > gpurun_out/r06e/training_tests.log 2>&1
tail -3 gpurun_out/r06e/training_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06e/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r06e/train128_bf16_replay_hist.txt
head -3 $GRAFT_REPO_ROOT/gpurun_out/r06e/train128_bf16_replay_hist.txt
cd $GRAFT_REPO_ROOT
for g in 1 0; do
KPF_DESA_GROUPED=$g timeout 600 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06e/train_bf16_grouped$g.json 2> gpurun_out/r06e/train_bf16_grouped$g.err
python -c "
import json
d=json.load(open('gpurun_out/r06e/train_bf16_grouped$g.json'))
print('grouped=$g', d['value'], d['ms_per_step'])
"
done
