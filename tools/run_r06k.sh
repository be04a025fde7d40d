set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
timeout 1200 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "group_max or grouped_desa or bert_stack21 or ball_group" -s > gpurun_out/r06k/new_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06k/new_tests.log
grep -E "grouped DESA|passed|failed|Error|error" gpurun_out/r06k/new_tests.log | tail -12
timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q > gpurun_out/r06k/training_tests.log 2>&1
tail -2 gpurun_out/r06k/training_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06k/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r06k/train128_bf16_replay_hist.txt 2>/dev/null
head -1 $GRAFT_REPO_ROOT/gpurun_out/r06k/train128_bf16_replay_hist.txt
grep -E "ball_group_bwd|tr_stack" $GRAFT_REPO_ROOT/gpurun_out/r06k/train128_bf16_replay_hist.txt
cd $GRAFT_REPO_ROOT
for g in 1 0; do
KPF_EMB_GROUPED=$g timeout 600 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06k/train_bf16_emb$g.json 2> gpurun_out/r06k/err.txt
python -c "
import json
d=json.load(open('gpurun_out/r06k/train_bf16_emb$g.json'))
print('emb grouped=$g', d['value'], d['ms_per_step'])
"
done
timeout 600 python bench.py --workload cnb512_f16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06k/cnb512.json 2>> gpurun_out/r06k/err.txt
python -c "
import json
d=json.load(open('gpurun_out/r06k/cnb512.json'))
print('cnb512 (nt stores)', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
