set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "bert_stack21" -s > gpurun_out/r06d/stack_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06d/stack_tests.log
tail -c 1200 gpurun_out/r06d/stack_tests.log
timeout 900 python -m pytest tests/test_training.py -m gpu -x -q -k "reference_loss or bit_identical or replays" > gpurun_out/r06d/training_tests.log 2>&1
tail -3 gpurun_out/r06d/training_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06d/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r06d/train128_bf16_replay_hist.txt
grep -E "one replay|tr_stack" $GRAFT_REPO_ROOT/gpurun_out/r06d/train128_bf16_replay_hist.txt
cd $GRAFT_REPO_ROOT
# configs[4] stage-3 shape: store policy A/B
for st in 0 1 2; do
KPF_G8_ST=$st timeout 600 python bench.py --workload cnb512_f16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06d/cnb512_st$st.json 2> gpurun_out/r06d/cnb512_st$st.err
python -c "
import json
d=json.load(open('gpurun_out/r06d/cnb512_st$st.json'))
print('KPF_G8_ST=$st', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
done
