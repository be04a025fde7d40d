set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -x -q -k "bert_stack21" -s > gpurun_out/r06c/stack_tests.log 2>&1
echo "rc $?" >> gpurun_out/r06c/stack_tests.log
tail -c 1500 gpurun_out/r06c/stack_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06c/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r06c/train128_bf16_replay_hist.txt
head -40 $GRAFT_REPO_ROOT/gpurun_out/r06c/train128_bf16_replay_hist.txt
