cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/hist_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py /tmp/ph $GRAFT_REPO_ROOT/gpurun_out/r04_train128_bf16_replay_hist.txt
cp $(find /tmp/ph -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/train_trace.csv
