#!/bin/bash
# One rocprofv3 kernel trace of the training workload on the GPU box -> histogram of one replay (tools/replay_histogram.py) and every launch of it with its
# grid (tools/replay_launches.py) under gpurun_out/<tag>/.   usage: gpurun -- 'bash tools/trace_train.sh [tag] [workload]'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-trace}
W=${2:-train128_bf16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
rm -rf $OUT/prof
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/prof.log 2> $OUT/prof.err
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py $OUT/prof $OUT/hist.txt 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/replay_launches.py $OUT/prof "" $OUT/all_launches.txt
head -3 $OUT/hist.txt
rm -rf $OUT/prof
