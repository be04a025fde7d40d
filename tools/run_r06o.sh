cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "wgrad or weight_grad or deferred or grouped or batched or dwconv or linear" > $OUT/t1.log 2>&1; echo "rc $?" >> $OUT/t1.log; tail -4 $OUT/t1.log
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x > $OUT/t2.log 2>&1; echo "rc $?" >> $OUT/t2.log; tail -3 $OUT/t2.log
cd /tmp
for d in "1 8"; do
set -- $d
export KPF_REDUCE_DEFER=$1 KPF_REDUCE_BATCH=$2
rm -rf $OUT/prof$1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/prof$1.log 2> $OUT/prof$1.err
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py $OUT/prof$1 $OUT/hist$1.txt 2>/dev/null
grep -E "one replay|reduce" $OUT/hist$1.txt
rm -rf $OUT/prof$1
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
