cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x > $OUT/t1.log 2>&1; echo "rc $?" >> $OUT/t1.log; tail -4 $OUT/t1.log
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x > $OUT/t2.log 2>&1; echo "rc $?" >> $OUT/t2.log; tail -4 $OUT/t2.log
cd /tmp
rm -rf $OUT/prof
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/prof.log 2> $OUT/prof.err
python3 $GRAFT_REPO_ROOT/tools/replay_histogram.py $OUT/prof $OUT/hist.txt 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/replay_launches.py $OUT/prof "" $OUT/all_launches.txt
head -3 $OUT/hist.txt
rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
