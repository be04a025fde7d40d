cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06ae; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "unstack or grouped_desa" > $OUT/t1.log 2>&1; echo "rc $?" >> $OUT/t1.log; tail -12 $OUT/t1.log
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x > $OUT/t2.log 2>&1; echo "rc $?" >> $OUT/t2.log; tail -3 $OUT/t2.log
for d in 1 2 3; do
python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
