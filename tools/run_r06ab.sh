cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_train_gpu.py -m gpu -q -x -k "dwconv or skip_path" > $OUT/t1.log 2>&1; echo "rc $?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for d in "768" "0" "384" "1536" "3072" "768" "0"; do
KPF_DW7_MIN_BLOCKS=$d python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('min_blocks=$d', d['value'], d['ms_per_step'])"
done
