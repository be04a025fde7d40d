cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_training.py -m gpu -q -x -k "graphed" > $OUT/t2.log 2>&1; echo "rc $?" >> $OUT/t2.log; tail -3 $OUT/t2.log
for d in 0 512 2048 0 512; do
KPF_WG16_BIG_M=$d python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bigM=$d', d['value'], d['ms_per_step'])"
done
