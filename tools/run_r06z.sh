cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06z; mkdir -p $OUT
for d in 384 512 640 768 1024 256 384; do
KPF_WG16S_TARGET=$d python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('target=$d', d['value'], d['ms_per_step'])"
done
