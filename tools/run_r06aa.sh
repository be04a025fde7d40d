cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06aa; mkdir -p $OUT
for mb in 0 200 64 16 0 64; do
KPF_F32_ST_MB=$mb python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('st_mb=$mb', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
