cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $OUT
for d in "768 4" "384 4" "512 8" "1024 4" "1536 2" "256 8" "768 4"; do
set -- $d
KPF_DW7_TARGET=$1 KPF_DW7_MINROWS=$2 python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>$OUT/b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('target=$1 minrows=$2', d['value'], d['ms_per_step'])"
done
