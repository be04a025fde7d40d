S5="16384:1536:384:1=8,4096:3072:768:1=0,65536:768:192:1=8,4096:384:3456:3=7,262144:128:64:1=2,4096:384:768:1=7,65536:192:96:1=6"
run() { echo "== $1 $3: $(KPF_TILE_RULES="$2" python bench.py $3 --no-cpu-baseline --no-extra --no-split-record --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"; }
for rep in 1 2; do
run built-in-rules ""
run S5-table "$S5"
done
run built-in-rules "" "--workload full128"
run built-in-rules "" "--workload full256"
run built-in-rules "" "--workload train128"
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d or backbone or full_forward or full_size" 2>&1 | grep -E "passed|failed|Error" | tail -2
