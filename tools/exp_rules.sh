run() { echo "== $1: $(python bench.py $1 --no-cpu-baseline --no-extra --no-split-record --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"; }
for rep in 1 2; do
run ""
run "--workload full128"
run "--workload full256"
done
run "--workload train128"
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d or backbone or full_forward or full_size" 2>&1 | grep -E "passed|failed|Error" | tail -2
