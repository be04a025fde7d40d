run() { echo "== $1 $2: $(env $1 python bench.py --workload $2 --no-cpu-baseline --no-extra --steps ${3:-10} --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"; }
for rep in 1 2; do
run KPF_NO_KXK16=1 cnb512_f16
run A=1 cnb512_f16
done
run KPF_NO_KXK16=1 full128_bf16_r18 30
run A=1 full128_bf16_r18 30
run KPF_NO_KXK16=1 full128_bf16 30
run A=1 full128_bf16 30
run KPF_NO_KXK16=1 train128_bf16 20
run A=1 train128_bf16 20
python -m pytest tests/test_reduced_precision_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
