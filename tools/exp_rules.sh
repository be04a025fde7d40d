run() { echo "== $1: $(python bench.py $1 --no-cpu-baseline --no-extra --no-split-record --steps ${2:-30} --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"; }
run ""
run "--workload train128" 20
run "--workload train128" 20
run "--workload train128_bf16" 20
run "--workload train128_bf16" 20
run "--workload full128"
run "--workload cnb512_f16" 10
python -m pytest tests/test_training.py tests/test_kernels_train_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d or backbone or full_forward or full_size" 2>&1 | grep -E "passed|failed|Error" | tail -2
