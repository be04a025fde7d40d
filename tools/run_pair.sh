python -m pytest tests/test_kernels_train_gpu.py -x -q -k "paired" 2>&1 | tail -8
for W in train128_bf16 train128; do for P in 0 1; do echo "== $W PAIR=$P"; KPF_TRAIN_PAIR=$P python bench.py --workload $W --no-cpu-baseline --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
