# usage: bash tools/exp_env.sh "VAR=a VAR=b ..." [workload]   — one bench line per setting
W=${2:-train128_bf16}
for KV in $1; do echo "== $KV"; env $KV python bench.py --workload $W --no-cpu-baseline --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
