mkdir -p gpurun_out/r5h
for w in full128_bf16 full128; do for f in 1 2 3 4; do
  echo "== $w in-flight $f: $(python bench.py --workload $w --no-cpu-baseline --no-extra --steps 40 --warmup 8 --in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'])")"
done; done > gpurun_out/r5h/stage.log 2>&1
cat gpurun_out/r5h/stage.log
