mkdir -p gpurun_out/r5h
for f in 1 2 3 4; do
 for m in "" "--one-stream-graph"; do
  echo "in-flight $f $m: $(python bench.py --workload full128_bf16 --no-cpu-baseline --no-extra --steps 40 --warmup 8 --in-flight $f $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d.get('single_batch_latency',{}).get('median_ms'))")"
 done
done > gpurun_out/r5h/exp1.log 2>&1
cat gpurun_out/r5h/exp1.log
