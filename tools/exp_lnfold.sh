mkdir -p gpurun_out/r5h
for rep in 1 2; do for f in 0 1; do
  echo "== cnb512_f16 KPF_LN_FOLD=$f: $(KPF_LN_FOLD=$f python bench.py --workload cnb512_f16 --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('whole_step_tflops'))")"
done; done > gpurun_out/r5h/lnfold.log 2>&1
cat gpurun_out/r5h/lnfold.log
python -m pytest tests/test_reduced_precision_gpu.py -x -q -m gpu 2>&1 | tail -5
