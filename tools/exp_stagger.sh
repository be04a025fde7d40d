# A/B of the start skew between co-resident workgroups of igemm_f32_kernel (KPF_STAGGER: cap in units of ~1 us; 0 = off) on the headline and on the fp32 workloads
mkdir -p gpurun_out/r5h
for rep in 1 2; do
for s in 0 8; do
  echo "== bench KPF_STAGGER=$s: $(KPF_STAGGER=$s python bench.py --no-cpu-baseline --no-extra --no-split-record --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done > gpurun_out/r5h/stagger2.log 2>&1
for w in full256 train128 full128; do for s in 0 8; do
  echo "== $w KPF_STAGGER=$s: $(KPF_STAGGER=$s python bench.py --workload $w --no-cpu-baseline --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done >> gpurun_out/r5h/stagger2.log 2>&1
cat gpurun_out/r5h/stagger2.log
