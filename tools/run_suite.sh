cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.log 2>&1
echo "rc $?" >> gpurun_out/r06_gpu_suite.log
tail -6 gpurun_out/r06_gpu_suite.log
for d in 1 2; do
python bench.py --workload train128_bf16 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
