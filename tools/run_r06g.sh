cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06g
rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -- python3 $GRAFT_REPO_ROOT/bench.py --workload full128_bf16 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r06g/lat_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/latency_segments.py /tmp/pl $GRAFT_REPO_ROOT/gpurun_out/r06g/full128_bf16_latency_segments.txt
tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/r06g/lat_run.log
