set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_reduced_precision_gpu.py tests/test_bench_launcher.py tests/test_training.py -m gpu -x -q -k "seed_sweep or wide_model or deviation or self_launched or loss_scaler or fp16_training or demo_crop" -s > gpurun_out/r06a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06a/tests.log
timeout 300 python tools/launch_sources.py bf16 > gpurun_out/r06a/census_bf16.txt 2>&1
timeout 300 python tools/launch_sources.py bf16 copies > gpurun_out/r06a/census_bf16_copies.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
tail -c 1500 gpurun_out/r06a/tests.log
