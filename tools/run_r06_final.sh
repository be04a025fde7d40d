cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.log 2>&1
echo "rc $?" >> gpurun_out/r06_gpu_suite.log
tail -5 gpurun_out/r06_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
bash tools/profile_round.sh r06 > gpurun_out/r06_profile.log 2>&1
tail -30 gpurun_out/r06_profile.log
