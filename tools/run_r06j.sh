cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06j
for sk in 0 1 2; do
  export KPF_G8_SKEW=$sk
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/tools/gemm16_bench.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/tools/gemm16_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_by_run.py /tmp/pf /tmp/pw gemm16 2>&1 | head -8 > $R/gpurun_out/r06j/g16_pmc_sk$sk.txt
done
head -5 $R/gpurun_out/r06j/g16_pmc_sk*.txt
