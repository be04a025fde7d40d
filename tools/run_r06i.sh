cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06i
for st in 0 1 2; do
  export KPF_G8_ST=$st
  python3 $R/tools/gemm16_bench.py > $R/gpurun_out/r06i/g16_time_st$st.txt 2>&1
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/tools/gemm16_bench.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/tools/gemm16_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_by_run.py /tmp/pf /tmp/pw gemm16 > $R/gpurun_out/r06i/g16_pmc_st$st.txt 2>&1
done
head -12 $R/gpurun_out/r06i/g16_time_st*.txt
head -14 $R/gpurun_out/r06i/g16_pmc_st*.txt
