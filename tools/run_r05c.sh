#!/bin/bash
# one-off experiment batch (GPU box)
O=gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_kernels_train_gpu.py tests/test_reduced_precision_gpu.py tests/test_training.py -m gpu -x -q > $O/r05c_gputest.log 2>&1; tail -3 $O/r05c_gputest.log
B="python bench.py --no-cpu-baseline --no-extra"
$B --workload cnb512_f16 > $O/r05c_cnb_base.json 2> $O/r05c_cnb_base.err
KPF_MLP16_CHUNKS=2 $B --workload cnb512_f16 > $O/r05c_cnb_ch2.json 2> $O/r05c_cnb_ch2.err
KPF_MLP16_CHUNKS=4 $B --workload cnb512_f16 > $O/r05c_cnb_ch4.json 2> $O/r05c_cnb_ch4.err
$B --workload full128_bf16 --steps 30 --warmup 5 > $O/r05c_full_ring.json 2> $O/r05c_full_ring.err
KPF_NO_RING16=1 $B --workload full128_bf16 --steps 30 --warmup 5 > $O/r05c_full_noring.json 2> $O/r05c_full_noring.err
$B --workload train128_bf16 --steps 12 > $O/r05c_train_ring.json 2> $O/r05c_train_ring.err
KPF_NO_RING16=1 $B --workload train128_bf16 --steps 12 > $O/r05c_train_noring.json 2> $O/r05c_train_noring.err
for f in cnb_base cnb_ch2 cnb_ch4 full_ring full_noring train_ring train_noring head head_nopre; do python - <<PY
import json
try:
    d=json.loads(open("$O/r05c_$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d.get("single_batch_latency"))
except Exception as e: print("$f", "failed", e)
PY
done
python tools/shape_table.py > $O/r05c_shape_headline.txt 2>&1
$B > $O/r05c_head.json 2> $O/r05c_head.err
KPF_NO_PRE_RES=1 $B > $O/r05c_head_nopre.json 2> $O/r05c_head_nopre.err
python tools/precision_stats.py convnext-tiny 16 8 > $O/r05c_precision_stats.txt 2>&1
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$O/pmc_g16_fetch -- python3 $ROOT/tools/gemm16_bench.py > $ROOT/$O/r05c_g16_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$O/pmc_g16_write -- python3 $ROOT/tools/gemm16_bench.py > $ROOT/$O/r05c_g16_write.log 2>&1
cd $ROOT
python tools/pmc_by_run.py $O/pmc_g16_fetch $O/pmc_g16_write > $O/r05c_g16_pmc_by_shape.txt 2>&1
rm -rf $O/pmc_g16_fetch $O/pmc_g16_write
tail -5 $O/r05c_precision_stats.txt

for f in head head_nopre; do python - <<PY
import json
try:
    d=json.loads(open("$O/r05c_$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"])
except Exception as e: print("$f", "failed", e)
PY
done
