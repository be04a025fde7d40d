#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r04
# kernel-trace statistics of the headline command (both backbones on one stream, so per-kernel durations are each kernel's own) and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of the same command; summaries land in gpurun_out/<tag>_*.
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/bench.py --serial-streams --no-cpu-baseline --no-split-record"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_f32 -- python3 $CMD --steps 15 > $OUT/prof_${TAG}_f32.log 2>&1
cp $(find $OUT/prof_${TAG}_f32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -- python3 $CMD --steps 3 --warmup 1 > $OUT/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -- python3 $CMD --steps 3 --warmup 1 > $OUT/pmc_${TAG}_write.log 2>&1
python3 $ROOT/tools/collect_traffic.py $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/${TAG}_traffic.json igemm_f32_kernel
# configs[4]: HBM traffic of the dominant 16-bit kernel (gemm16_8ph_kernel), same two PMC passes
CMD4="$ROOT/bench.py --workload cnb512_f16 --serial-streams --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch4 -- python3 $CMD4 --steps 2 --warmup 1 > $OUT/pmc_${TAG}_fetch4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write4 -- python3 $CMD4 --steps 2 --warmup 1 > $OUT/pmc_${TAG}_write4.log 2>&1
python3 $ROOT/tools/collect_traffic.py $OUT/pmc_${TAG}_fetch4 $OUT/pmc_${TAG}_write4 $OUT/${TAG}_traffic_cnb512.json gemm16_8ph_kernel
for W in full128 full128_bf16 cnb512_f16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$W -- python3 $ROOT/bench.py --workload $W --serial-streams --no-cpu-baseline --no-split-record --steps 10 > $OUT/prof_${TAG}_$W.log 2>&1
  cp $(find $OUT/prof_${TAG}_$W -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${W}_kernel_stats.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train128 -- python3 $ROOT/bench.py --workload train128 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_${TAG}_train128.log 2>&1
cp $(find $OUT/prof_${TAG}_train128 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_train128_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train128_bf16 -- python3 $ROOT/bench.py --workload train128_bf16 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_${TAG}_train128_bf16.log 2>&1
cp $(find $OUT/prof_${TAG}_train128_bf16 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_train128_bf16_kernel_stats.csv
# ONE graph replay of the training iteration cut out of the same traces: launches, kernel time, library share (tools/replay_histogram.py)
python3 $ROOT/tools/replay_histogram.py $OUT/prof_${TAG}_train128_bf16 $OUT/${TAG}_train128_bf16_replay_hist.txt 2> /dev/null
python3 $ROOT/tools/replay_histogram.py $OUT/prof_${TAG}_train128 $OUT/${TAG}_train128_replay_hist.txt 2> /dev/null
# the bench lines of the same build, without the profiler (the traffic files just collected are what `roofline.traffic` quotes)
cd $ROOT
cp $OUT/${TAG}_traffic.json $OUT/${TAG}_traffic_cnb512.json $ROOT/profiles/ 2>/dev/null
python3 bench.py --no-extra > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for W in full128 full128_bf16 cnb512_f16 train128 train128_bf16; do
  python3 bench.py --workload $W --no-cpu-baseline > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
done
# keep the merge small: the raw traces stay on the box
for d in $OUT/prof_${TAG}_* $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_fetch4 $OUT/pmc_${TAG}_write4; do [ -d "$d" ] && rm -rf "$d"; done
python3 tools/shape_table.py > $OUT/${TAG}_shape_table.txt 2>/dev/null
[ -x tools/bin/mfma_issue_rate2 ] && tools/bin/mfma_issue_rate2 > $OUT/${TAG}_mfma_issue_rate2.txt
ls -la $OUT | grep ${TAG}_
