#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r05 [quick]
# Every profiled command is ONE process (--no-extra: bench.py also refuses to start its secondary workloads under a profiler), the file
# that is copied is the one whose rows contain the expected kernel with the expected launch count (tools/pick_stats.py), and the PMC
# passes are checked the same way (tools/collect_traffic.py): VERDICT r04 found a child workload's files behind the headline's labels.
# kernel-trace statistics of the headline command (both backbones on one stream, so per-kernel durations are each kernel's own) and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of the same command; summaries land in gpurun_out/<tag>_*.
TAG=${1:-r05}
MODE=${2:-full}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <kernel> <bench args...>: rocprofv3 --kernel-trace --stats of one bench.py command -> <tag>_<name>_kernel_stats.csv
  local name=$1 kern=$2; shift 2
  rm -rf $OUT/prof_${TAG}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$name -- python3 $ROOT/bench.py "$@" > $OUT/prof_${TAG}_$name.log 2> $OUT/prof_${TAG}_$name.err
  python3 $ROOT/tools/pick_stats.py $OUT/prof_${TAG}_$name $kern $OUT/${TAG}_${name}_kernel_stats.csv $OUT/prof_${TAG}_$name.log
}
traffic() {  # traffic <name> <kernel> <bench args...>: two PMC passes of one bench.py command -> <tag>_traffic<name>.json
  local name=$1 kern=$2; shift 2
  rm -rf $OUT/pmc_${TAG}_fetch$name $OUT/pmc_${TAG}_write$name
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch$name -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${TAG}_fetch$name.log 2> $OUT/pmc_${TAG}_fetch$name.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write$name -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${TAG}_write$name.log 2> $OUT/pmc_${TAG}_write$name.err
  python3 $ROOT/tools/collect_traffic.py $OUT/pmc_${TAG}_fetch$name $OUT/pmc_${TAG}_write$name $OUT/${TAG}_traffic$name.json $kern $OUT/pmc_${TAG}_fetch$name.log \
    && cp $OUT/${TAG}_traffic$name.json $ROOT/profiles/
  rm -rf $OUT/pmc_${TAG}_fetch$name $OUT/pmc_${TAG}_write$name
}
HEAD="--serial-streams --no-cpu-baseline --no-split-record --no-extra"
if [ "$MODE" != rest ]; then  # (rest: the first half already ran)
# configs[1], the headline: 136 igemm_f32_kernel launches per step
stats bench igemm_f32_kernel $HEAD --steps 15
traffic "" igemm_f32_kernel $HEAD --steps 3 --warmup 1
# configs[4]
stats cnb512_f16 gemm16_8ph_kernel --workload cnb512_f16 $HEAD --steps 6 --warmup 2
traffic _cnb512 gemm16_8ph_kernel --workload cnb512_f16 $HEAD --steps 2 --warmup 1
# configs[2]
stats full128_bf16 igemm_h16_kernel --workload full128_bf16 $HEAD --steps 10
traffic _full128_bf16 igemm_h16_kernel --workload full128_bf16 $HEAD --steps 3 --warmup 1
fi
if [ "$MODE" != quick ]; then
  [ "$MODE" != rest ] && stats full128 igemm_f32_kernel --workload full128 $HEAD --steps 10
  # configs[3]: no PMC traffic — rocprofv3 --pmc on the training iteration (eager: counters are collected per dispatch) crashed with a segmentation fault in
  # its FETCH_SIZE pass and did not return from its WRITE_SIZE pass within 40 minutes on this pool (round 5, gpurun_out/r05_profile.log); `traffic` stays null there
  for W in train128 train128_bf16; do
    rm -rf $OUT/prof_${TAG}_$W
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$W -- python3 $ROOT/bench.py --workload $W --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/prof_${TAG}_$W.log 2> $OUT/prof_${TAG}_$W.err
    python3 $ROOT/tools/pick_stats.py $OUT/prof_${TAG}_$W adamw_multi_kernel $OUT/${TAG}_${W}_kernel_stats.csv
    # ONE graph replay of the training iteration cut out of the same trace: launches, kernel time, library share
    python3 $ROOT/tools/replay_histogram.py $OUT/prof_${TAG}_$W $OUT/${TAG}_${W}_replay_hist.txt 2> /dev/null
  done
fi
# the bench lines of the same build, without the profiler (the traffic files just collected are what `roofline.traffic` quotes)
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
if [ "$MODE" != quick ]; then
  for W in full128 full128_bf16 full128_f16 cnb512_f16 train128 train128_bf16 train128_f16 full256; do
    python3 bench.py --workload $W --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
  done
  python3 tools/shape_table.py > $OUT/${TAG}_shape_table.txt 2>/dev/null
  python3 tools/shape_table.py 64 512 f16 KPFusion-convnext-base > $OUT/${TAG}_shape_cnb512.txt 2>/dev/null
  python3 tools/shape_table.py 32 128 bf16 > $OUT/${TAG}_shape_full128_bf16.txt 2>/dev/null
fi
# configs[2]: one synchronised forward at a time, cut out of a kernel trace (what bounds the single-batch latency)
cd /tmp
rm -rf /tmp/pl_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_$TAG -- python3 $ROOT/bench.py --workload full128_bf16 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > /dev/null 2>&1
python3 $ROOT/tools/latency_segments.py /tmp/pl_$TAG $OUT/${TAG}_full128_bf16_latency_segments.txt
rm -rf /tmp/pl_$TAG
cd $ROOT
# keep the merge small: the raw traces stay on the box
for d in $OUT/prof_${TAG}_*; do [ -d "$d" ] && rm -rf "$d"; done
ls -la $OUT | grep ${TAG}_
