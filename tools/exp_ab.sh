# A/B on ONE box (boxes differ by 10-15 %): variants are source trees prepared locally under .ab/<name>/ (git archive <ref> keypointfusion_amd bench.py | tar -x -C .ab/<name>),
# all on the working tree's library (same ABI).  usage: bash tools/exp_ab.sh name1 name2 ...   ("wt" = the working tree);  W=<workload>
W=${W:-train128_bf16}
ROOT=$PWD
export KPF_LIB_PATH=$ROOT/keypointfusion_amd/libkpf_hip.so
for rep in 1 2; do
for V in "$@"; do
  if [ "$V" = "wt" ]; then D=$ROOT; else D=$ROOT/.ab/$V; fi
  if [ -f $ROOT/.ab/lib_$V.so ]; then export KPF_LIB_PATH=$ROOT/.ab/lib_$V.so; else export KPF_LIB_PATH=$ROOT/keypointfusion_amd/libkpf_hip.so; fi  # (a variant may bring its own library build)
  echo -n "$V: "; (cd $D && python bench.py --workload $W --no-cpu-baseline --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
done; done
