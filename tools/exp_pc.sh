cat > /tmp/pc.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import keypointfusion_amd.graphs as G
G.assert_replay_is_sound = lambda *a, **k: None
import keypointfusion_amd.training as T
sys.argv = ["bench.py", "--workload", sys.argv[1], "--no-cpu-baseline", "--no-extra", "--steps", "20", "--warmup", "5"]
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"), run_name="__main__")
PY
for W in train128_bf16 train128; do
for PC in 0 1; do
echo "== $W PACKET_CAPTURE=$PC"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=$PC python /tmp/pc.py $W 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
