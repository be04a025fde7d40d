"""bench.py's own launcher (replaces train.py:81 / demo_RGBD.py:49, `DataParallel(net).cuda()`: one command, all GPUs of the node).
CPU: the command `python bench.py --gpus 8` would start is the driver's form, and the traffic guard refuses impossible figures.
GPU (one device is enough): the launcher route with ONE RCCL rank, as a fresh child process, for the eval headline and for the graphed
data-parallel training step — world_size, the RCCL backend and the one-graph `overlap` form are asserted from the printed line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench.py makes no HIP call)


def test_launcher_builds_the_drivers_torchrun_command_for_8_gpus():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.launch_command(8, argv, port=29641)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29641"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv  # the ranks get the launcher's own arguments, unchanged
    assert bench.launch_command(2, [])[cmd.index("--master-port") + 1].isdigit()  # a free port is picked when none is given
    env = bench.launch_env({"PATH": "/usr/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    assert bench.launch_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"  # an explicit setting wins


def test_launcher_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 8 but WORLD_SIZE=2" in out.stderr


def test_profiler_preload_is_detected(monkeypatch):
    for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY"):
        monkeypatch.delenv(k, raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()


def _run_bench(extra_args):
    env = dict(os.environ, KPF_BENCH_FORCE_DIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extra", "--no-cpu-baseline", "--no-split-record",
                          "--steps", "3", "--warmup", "2"] + extra_args, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # ONE JSON line, relayed from rank 0
    assert "launching 1 rank(s)" in out.stderr
    return json.loads(lines[0])


@pytest.mark.gpu
def test_self_launched_one_rank_rccl_eval_bench():
    rec = _run_bench([])
    assert rec["world_size"] == 1 and rec["n_gpus"] == 1 and rec["collective_backend"].startswith("rccl")
    assert rec["scaling"] == "weak" and rec["value"] > 0 and rec["roofline"]["kernel"] == "igemm_f32_kernel"
    assert rec["roofline"]["launches_per_step"] == 136
    # what makes the first N > 1 line diagnosable: every rank's own step time and one measured point of the fabric
    pr = rec["per_rank_ms_per_step"]
    assert len(pr["all"]) == 1 and pr["min"] == pr["max"] == pr["all"][0] and abs(pr["max"] - rec["ms_per_step"]) < 1e-2
    probe = rec["collectives"]["probe_64MiB"]
    assert probe["bytes"] == 64 * 1024 * 1024 and probe["ms"] > 0 and probe["algbw_GBps"] > 0 and probe["busbw_GBps"] == 0.0  # (one rank: nothing crosses a link)


@pytest.mark.gpu
def test_self_launched_one_rank_rccl_graphed_training_step():
    rec = _run_bench(["--workload", "train128_bf16"])
    assert rec["world_size"] == 1 and rec["collective_backend"].startswith("rccl")
    assert rec["dp_graph"]["mode"] == "overlap"  # forward, loss, backward WITH the bucket collectives, optimiser: one captured graph
    assert rec["dp_graph"]["payload_bytes_per_rank"] > 200e6 and rec["launch"].startswith("hipGraph replay")
    assert "training iteration" in rec["config"]["workload"]
    co = rec["collectives"]
    assert len(co["buckets"]) == rec["dp_graph"]["buckets"] >= 2 and all(b["ms"] > 0 and b["bytes"] > 0 for b in co["buckets"])
    assert co["all_buckets"]["bytes"] == rec["dp_graph"]["payload_bytes_per_rank"]
    assert len(rec["per_rank_ms_per_step"]["all"]) == 1 and "prediction" in rec["dp_graph"]
