"""The round's evidence tooling (tools/collect_traffic.py, tools/pick_stats.py) refuses the failure VERDICT r04 found: a child workload's
rocprofv3 files behind the headline's label, a launch count that is not a whole number of steps, a traffic figure below the algorithmic bytes."""
import json
import os
import subprocess
import sys

from conftest import ROOT

HDR = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"


def _pmc(d, pid, counter, kernel, n, value):
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    with open(os.path.join(d, "host", "%d_counter_collection.csv" % pid), "w") as f:
        f.write(HDR)
        for i in range(n):
            f.write('%d,%d,0,1,%d,1,65536,7,"void (anonymous namespace)::%s<4, 4>((anonymous namespace)::ConvArgs)",256,0,0,128,0,32,%s,%f,0,1\n' % (i, i, pid, kernel, counter, value))


def _log(path, kernel, lps, algo):
    rec = {"steps": 3, "warmup": 1, "launch": "eager", "config": {"workload": "test workload"},
           "roofline": {"kernel": kernel, "launches_per_step": lps, "algo_bytes_per_launch": algo}}
    with open(path, "w") as f:
        f.write("some line\n" + json.dumps(rec) + "\n")


def _run(tool, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], capture_output=True, text=True)


def test_collect_traffic_accepts_one_process_and_whole_steps(tmp_path):
    fd, wd, log, out = tmp_path / "f", tmp_path / "w", tmp_path / "log", tmp_path / "t.json"
    _pmc(str(fd), 100, "FETCH_SIZE", "igemm_f32_kernel", 12, 50000.0)   # KiB per launch; doubled by the tool: 102.4 MB
    _pmc(str(wd), 101, "WRITE_SIZE", "igemm_f32_kernel", 12, 55000.0)   # 56.3 MB
    _log(str(log), "igemm_f32_kernel", 4, 138_000_000)
    r = _run("collect_traffic.py", fd, wd, out, "igemm_f32_kernel", log)
    assert r.returncode == 0, r.stderr
    t = json.load(open(out))
    assert t["launches_profiled"] == 12 and t["steps_profiled"] == 3 and abs(t["hbm_bytes_per_launch"] - (2 * 50000 + 55000) * 1024) < 1
    assert 1.1 < t["ratio_to_algorithmic"] < 1.2


def test_collect_traffic_refuses_child_processes_partial_steps_and_impossible_figures(tmp_path):
    fd, wd, log, out = tmp_path / "f", tmp_path / "w", tmp_path / "log", tmp_path / "t.json"
    _log(str(log), "igemm_f32_kernel", 4, 138_000_000)
    _pmc(str(fd), 100, "FETCH_SIZE", "igemm_f32_kernel", 12, 50000.0)
    _pmc(str(fd), 200, "FETCH_SIZE", "igemm_f32_kernel", 40, 9000.0)    # a child workload's small launches of the same kernel (r04's contamination)
    _pmc(str(wd), 101, "WRITE_SIZE", "igemm_f32_kernel", 12, 55000.0)
    r = _run("collect_traffic.py", fd, wd, out, "igemm_f32_kernel", log)
    assert r.returncode != 0 and "2 processes" in r.stderr and not out.exists()
    os.remove(str(fd / "host" / "200_counter_collection.csv"))
    _pmc(str(wd), 101, "WRITE_SIZE", "igemm_f32_kernel", 13, 55000.0)   # not a whole number of 4-launch steps
    r = _run("collect_traffic.py", fd, wd, out, "igemm_f32_kernel", log)
    assert r.returncode != 0 and "whole number" in r.stderr and not out.exists()
    _pmc(str(wd), 101, "WRITE_SIZE", "igemm_f32_kernel", 12, 5500.0)
    _pmc(str(fd), 100, "FETCH_SIZE", "igemm_f32_kernel", 12, 5000.0)    # 16 MB per launch against 138 MB algorithmic: impossible
    r = _run("collect_traffic.py", fd, wd, out, "igemm_f32_kernel", log)
    assert r.returncode != 0 and "below the algorithmic" in r.stderr and not out.exists()


def test_pick_stats_takes_the_file_with_the_kernel_and_checks_the_launch_count(tmp_path):
    d = tmp_path / "prof" / "host"
    os.makedirs(d)
    hdr = '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
    open(d / "1_kernel_stats.csv", "w").write(hdr + '"void (anonymous namespace)::igemm_f32_kernel<4, 4>(ConvArgs)",24,1000,41.6,90.0,1,2,0.1\n')
    open(d / "2_kernel_stats.csv", "w").write(hdr + '"void (anonymous namespace)::gemm16_8ph_kernel<1, 4, true>(ConvArgs)",99,1000,10.1,90.0,1,2,0.1\n')
    log = tmp_path / "log"
    _log(str(log), "igemm_f32_kernel", 4, 1)  # 4 x (3 steps + 1 warm-up + 2 instrumented passes) = 24 launches
    out = tmp_path / "out.csv"
    r = _run("pick_stats.py", tmp_path / "prof", "igemm_f32_kernel", out, log)
    assert r.returncode == 0, r.stderr
    assert "igemm_f32_kernel" in open(out).read()
    _log(str(log), "igemm_f32_kernel", 5, 1)
    r = _run("pick_stats.py", tmp_path / "prof", "igemm_f32_kernel", tmp_path / "out2.csv", log)
    assert r.returncode != 0 and "the command issued 30" in r.stderr


def test_replay_grids_picks_one_replay_and_groups_by_kernel_and_grid(tmp_path):
    """tools/replay_grids.py: iterations are delimited by pack_weights_multi_kernel, the shortest complete one is the replay, and its launches are grouped by
    (kernel, workgroups, workgroup size) — the table that showed the LayerNorm backward on one workgroup per CU."""
    d = tmp_path / "p" / "host"
    os.makedirs(d)
    hdr = "Kind,Agent_Id,Queue_Id,Kernel_Name,Workgroup_Size_X,Workgroup_Size_Y,Workgroup_Size_Z,Grid_Size_X,Grid_Size_Y,Grid_Size_Z,Start_Timestamp,End_Timestamp\n"
    rows, t = [], 0

    def emit(name, wgs, wg, dur):
        nonlocal t
        rows.append('KERNEL_DISPATCH,0,1,"%s",%d,1,1,%d,1,1,%d,%d\n' % (name, wg, wgs * wg, t, t + dur))
        t += dur + 100

    for it, slow in enumerate((3, 1, 1)):  # an eager (slower) iteration, then two replays; a last marker closes the third
        emit("(anonymous namespace)::pack_weights_multi_kernel(float*)", 64, 256, 1000 * slow)
        for i in range(600):
            if i % 3 == 0:
                emit("void (anonymous namespace)::ln_bwd_kernel<float>(float const*)", 256, 256, 20000 * slow)
            else:
                emit("void (anonymous namespace)::igemm_f32_kernel<2, 1, 1, 4>(ConvArgs)", 42, 256, 5000 * slow)
    emit("(anonymous namespace)::pack_weights_multi_kernel(float*)", 64, 256, 1000)
    with open(d / "1_kernel_trace.csv", "w") as f:
        f.write(hdr + "".join(rows))
    out = tmp_path / "grids.txt"
    r = _run("replay_grids.py", tmp_path / "p", out)
    assert r.returncode == 0, r.stderr
    text = open(out).read().splitlines()
    assert text[0].startswith("one replay: 601 launches")
    assert " 200 x    20.0 us" in text[1] and "256 workgroups x  256 threads   ln_bwd_kernel<float>" in text[1]
    assert " 400 x     5.0 us" in text[2] and "42 workgroups x  256 threads   igemm_f32_kernel<2, 1, 1, 4>" in text[2]
