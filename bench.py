#!/usr/bin/env python3
"""Benchmark of the KPFusion forward hot path on MI355X (driver contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched once per rank by torch.distributed.run)

Workload = BASELINE.json configs[1]: B=64 synthetic 256x256 RGB-D crops per GPU, KPFusion-convnext-tiny, both UNet
backbones (depth + RGB) forward-only, fp32, eval.  (The fusion head only exists at 128x128 in the reference — SURVEY D1
— so configs[1] is "backbones only"; `--workload full128` times the whole model at B=64, 128x128 instead.)
A "step" is one forward over one batch already resident in HBM.  N GPUs = N independent shards of the batch
dimension (weak scaling, no data-path collective: every sample is independent in eval).

One JSON line is printed by rank 0 with, besides the contract fields:
  roofline     : the dominant MFMA kernel family — achieved = algorithmic conv/GEMM FLOPs (2*M*N*K) of all its launches in one step /
                 summed device time of those launches (HIP events on the launch stream, measured in an instrumented pass after
                 the timed region).  peak: split-arithmetic kernels execute 3 f16 MFMAs per algorithmic product, so their roof is
                 2500/3 = 833.3 algorithmic TFLOP/s (dense f16 MFMA peak / 3); f32-input MFMA kernels: 157.3 TFLOP/s.
  f32_mfma     : the same workload timed with KPF_GEMM=f32 (every GEMM on v_mfma_f32_16x16x4_f32), rank 0 at N=1, for reference.
  cpu_baseline : the CPU oracle (oracle/kpf_oracle.py, torch-CPU fp32 = the reference's own arithmetic) on the host
                 cores of this box, on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 matrix peak
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3  # 3 f16 MFMAs per algorithmic fp32 product (hi*hi + hi*lo + lo*hi)


def kernel_peak(name):
    return PEAK_SPLIT_TFLOPS if "split" in name else PEAK_F32_MFMA_TFLOPS
NET = "KPFusion-convnext-tiny"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", default="backbones256", choices=["backbones256", "full128"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-streams", action="store_true", help="issue both backbones on one stream (per-kernel profiling: rocprofv3 "
                    "durations of overlapped kernels are otherwise shared-GPU durations)")
    ap.add_argument("--graph", action="store_true", help="full128: replay the forward from a captured hipGraph")
    ap.add_argument("--no-graph", action="store_true", help="backbones256: issue the ~250 launches of a step from Python instead of replaying the "
                    "captured hipGraph (the default; same device time, but the step then depends on the host keeping up)")
    ap.add_argument("--per-launch", default="", help="write a per-launch table of the implicit-GEMM kernel to this file")
    ap.add_argument("--cpu-sample", type=int, default=8, help="images in the CPU-baseline sample")
    ap.add_argument("--gemm", default=None, choices=["split", "f32"], help="GEMM arithmetic (default: the engine's, KPF_GEMM or 'split')")
    ap.add_argument("--no-f32-reference", action="store_true", help="skip the extra KPF_GEMM=f32 timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    dist = None
    if world > 1 or os.environ.get("KPF_BENCH_FORCE_DIST"):  # (the env switch exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from keypointfusion_amd import engine as E, lib as L
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict
    L.load()
    if args.gemm:
        E.GEMM_MODE = args.gemm

    if args.workload == "full128":
        args.size = 128
    B, S = args.batch, args.size
    model = KPFusion(NET, "", 21, "dexycb", "")
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(NET, 0).items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.use_graphs = bool(args.graph)
    model._plan(dev).serial_streams = bool(args.serial_streams)
    hb = synthetic_batch(B, S, seed=1 + rank)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in hb.items()}

    class _Loader:
        img_size, flip = 128, 1

    # backbones256 replays a captured hipGraph by default (both streams as parallel branches, inputs copied into its static buffers
    # inside the timed step): ~1.7 ms of host work per step instead of ~3.3 ms, so host jitter on a shared box cannot stall the GPU
    graph_on = [args.workload == "backbones256" and not args.no_graph and not args.serial_streams]

    def step():
        with torch.no_grad():
            if args.workload == "backbones256":
                if graph_on[0]:
                    model._plan(dev).backbones_graphed(batch["img"], batch["img_rgb"])
                else:
                    model._plan(dev).backbones(batch["img"], batch["img_rgb"])
            else:
                model(batch["img_rgb"], batch["img"], batch["pcl"], _Loader(), batch["center"], batch["M"], batch["cube"],
                      batch["cam_para"], 0.8)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if graph_on[0]:  # capture during warm-up; if the runtime refuses the capture, time the eager launches instead of failing
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print("bench.py: hipGraph capture failed (%s: %s); falling back to eager launches" % (type(e).__name__, e), file=sys.stderr)
            graph_on[0] = False
            model._plans.clear()
            model._plan(dev).serial_streams = bool(args.serial_streams)
    launch_mode = "hipGraph replay" if graph_on[0] or (args.graph and args.workload == "full128") else "eager"
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0  # host time to enqueue the K steps (close to dt => the step is launch-bound)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- instrumented pass: per-launch HIP events around every MFMA-kernel launch (same stream, streams serialised) ----
    roofline = None
    if rank == 0:
        plan = model._plan(dev)
        model.use_graphs = False  # the instrumented pass needs the individual launches
        graph_on[0] = False
        plan.serial_streams = True  # one stream: per-launch durations are each kernel's own, not a shared GPU's
        try:
            E.PROFILE = []
            step()
            torch.cuda.synchronize()
            E.PROFILE = []
            step()
            torch.cuda.synchronize()
            recs = E.PROFILE
        finally:
            E.PROFILE = None
            plan.serial_streams = bool(args.serial_streams)
        per = {}
        for name, e0, e1, fl_i, nb, shp in recs:
            d = per.setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl_i
            d[3] += nb
        if args.per_launch:
            with open(args.per_launch, "w") as f:
                f.write("kernel,M,N,K,KH,KW,ms,TFLOPs\n")
                for name, e0, e1, fl_i, nb, shp in recs:
                    ms = e0.elapsed_time(e1)
                    f.write("%s,%d,%d,%d,%d,%d,%.4f,%.1f\n" % ((name,) + tuple(shp) + (ms, fl_i / ms / 1e9)))
        dom = max(per, key=lambda k: per[k][1])  # dominant kernel by device time
        n, t_ms, fl, nb = per[dom]
        all_fl = sum(v[2] for v in per.values())
        all_ms = sum(v[1] for v in per.values())
        ach = fl / (t_ms * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the rocprofv3 PMC passes of this command (tools/collect_traffic.py), when committed
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if args.workload == "backbones256" and B == 64 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if dom.startswith(tj.get("kernel", "~")):  # (the counter file aggregates igemm_split_kernel and its _occ variant)
                traffic = round(tj["hbm_bytes_per_launch"])
        peak = kernel_peak(dom)
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "peak_note": ("algorithmic-FLOP roof of the 3 x f16 split scheme = dense f16 MFMA peak 2500 / 3; executed MFMA rate = %.0f TFLOP/s"
                                  % (3 * ach)) if "split" in dom else "dense f32-input MFMA peak",
                    "algo_bytes_per_launch": round(nb / n), "launches_per_step": n, "avg_launch_ms": round(t_ms / n, 4),
                    "gflop_per_step": round(fl / 1e9, 1), "kernel_ms_per_step": round(t_ms, 3),
                    "all_mfma_kernels": {k: {"launches": v[0], "ms": round(v[1], 3), "tflops": round(v[2] / v[1] / 1e9, 2),
                                             "frac": round(v[2] / v[1] / 1e9 / kernel_peak(k), 4)} for k, v in per.items()},
                    "all_mfma_tflops": round(all_fl / all_ms / 1e9, 2),
                    "whole_step_tflops": round(all_fl / (ms_per_step * 1e-3) / 1e12, 2)}

    # ---- reference timing of the strict f32-input MFMA arithmetic (outside the timed region; rank 0, N=1) ----
    f32_ref = None
    if rank == 0 and world == 1 and E.GEMM_MODE != "f32" and not args.no_f32_reference:
        mode = E.GEMM_MODE
        try:
            E.GEMM_MODE = "f32"
            model._plans.clear()
            model._plan(dev).serial_streams = bool(args.serial_streams)
            for _ in range(max(2, args.warmup)):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt32 = time.perf_counter() - t1
            f32_ref = {"value": round(B * args.steps / dt32, 2), "unit": "img/s", "ms_per_step": round(dt32 / args.steps * 1e3, 3),
                       "gemm_arithmetic": "v_mfma_f32_16x16x4_f32 (KPF_GEMM=f32)", "launch": "eager"}
            # the same instrumented pass for this arithmetic: dominant kernel against the dense f32-input MFMA peak
            plan32 = model._plan(dev)
            plan32.serial_streams = True
            try:
                E.PROFILE = []
                step()
                torch.cuda.synchronize()
                E.PROFILE = []
                step()
                torch.cuda.synchronize()
                recs32 = E.PROFILE
            finally:
                E.PROFILE = None
            per32 = {}
            for name, e0, e1, fl_i, nb, shp in recs32:
                d = per32.setdefault(name, [0, 0.0, 0.0])
                d[0] += 1
                d[1] += e0.elapsed_time(e1)
                d[2] += fl_i
            dom32 = max(per32, key=lambda k: per32[k][1])
            a32 = per32[dom32][2] / (per32[dom32][1] * 1e-3) / 1e12
            f32_ref["roofline"] = {"bound": "mfma", "kernel": dom32, "achieved": round(a32, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(a32 / PEAK_F32_MFMA_TFLOPS, 4), "launches_per_step": per32[dom32][0],
                                   "kernel_ms_per_step": round(per32[dom32][1], 3)}
        finally:
            E.GEMM_MODE = mode
            model._plans.clear()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import kpf_oracle as O
        n = min(args.cpu_sample, B)
        cb = {k: torch.from_numpy(v[:n]) for k, v in hb.items()}
        threads = torch.get_num_threads()

        def cpu_step():
            if args.workload == "backbones256":
                O.backbones_forward(sd, cb["img_rgb"], cb["img"])
            else:
                O.kpfusion_forward(sd, cb["img_rgb"], cb["img"], cb["pcl"], cb["center"], cb["M"], cb["cube"], cb["cam_para"], 0.8)

        cpu_step()  # warm-up (oneDNN primitive caches)
        reps, t1 = 0, time.perf_counter()
        while True:
            cpu_step()
            reps += 1
            if time.perf_counter() - t1 > 10.0 or reps >= 5:
                break
        ct = time.perf_counter() - t1
        cpu = {"value": round(n * reps / ct, 2), "unit": "img/s", "cores": threads, "kind": "port",
               "sample": "%d x %d images of the same synthetic batch, oracle/kpf_oracle.py (torch-CPU fp32), %d threads" % (reps, n, threads)}

    if rank == 0:
        line = {
            "metric": "RGB-D img/sec fwd (B=64, 256x256)" if args.workload == "backbones256" else "RGB-D img/sec fwd full model (B=%d, 128x128)" % B,
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if E.GEMM_MODE == "f32" else "f32 (GEMM products as 3 x f16 split MFMA, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "KPFusion-convnext-tiny, depth+RGB UNet backbones forward, B=%d/GPU %dx%d fp32 (BASELINE configs[1])" % (B, S, S)
                       if args.workload == "backbones256" else "KPFusion-convnext-tiny full forward, B=%d/GPU 128x128 fp32" % B,
                       "batch_per_gpu": B, "global_batch": B * world, "input": "%dx%d" % (S, S), "parallelism": "dp%d (batch shards, no collective)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "f32_mfma": f32_ref, "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 3),
            "launch": launch_mode,
        }
        line["config"]["gemm_arithmetic"] = ("fp32 emulation on the f16 matrix cores: operands split into f16 hi+lo (22 bits), 3 MFMAs per product, "
                                             "fp32 accumulate; error vs fp64 <= the f32-input MFMA path's (tests/test_parity_gpu.py)"
                                             if E.GEMM_MODE == "split" else "f32-input MFMA")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
