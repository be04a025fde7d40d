#!/usr/bin/env python3
"""Benchmark of the KPFusion forward hot path on MI355X (driver contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched once per rank by torch.distributed.run (the driver's form; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment) or plainly as `python bench.py --gpus N`, in which case this process starts the N ranks itself as a child torchrun job
(before any HIP call) and relays their one JSON line and exit code.

Workload = BASELINE.json configs[1]: B=64 synthetic 256x256 RGB-D crops per GPU, KPFusion-convnext-tiny, both UNet
backbones (depth + RGB) forward-only, fp32, eval.  (The fusion head only exists at 128x128 in the reference — SURVEY D1
— so configs[1] is "backbones only"; `--workload full128` times the whole model at B=64, 128x128 instead.)
A "step" is one forward over one batch already resident in HBM.  N GPUs = N independent shards of the batch
dimension (weak scaling, no data-path collective: every sample is independent in eval).

Arithmetic of the headline (`value`, `dtype: "f32"`): IEEE fp32 everywhere — every GEMM on v_mfma_f32_16x16x4_f32 (bit for bit an
fmaf chain), fp32 storage and elementwise math — replayed from a captured hipGraph (both backbone streams as parallel branches).

One JSON line is printed by rank 0 with, besides the contract fields:
  roofline     : the dominant kernel of the headline run (igemm_f32_kernel) — achieved = algorithmic conv/GEMM FLOPs (2*M*N*K) of all its
                 launches in one step / summed device time of those launches (HIP events recorded on the launch stream in an
                 instrumented pass after the timed region, both backbones on ONE stream), against the dense f32-input MFMA peak
                 157.3 TFLOP/s.  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of this same command, collected
                 offline and committed (profiles/r06_traffic*.json, tools/collect_traffic.py): `traffic_source` says so; a figure
                 below 0.9 x the algorithmic bytes, or one whose launch count is not a whole number of this run's steps, is refused
                 (`traffic: null` + `traffic_rejected`).
  split_f16x3  : SECONDARY record, not the headline and not IEEE fp32: the same workload with KPF_GEMM=split (the ConvNeXt-block GEMMs
                 as 3 x f16 MFMA on hi/lo-split operands with a pack-time range proof; everything else stays on the f32 MFMA).
  extra        : (default run, N = 1) one timed record per further BASELINE config that fits one GPU — `cnb512_f16` (configs[4]: ConvNeXt-B,
                 512x512, f16, B=64), `full128_bf16` (configs[2]: full model, B=32, bf16), `train128_bf16` (configs[3]: one training
                 iteration, B=32, bf16), `full128_f16` (configs[2] in the recommended 16-bit mode) — and `full256` (the FULL model at the headline's batch and crop size: the labelled wide extension,
                 not a reference configuration), each measured by this script as a child process on that workload, with its own `roofline`.
  world_size   : ranks that took part (dist.get_world_size()), `collective_backend` the RCCL version when a process group exists.
  cpu_baseline : the CPU oracle (oracle/kpf_oracle.py, torch-CPU fp32 = the reference's own arithmetic) on the host
                 cores of this box: thread-count sweep, then 5 timed passes at the best count over a bounded sample of the same
                 workload — `value` is their median, the best pass is stated beside it (rank 0, N=1 only).
"""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # HIP-runtime graph-replay workaround (keypointfusion_amd/graphs.py): before the first HIP call
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 matrix peak
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3  # 3 f16 MFMAs per algorithmic fp32 product (hi*hi + hi*lo + lo*hi)
# workload -> (net, crop size, default batch per GPU, storage precision, "backbones" | "full", BASELINE.json config it measures)
WORKLOADS = {
    "backbones256": ("KPFusion-convnext-tiny", 256, 64, "f32", "backbones", "configs[1]"),
    "full128": ("KPFusion-convnext-tiny", 128, 64, "f32", "full", "full model at configs[1]'s batch"),
    "full128_bf16": ("KPFusion-convnext-tiny", 128, 32, "bf16", "full", "configs[2]"),
    # the same in f16 — the reduced-precision mode the accuracy statistics recommend (f16 deviates 0.2 mm from the fp32 path, bf16 0.9 mm)
    "full128_f16": ("KPFusion-convnext-tiny", 128, 32, "f16", "full", "configs[2] in f16 (the recommended 16-bit mode)"),
    "full128_bf16_r18": ("KPFusion-resnet-18", 128, 32, "bf16", "full", "configs[2] with the ResNet-18 backbones (SURVEY 8d config 3 names both families)"),
    # the full model at the crop size BASELINE's metric is quoted on: the labelled wide extension (KPFusion(..., crop_size=256): fc_spatial2joint_feature sized
    # for the 64 x 64 feature map; the reference hard-codes 32 x 32 and cannot run this size) — not a reference configuration
    "full256": ("KPFusion-convnext-tiny", 256, 64, "f32", "full", "full model at configs[1]'s batch and crop size: wide extension, not a reference configuration"),
    "cnb512_f16": ("KPFusion-convnext-base", 512, 64, "f16", "backbones", "configs[4]"),
    # one training iteration of train.py:209-265 (forward in train mode, loss, backward, gradient all-reduce over RCCL when N > 1, AdamW
    # step); fp32 — bf16 training is not built, so this is configs[3]'s schedule, not its precision
    "train128": ("KPFusion-convnext-tiny", 128, 32, "f32", "train", "configs[3] (fp32)"),
    "train128_bf16": ("KPFusion-convnext-tiny", 128, 32, "bf16", "train", "configs[3]"),
    "train128_f16": ("KPFusion-convnext-tiny", 128, 32, "f16", "train", "configs[3] in fp16 with dynamic loss scaling (training.LossScaler)"),
}


# committed PMC traffic figures (tools/profile_round.sh -> tools/collect_traffic.py), per workload; quoted only for the stated batch
TRAFFIC_FILES = {"backbones256": "r06_traffic.json", "cnb512_f16": "r06_traffic_cnb512.json", "full128_bf16": "r06_traffic_full128_bf16.json"}
NO_TRAFFIC_REASON = {"train128_bf16": "rocprofv3 --pmc does not complete on the training iteration on this pool (segmentation fault in the FETCH_SIZE pass, no return from "
                                      "the WRITE_SIZE pass within 40 min: tools/profile_round.sh, round 5)"}


def under_profiler():
    """True when a rocprofiler tool library is preloaded into this process (children would inherit it and write their own trace files
    into the same output directory: VERDICT r04)."""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY"))


def kernel_peak(name):
    if "h16" in name or "gemm16" in name:
        return PEAK_F16_MFMA_TFLOPS
    return PEAK_SPLIT_TFLOPS if "split" in name else PEAK_F32_MFMA_TFLOPS


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_command(n, argv, port=None):
    """The child job `python bench.py --gpus n` starts: the driver's own form (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def launch_env(environ):
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it across processes
    return env


def self_launch(n):
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child, relay its output, return its rc."""
    import subprocess
    cmd = launch_command(n, sys.argv[1:])
    print("bench.py: launching %d rank(s): %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=launch_env(os.environ), stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


# Secondary workloads of the default run (N = 1): each is this same script as a child process on its own workload — its own model,
# capture and `roofline` — after the headline has been timed, so the driver's one invocation carries a timed record for every
# BASELINE config that runs on one GPU.  (workload, steps, warmup)
# (order: the launch-bound workloads first, the two that hold the chip at its power limit last — on some boxes the training iteration measured 10 % slower
#  right behind the ConvNeXt-B run than on its own: 17.6 vs 15.8 ms, profiles/r05_bench.json of the first r05 collection)
EXTRA_WORKLOADS = (("train128_bf16", 30, 5), ("full128_bf16", 30, 5), ("full128_f16", 30, 5), ("full256", 8, 3), ("cnb512_f16", 6, 2))


def run_extra(workload, steps, warmup, timeout):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-extra"]
    t0 = time.perf_counter()
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout}
    rec = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{"):
            try:
                rec = json.loads(ln)
            except ValueError:
                pass
    if proc.returncode != 0 or rec is None:
        return {"error": "rc %d: %s" % (proc.returncode, proc.stderr[-400:])}
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "launch", "roofline", "world_size", "single_batch_latency", "single_step_into_idle_gpu")
    out = {k: rec.get(k) for k in keep}
    out["workload"] = rec["config"]["workload"]
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def collective_probe(dist, dev, world, gstep, iters=5):
    """EVERY rank calls this (it issues collectives).  HIP events around RCCL calls, outside the timed region, so that one SCALE line says what the
    fabric gave: (a) the training step's own gradient buckets, one at a time, on its static buffers (`buckets`: bytes, median ms, algorithm and bus
    bandwidth, bus = alg x 2 (n - 1) / n for an all-reduce: the per-link figure a ring is bound by); (b) for any workload a fixed 64 MiB fp32 all-reduce
    (`probe_64MiB`: the bucket size the training step uses), so that the eval workloads' N > 1 lines — which have no data-path collective — still carry one
    measured point of this node's xGMI.  Compare with DESIGN.md section 6's prediction (268 MB of fp32 gradients per iteration, per-link ring bound)."""
    import statistics as st

    def time_one(fn, nbytes):
        ts = []
        for _ in range(iters + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.barrier()
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = st.median(ts[1:])  # (first call: channel setup)
        alg = nbytes / (ms * 1e-3) / 1e9
        return {"bytes": int(nbytes), "ms": round(ms, 4), "algbw_GBps": round(alg, 1), "busbw_GBps": round(alg * 2 * (world - 1) / max(world, 1), 1)}

    out = {"world_size": world, "note": "HIP events around eager RCCL calls after the timed region, median of %d (every rank takes part); busbw = algbw x 2(n-1)/n" % iters}
    probe = torch.zeros(64 * 1024 * 1024 // 4, device=dev, dtype=torch.float32)
    out["probe_64MiB"] = time_one(lambda: dist.all_reduce(probe), probe.numel() * 4)
    if gstep is not None and getattr(gstep, "buckets", None):
        out["buckets"] = gstep.time_collectives(time_one)
        tot_b = sum(b["bytes"] for b in out["buckets"])
        tot_ms = sum(b["ms"] for b in out["buckets"])
        out["all_buckets"] = {"bytes": tot_b, "ms_back_to_back": round(tot_ms, 3), "algbw_GBps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1) if tot_ms > 0 else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="batch per GPU (default: the workload's)")
    ap.add_argument("--workload", default="backbones256", choices=sorted(WORKLOADS),
                    help="backbones256 = BASELINE configs[1] (the headline); full128_bf16 = configs[2]; cnb512_f16 = configs[4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-streams", action="store_true", help="issue both backbones on one stream (per-kernel profiling: rocprofv3 "
                    "durations of overlapped kernels are otherwise shared-GPU durations)")
    ap.add_argument("--stage-pipeline", action="store_true", help="full-model eval workloads: two graphs per batch (backbones of batch i + 1 on one stream beside "
                    "the head of batch i on another) instead of whole forwards on --in-flight streams")
    ap.add_argument("--one-stream-graph", action="store_true", help="eval workloads: capture both backbones on ONE stream (the graph stays on; a single-stream "
                    "graph replays on the runtime's batched path, a forked one node by node)")
    ap.add_argument("--no-graph", action="store_true", help="issue the launches of a step from Python instead of replaying the captured "
                    "hipGraph (the default; same device time, but the step then depends on the host keeping up)")
    ap.add_argument("--per-launch", default="", help="write a per-launch table of the MFMA kernels to this file")
    ap.add_argument("--cpu-sample", type=int, default=16, help="images in the CPU-baseline sample")
    ap.add_argument("--gemm", default="f32", choices=["f32", "split"], help="GEMM arithmetic of the headline run (default f32 = IEEE fp32)")
    ap.add_argument("--no-split-record", action="store_true", help="skip the secondary KPF_GEMM=split timing")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workload records (cnb512_f16, full128_bf16, train128_bf16) "
                    "that the default run adds under `extra`")
    ap.add_argument("--in-flight", type=int, default=2, help="full-model eval workloads: batches in flight (serving.PipelinedEval: independent "
                    "hipGraph slots on their own streams, the latency-bound fusion head of one batch beside the backbones of the next); "
                    "1 = one synchronous forward per step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("KPF_BENCH_FORCE_DIST")):
        # `python bench.py --gpus N` by itself: become the launcher.  No HIP call has been made yet (importing torch and parsing
        # arguments do not initialise the runtime), so the ranks are started as CHILD processes of this one (never an exec) — one
        # per GPU over RCCL — their output is relayed and their exit code returned.  Replaces train.py:81 / demo_RGBD.py:49
        # (`DataParallel(net).cuda()`: one command, all GPUs).  KPF_BENCH_FORCE_DIST=1 takes the same route with --gpus 1 (a one-rank
        # RCCL group on a one-GPU box).
        return self_launch(args.gpus)
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1 or os.environ.get("KPF_BENCH_FORCE_DIST"):  # (the env switch exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from keypointfusion_amd import engine as E, lib as L
    from keypointfusion_amd.model.model import KPFusion
    from keypointfusion_amd.weights import synthetic_batch, synthetic_state_dict
    L.load()
    E.GEMM_MODE = args.gemm

    NET, S, B0, precision, kind, cfg_name = WORKLOADS[args.workload]
    B = args.batch or B0
    backbones_only = kind == "backbones"
    train = kind == "train"
    if train:
        args.no_split_record = True
    if precision != "f32":
        args.no_split_record = True  # (the split record belongs to the fp32 workloads)
        if args.workload == "cnb512_f16":
            args.cpu_sample = min(args.cpu_sample, 1)  # 373 GFLOP per image on the CPU
    crop = 128 if backbones_only else S  # (backbones-only workloads run any size through forward_backbones: the reference-sized module)
    model = KPFusion(NET, "", 21, "dexycb", "", crop_size=crop)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(NET, 0, crop_size=crop).items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.precision = precision
    hb = synthetic_batch(B, S, seed=1 + rank)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in hb.items()}
    reducer = opt = None
    gstep = [None]
    if train:
        from keypointfusion_amd import training as T
        from keypointfusion_amd.parallel import GradBucketReducer, live_parameters
        model.train()
        gg = torch.Generator().manual_seed(100 + rank)
        uvd_gt = (torch.rand(B, 21, 3, generator=gg) * 1.2 - 0.6).to(dev)
        xyz_gt = (torch.rand(B, 21, 3, generator=gg) * 1.2 - 0.6).to(dev)
        live = live_parameters(model)
        graphed_train = not args.no_graph and not args.serial_streams  # the iteration replays from hipGraphs (N > 1: bucketed all-reduce between two graphs)
        scaler = T.LossScaler() if precision == "f16" else None  # fp16 gradients underflow without it; bf16 / fp32 need none
        opt, _ = T.make_optimizer(live, capturable=graphed_train or scaler is not None)
        reducer = GradBucketReducer(live, dist if dist is not None else None) if not graphed_train else None
        tbatch = dict(batch, uvd_gt=uvd_gt, xyz_gt=xyz_gt)

        def train_loss(mdl, bt):
            results, sws, _ = mdl(bt["img_rgb"], bt["img"], bt["pcl"], _Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
            return T.kpfusion_loss(results, sws, bt["img"], bt["uvd_gt"], bt["xyz_gt"], epoch=0)[0]

    class _Loader:
        img_size, flip = (128 if backbones_only else S), 1

    graph_on = [not args.no_graph and not args.serial_streams]
    pipe, pending = [None], []

    def step():
        if train:
            if graphed_train and graph_on[0]:
                if gstep[0] is None:
                    gstep[0] = T.GraphedTrainStep(model, opt, train_loss, tbatch, dist_mod=dist, params=live, scaler=scaler)
                gstep[0](tbatch)
                return
            opt.zero_grad(set_to_none=False)
            if reducer is not None:
                reducer.reset()
            loss = train_loss(model, tbatch)
            (loss if scaler is None else scaler.scale(loss)).backward()
            if reducer is not None:
                reducer.finish()
            if scaler is None:
                opt.step()
            else:
                opt.step(scaler=scaler)
            return
        with torch.no_grad():
            if backbones_only:
                if graph_on[0]:
                    model._plan(dev).backbones_graphed(batch["img"], batch["img_rgb"])
                else:
                    model._plan(dev).backbones(batch["img"], batch["img_rgb"])
            elif graph_on[0] and args.in_flight > 1:
                if pipe[0] is None:
                    from keypointfusion_amd.serving import PipelinedEval
                    pipe[0] = PipelinedEval(model, depth=args.in_flight, stages=args.stage_pipeline)
                pending.append(pipe[0].submit(batch["img_rgb"], batch["img"], batch["pcl"], _Loader(), batch["center"], batch["M"], batch["cube"],
                                              batch["cam_para"], 0.8))
                if len(pending) > args.in_flight:  # a real loop consumes the oldest batch's outputs here
                    pipe[0].collect(pending.pop(0))
            else:
                model.use_graphs = graph_on[0]
                model(batch["img_rgb"], batch["img"], batch["pcl"], _Loader(), batch["center"], batch["M"], batch["cube"],
                      batch["cam_para"], 0.8)

    def barrier():
        while pending:  # (pipelined eval: every submitted batch is collected inside the timed region)
            pipe[0].collect(pending.pop(0))
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def fresh_plan():
        model._plans.clear()
        pipe[0] = None
        del pending[:]
        if not train:
            model._plan(dev).serial_streams = bool(args.serial_streams or args.one_stream_graph)

    # The pipelined eval loop submits from a stream of its own: an event recorded on the DEFAULT stream (what `stream.wait_stream(default)` does for every
    # submitted batch) acts as a join over every stream of the device on this runtime and serialises the pipeline (3.35 vs 2.11 ms per B = 32 batch,
    # tools/exp_overlap3.py); serving.PipelinedEval documents the same for its callers.
    loop_stream = torch.cuda.Stream(device=dev) if (not train and not backbones_only and graph_on[0] and args.in_flight > 1) else None
    if loop_stream is not None:
        loop_stream.wait_stream(torch.cuda.current_stream(dev))  # the synthetic batch was produced on the default stream

    def on_loop_stream():
        return torch.cuda.stream(loop_stream) if loop_stream is not None else contextlib.nullcontext()

    def timed(K, W):
        """W untimed steps, then exactly K steps between barrier + synchronize pairs; returns (seconds, host issue seconds)."""
        with on_loop_stream():
            for _ in range(W):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(K):
                step()
            t_issue = time.perf_counter() - t0  # host time to enqueue the K steps (close to dt => the step is launch-bound)
            barrier()
            return time.perf_counter() - t0, t_issue

    def instrumented():
        """Per-launch HIP events around every MFMA-kernel launch, both backbones on one stream (each duration is the kernel's own)."""
        nonlocal reducer
        plan = model._plan(dev) if not train else type("P", (), {"serial_streams": False})()
        g = graph_on[0]
        graph_on[0] = False
        plan.serial_streams = True
        # rank 0 alone runs this pass: an eager data-parallel step here would issue bucket all-reduces no other rank joins (they are already in the
        # final barrier) — a mismatched RCCL sequence, i.e. a hang.  The per-launch kernel times do not need the collective.
        keep_reducer, reducer = reducer, (None if dist is not None else reducer)
        try:
            recs = None
            for _ in range(2):  # the second pass is the one kept (first: warm caches, lazily packed weights)
                E.PROFILE = []
                step()
                torch.cuda.synchronize()
                recs = E.PROFILE
        finally:
            E.PROFILE = None
            plan.serial_streams = bool(args.serial_streams or args.one_stream_graph)
            graph_on[0] = g
            reducer = keep_reducer
        return recs

    def roofline_of(recs, ms_per_step, traffic_file):
        per = {}
        for name, e0, e1, fl_i, nb, shp in recs:
            d = per.setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl_i
            d[3] += nb
        dom = max(per, key=lambda k: per[k][1])  # dominant kernel by device time
        n, t_ms, fl, nb = per[dom]
        all_fl = sum(v[2] for v in per.values())
        all_ms = sum(v[1] for v in per.values())
        ach = fl / (t_ms * 1e-3) / 1e12
        traffic, tsrc, trej = None, None, NO_TRAFFIC_REASON.get(args.workload)
        tpath = os.path.join(ROOT, "profiles", traffic_file) if traffic_file else None
        if tpath and B == B0 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            t_b, t_n = tj["hbm_bytes_per_launch"], int(tj.get("launches_profiled", 0))
            if not dom.startswith(tj.get("kernel", "~")):
                trej = "%s describes %s, the dominant kernel here is %s" % (traffic_file, tj.get("kernel"), dom)
            elif t_n == 0 or t_n % n:
                trej = "%s averages %d launches, not a whole number of this run's %d-launch steps" % (traffic_file, t_n, n)
            elif t_b < 0.9 * nb / n:
                trej = "%s: %.1f MB per launch is below the algorithmic %.1f MB (a kernel cannot move less than its compulsory bytes)" % (
                    traffic_file, t_b / 1e6, nb / n / 1e6)
            else:
                traffic = round(t_b)
                tsrc = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (%d launches = %d steps), collected offline "
                        "(tools/collect_traffic.py; FETCH_SIZE doubled per the gfx950 note for 16-B/lane reads)" % (traffic_file, t_n, t_n // n))
        peak = kernel_peak(dom)
        return {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc, "traffic_rejected": trej,
                "peak_note": ("algorithmic-FLOP roof of the 3 x f16 split scheme = dense f16 MFMA peak 2500 / 3; executed MFMA rate = %.0f TFLOP/s"
                              % (3 * ach)) if "split" in dom else ("dense 16-bit MFMA peak (v_mfma_f32_16x16x32_bf16 / _f16)" if ("h16" in dom or "gemm16" in dom)
                                                                    else "dense f32-input MFMA peak (v_mfma_f32_16x16x4_f32)"),
                "algo_bytes_per_launch": round(nb / n), "algo_hbm_gbps": round(nb / (t_ms * 1e-3) / 1e9, 1), "hbm_frac_of_8000": round(nb / (t_ms * 1e-3) / 8e12, 4),
                "launches_per_step": n, "avg_launch_ms": round(t_ms / n, 4),
                "gflop_per_step": round(fl / 1e9, 1), "kernel_ms_per_step": round(t_ms, 3),
                "all_mfma_kernels": {k: {"launches": v[0], "ms": round(v[1], 3), "tflops": round(v[2] / v[1] / 1e9, 2),
                                         "frac": round(v[2] / v[1] / 1e9 / kernel_peak(k), 4)} for k, v in per.items()},
                "all_mfma_tflops": round(all_fl / all_ms / 1e9, 2),
                "whole_step_tflops": round(all_fl / (ms_per_step * 1e-3) / 1e12, 2)}

    # ---- headline: capture during warm-up; if the runtime refuses the capture, time the eager launches instead of failing ----
    fresh_plan()
    if graph_on[0]:
        try:
            with on_loop_stream():
                step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print("bench.py: hipGraph capture failed (%s: %s); falling back to eager launches" % (type(e).__name__, e), file=sys.stderr)
            graph_on[0] = False
            fresh_plan()
    launch_mode = "hipGraph replay" if graph_on[0] else "eager"
    if graph_on[0] and not train and not backbones_only and args.in_flight > 1:
        launch_mode += (", %d batches in flight (independent graph slots / streams; the loop submits from its own stream)" % args.in_flight) if not args.stage_pipeline else (
            ", stage pipeline over %d buffer sets (backbone graphs on one stream beside the head graphs on another)" % args.in_flight)
    dt, t_issue = timed(args.steps, args.warmup)
    single = None
    if train:  # one step issued into an idle GPU: host time of the enqueue vs the step's whole duration (which of the two bounds a replay)
        iss, tot = [], []
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            step()
            t1 = time.perf_counter()
            barrier()
            iss.append(t1 - t0)
            tot.append(time.perf_counter() - t0)
        single = {"issue_ms": round(min(iss) * 1e3, 3), "total_ms": round(min(tot) * 1e3, 3)}
    latency = None
    if not train and not backbones_only and rank == 0:
        # the same forward ONE batch at a time (module forward, hipGraph replay when graphs are on): the latency a caller of model(...) sees, beside the
        # throughput of the pipelined loop above; and the library entry points one eager forward calls (a handful issue two kernels)
        model.use_graphs = graph_on[0]
        fwd = lambda: model(batch["img_rgb"], batch["img"], batch["pcl"], _Loader(), batch["center"], batch["M"], batch["cube"], batch["cam_para"], 0.8)
        with torch.no_grad():
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                fwd()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            model.use_graphs = False
            c0 = L.CALLS[0]
            fwd()
            torch.cuda.synchronize()
            calls = L.CALLS[0] - c0
            model.use_graphs = graph_on[0]
        latency = {"median_ms": round(statistics.median(ts), 3), "min_ms": round(min(ts), 3), "img_per_s_one_batch_at_a_time": round(B / statistics.median(ts) * 1e3, 1),
                   "library_calls_per_forward": calls,
                   "note": "model(...) on one batch, synchronised after every call (%s); `value` above is the pipelined loop" % ("hipGraph replay" if graph_on[0] else "eager")}
    per_rank = collectives = None
    if dist is not None:
        # every rank's own time for the K steps (the line's `ms_per_step` is their MAX): a straggler GPU, or ranks waiting on each other inside a
        # collective, show up here instead of hiding inside one number
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ms = [float(t.item()) / args.steps * 1e3 for t in every]
        per_rank = {"min": round(min(ms), 3), "max": round(max(ms), 3), "spread_pct": round((max(ms) / min(ms) - 1) * 100, 2), "all": [round(v, 3) for v in ms]}
        dt = max(float(t.item()) for t in every)
        collectives = collective_probe(dist, dev, world, gstep[0] if train else None)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    roofline = None
    if rank == 0:
        recs = instrumented()
        if args.per_launch:
            with open(args.per_launch, "w") as f:
                f.write("kernel,M,N,K,KH,KW,ms,TFLOPs\n")
                for name, e0, e1, fl_i, nb, shp in recs:
                    ms = e0.elapsed_time(e1)
                    f.write("%s,%d,%d,%d,%d,%d,%.4f,%.1f\n" % ((name,) + tuple(shp) + (ms, fl_i / ms / 1e9)))
        roofline = roofline_of(recs, ms_per_step, TRAFFIC_FILES.get(args.workload) if args.gemm == "f32" else None)

    # ---- secondary record: split (3 x f16) arithmetic where a range proof exists; NOT the headline, NOT IEEE fp32 ----
    split_rec = None
    if rank == 0 and world == 1 and args.gemm == "f32" and not args.no_split_record:
        try:
            E.GEMM_MODE = "split"
            fresh_plan()
            with on_loop_stream():
                step()
            torch.cuda.synchronize()
            dts, _ = timed(args.steps, max(2, args.warmup))
            split_rec = {"value": round(B * args.steps / dts, 2), "unit": "img/s", "ms_per_step": round(dts / args.steps * 1e3, 3), "launch": launch_mode,
                         "note": "SECONDARY, not IEEE fp32: ConvNeXt-block GEMMs as 3 x v_mfma_f32_16x16x32_f16 on f16 hi+lo operands (22-bit "
                                 "significands, f16 range, operands proven in range at pack time and pre-scaled), fp32 accumulate; all "
                                 "other GEMMs on the f32-input MFMA"}
            split_rec["roofline"] = roofline_of(instrumented(), dts / args.steps * 1e3, None)
        finally:
            E.GEMM_MODE = args.gemm
            fresh_plan()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import kpf_oracle as O
        n = min(args.cpu_sample, B)
        cb = {k: torch.from_numpy(v[:n]) for k, v in hb.items()}
        threads = torch.get_num_threads()

        def cpu_step():
            if train:  # the reference's arithmetic for a training iteration is not restated on the CPU side: forward only, said in `sample`
                with torch.no_grad():
                    O.kpfusion_forward(sd, cb["img_rgb"], cb["img"], cb["pcl"], cb["center"], cb["M"], cb["cube"], cb["cam_para"], 0.8, img_size=S)
            elif backbones_only:
                O.backbones_forward(sd, cb["img_rgb"], cb["img"])
            else:
                O.kpfusion_forward(sd, cb["img_rgb"], cb["img"], cb["pcl"], cb["center"], cb["M"], cb["cube"], cb["cam_para"], 0.8, img_size=S)

        # Thread count: all hardware threads is not the fastest setting for a batch this small (8 images on 128 SMT threads
        # oversubscribes the convolution's work split: round 3 reported 2.66 img/s that way, below an 8-vCPU box), so the pass is timed
        # at several counts and the best one is the baseline; which, and the whole sweep, are stated in `sample`.
        ncpu = os.cpu_count() or 1
        cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu}) or [ncpu]  # (beyond 64 threads the 16-image pass only gets slower: 128 threads 2.3 img/s, 256 0.12 on a 2 x 64-core host)
        if args.workload == "cnb512_f16":
            cands = cands[-2:]
        sweep = {}
        for t in cands:
            torch.set_num_threads(t)
            cpu_step()  # warm-up (oneDNN primitive caches, thread pool)
            t1 = time.perf_counter()
            cpu_step()
            sweep[t] = time.perf_counter() - t1
        best_t = min(sweep, key=sweep.get)
        torch.set_num_threads(best_t)
        cpu_step()  # (the sweep left the pool at another size: one untimed pass at the chosen count)
        times = []
        for _ in range(5):
            t1 = time.perf_counter()
            cpu_step()
            times.append(time.perf_counter() - t1)
        torch.set_num_threads(threads)
        med = statistics.median(times)
        # context-grade figure (the 5 passes span +-15 % on a shared host): two significant digits, with its spread beside it
        cpu = {"value": float("%.2g" % (n / med)), "best": float("%.2g" % (n / min(times))), "worst": float("%.2g" % (n / max(times))),
               "spread_pct": round((max(times) / min(times) - 1) * 100, 1), "unit": "img/s", "cores": best_t, "kind": "port",
               "sample": "median (`value`) and best (`best`) of 5 passes over %d images of the same synthetic batch, oracle/kpf_oracle.py (torch-CPU "
                         "fp32) at the best thread count of a one-pass sweep {%s} img/s; the 5 passes at %d threads: %s img/s%s" % (
                             n, ", ".join("%d: %.2f" % (t, n / sweep[t]) for t in cands), best_t, " ".join("%.2f" % (n / t) for t in times),
                             " (forward only: the reference's training iteration is not restated on the CPU side)" if train else ""),
               "host": "%s, nproc %d" % (cpu_model(), os.cpu_count() or 0)}

    extra = None
    if rank == 0 and world == 1 and dist is None and args.workload == "backbones256" and not args.no_extra and not args.batch and under_profiler():
        print("bench.py: a profiler is preloaded; the secondary workloads are not started as children of a profiled run", file=sys.stderr)
    elif rank == 0 and world == 1 and dist is None and args.workload == "backbones256" and not args.no_extra and not args.batch:
        # the headline's buffers are no longer needed: give the memory back before the children build their own models
        pipe[0] = None
        model._plans.clear()
        torch.cuda.empty_cache()
        extra = {}
        for wl, k, w in EXTRA_WORKLOADS:
            extra[wl] = run_extra(wl, k, w, timeout=420)

    if rank == 0:
        rccl = None
        if dist is not None:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl = "unknown"
        line = {
            "metric": "RGB-D img/sec fwd (B=64, 256x256)" if args.workload == "backbones256" else
                      "RGB-D img/sec trained (fwd + loss + bwd + AdamW, B=%d, %dx%d, %s)" % (B, S, S, precision) if train else
                      "RGB-D img/sec fwd %s (B=%d, %dx%d, %s)" % ("backbones" if backbones_only else "full model", B, S, S, precision),
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (precision + " storage, f32 accumulate") if precision != "f32" else
                     ("f32" if args.gemm == "f32" else "f32 storage, ConvNeXt-block GEMM products as 3 x f16 split MFMA (not IEEE fp32)"), "data": "synthetic",
            "config": {"workload": "%s, %s, B=%d/GPU %dx%d %s (BASELINE %s)" % (
                           NET, "depth+RGB UNet backbones forward" if backbones_only else
                           "full model (backbones + fusion head) training iteration (forward, loss, backward, AdamW)" if train else
                           "full model (backbones + fusion head) forward", B, S, S, precision, cfg_name),
                       "batch_per_gpu": B, "global_batch": B * world, "input": "%dx%d" % (S, S), "parallelism": "dp%d (batch shards, no collective)" % world,
                       "gemm_arithmetic": ("16-bit (%s) operands and activations in HBM, fp32 accumulate on v_mfma_f32_16x16x32_%s; fusion head fp32" % (precision, precision))
                       if precision != "f32" else ("IEEE fp32: v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate) for every GEMM" if args.gemm == "f32"
                                                   else "KPF_GEMM=split")},
            "roofline": roofline, "cpu_baseline": cpu, "split_f16x3": split_rec, "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 3),
            "launch": launch_mode,
            "world_size": dist.get_world_size() if dist is not None else 1,  # ranks that ran (N > 1: one process per GPU over RCCL)
            "collective_backend": ("rccl " + rccl) if dist is not None else None,
        }
        if extra is not None:
            line["extra"] = extra
        if single is not None:
            line["single_step_into_idle_gpu"] = single
        if latency is not None:
            line["single_batch_latency"] = latency
        if per_rank is not None:
            line["per_rank_ms_per_step"] = per_rank
        if collectives is not None:
            line["collectives"] = collectives
        if train and gstep[0] is not None and dist is not None:
            line["dp_graph"] = {"mode": gstep[0].dp_mode, "payload_bytes_per_rank": gstep[0].payload_bytes(), "grad_payload": gstep[0].grad_payload,
                                "collective": gstep[0].collective, "buckets": len(gstep[0].buckets),
                                "prediction": "DESIGN.md section 6: 8 GPUs 16.3 -> ~16.7-17.0 ms per iteration (efficiency ~0.96): the bucket collectives run under backward"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
