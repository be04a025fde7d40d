"""hipGraph replay soundness check.

The training iteration is replayed from a captured hipGraph (training.GraphedTrainStep) that contains, besides the product's kernels,
torch reductions.  ATen's multi-block reductions zero a semaphore with an 8-byte hipMemsetAsync in front of the kernel; on the ROCm 7
runtime of the MI355X boxes those memset nodes are mis-ordered from the SECOND replay of a graph on when the runtime's packet-capture
fast path is active (DEBUG_CLR_GRAPH_PACKET_CAPTURE unset or 1): the reduction then returns stale data — silently (found as wrong
layer-scale gradients in the captured training iteration; tools/replay_determinism.py).  The product's own kernels need
no memset nodes; the torch ops around them in the training step do.

`prepare_training_graphs()` sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 — a PROCESS-WIDE setting of the HIP runtime, read when the runtime
initialises, so it only helps before the process's first HIP call.  It is made only on the training side: importing
`keypointfusion_amd.training` calls it (a training script imports that module at its top, before touching the GPU) and so does
`GraphedTrainStep.__init__`; the inference path (model, engine, serving) never does — its captured forwards contain no ATen reduction.
`assert_replay_is_sound()` MEASURES the behaviour once per process and refuses to hand out graphed training steps on a runtime that
replays them wrongly (no silent wrong gradients), whatever the environment says."""
import os
import threading

import torch

_lock = threading.Lock()
_checked = {}
ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def prepare_training_graphs():
    """Switch the runtime's graph packet-capture fast path off for this process unless the host application has chosen a value itself.
    Returns the value in force.  KPF_KEEP_HIP_ENV=1 makes this a no-op (the canary in `assert_replay_is_sound` still decides)."""
    if os.environ.get("KPF_KEEP_HIP_ENV", "0") != "1":
        os.environ.setdefault(ENV, "0")
    return os.environ.get(ENV)


def replay_is_sound(device=None):
    """Capture [y = (a * b).sum(0) over 4096 x 192 (a multi-block reduction: memset node + kernel), z = y * 2] in a hipGraph, replay it
    four times with different inputs and compare every replay with the eager result."""
    dev = torch.device(device if device is not None else torch.cuda.current_device())
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index
    with _lock:
        if key in _checked:
            return _checked[key]
        with torch.cuda.device(dev), torch.no_grad():
            g = torch.Generator(device=dev).manual_seed(1234)
            a = torch.randn(16, 4096, 192, device=dev, generator=g)
            b = torch.randn(16, 4096, 192, device=dev, generator=g)
            sa, sb = a[0].clone(), b[0].clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                outs = [(sa * sb).sum(0) * 2 for _ in range(2)]  # warm-up (lazy initialisation outside the capture)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            del outs
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                # several multi-block reductions, as in a backward pass; between them small temporaries dirty the block the previous
                # semaphore occupied (the caching allocator hands the same 512-byte block out again), so every reduction depends on
                # its own memset node having run, in order
                acc = torch.zeros(192, device=dev)
                for i in range(4):
                    part = (sa[i * 1024:(i + 1) * 1024] * sb[i * 1024:(i + 1) * 1024]).sum(0)
                    dirt = [torch.full((2,), 3.0e38, device=dev) for _ in range(3)]
                    acc = acc + part + dirt[0][0] * 0
                    del dirt, part
                z = (sa * sb).sum(0) * 2 + acc * 0
            ok = True
            for i in range(1, 6):
                sa.copy_(a[i])
                sb.copy_(b[i])
                graph.replay()
                torch.cuda.synchronize()
                ref = (a[i] * b[i]).sum(0) * 2
                ok = ok and bool(torch.equal(z, ref))
            del graph
        _checked[key] = ok
        return ok


def assert_replay_is_sound(device=None):
    if not replay_is_sound(device):
        raise RuntimeError(
            "this HIP runtime replays captured graphs with torch reductions wrongly from the second replay on (memset nodes under the graph "
            "packet-capture fast path); DEBUG_CLR_GRAPH_PACKET_CAPTURE is %r in this process — export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before "
            "the process makes its first HIP call (importing keypointfusion_amd.training before touching the GPU does it)"
            % os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
