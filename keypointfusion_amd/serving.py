"""Throughput-oriented evaluation / serving loop on one GPU: several batches in flight.

One eval forward is two phases with opposite shapes: the backbones fill the chip (dense GEMMs over B*H*W pixels), the keypoint-fusion
head is a chain of per-sample, latency-bound kernels (21-token transformer stacks, ball queries, gathers: one workgroup per sample, so
B of the 256 CUs are busy).  Batches are independent in eval, so the head of batch i can run beside the backbones of batch
i+1: `PipelinedEval` keeps `depth` independent captured hipGraphs of the full forward (own static buffers and workspace pool each,
shared read-only weights), each replayed on its own HIP stream.  The reference's eval loop (train.py:330-362) is synchronous per batch
— this is the MI355X-side form of the same loop for callers that only need the outputs later (metrics accumulated on the device,
serving queues).

    pe = PipelinedEval(model, depth=2)
    tickets = [pe.submit(img_rgb, img, pcl, loader, center, M, cube, cam_para) for ... in batches]   # returns immediately
    results, spatial_weights, _ = pe.collect(ticket)      # makes the caller's stream wait for that batch only
"""
import torch


class PipelinedEval:
    def __init__(self, model, depth=2):
        if depth < 1:
            raise ValueError("PipelinedEval: depth must be >= 1")
        self.model, self.depth = model, int(depth)
        self._streams = None
        self._next = 0

    def submit(self, img_rgb, img, pcl, loader, center, M, cube, cam_para, kernel=0.8):
        """Enqueue one batch (copy-in, graph replay, copy-out on the next slot's stream); returns a ticket for collect()."""
        m = self.model
        if m.training:
            raise RuntimeError("PipelinedEval is an eval-mode loop: call model.eval() first")
        m._require_gpu(img)
        if img.shape[-1] != m.crop_size:
            raise RuntimeError("PipelinedEval: this model was built for %dx%d crops (got %d)" % (m.crop_size, m.crop_size, img.shape[-1]))
        dev = img.device
        plan = m._plan(dev)
        if self._streams is None or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        slot = self._next
        self._next = (self._next + 1) % self.depth
        st = self._streams[slot]
        st.wait_stream(torch.cuda.current_stream(dev))  # the inputs were produced on the caller's stream
        with torch.no_grad(), torch.cuda.device(dev), torch.cuda.stream(st):
            res, sws, _ = plan.forward_graphed(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), int(getattr(loader, "img_size", m.crop_size)),
                                               int(getattr(loader, "flip", 1)), slot=slot)
            ev = torch.cuda.Event()
            ev.record(st)
        return (res, sws, ev, st)

    def collect(self, ticket):
        """(list of 6 results, list of 2 spatial weights, None) of a submitted batch, ordered after it on the caller's stream."""
        res, sws, ev, st = ticket
        cur = torch.cuda.current_stream(res[0].device)
        cur.wait_event(ev)
        for t in res + sws:
            t.record_stream(cur)  # allocated on the slot's stream, consumed on the caller's
        return res, sws, None
