"""Throughput-oriented evaluation / serving loop on one GPU: several batches in flight.

One eval forward is two phases with opposite shapes: the backbones fill the chip (dense GEMMs over B*H*W pixels), the keypoint-fusion
head is a chain of per-sample, latency-bound kernels (21-token transformer stacks, ball queries, gathers: one workgroup per sample, so
B of the 256 CUs are busy).  Batches are independent in eval, so the head of batch i can run beside the backbones of batch
i+1: `PipelinedEval` keeps `depth` independent sets of captured hipGraphs of the forward (own static buffers and workspace pool each,
shared read-only weights), each replayed on its own HIP stream (or, `stages=True`, two graphs per batch — backbones and head — with all backbone graphs
on one stream and all head graphs on another).  The reference's eval loop (train.py:330-362) is synchronous per batch
— this is the MI355X-side form of the same loop for callers that only need the outputs later (metrics accumulated on the device,
serving queues).

    pe = PipelinedEval(model, depth=2)
    with torch.cuda.stream(pe.feed_stream(device)):       # see below: not the default stream
        tickets = [pe.submit(img_rgb, img, pcl, loader, center, M, cube, cam_para) for ... in batches]   # returns immediately
        results, spatial_weights, _ = pe.collect(ticket)      # makes the caller's stream wait for that batch only

Call submit() / collect() from a stream other than the device's DEFAULT stream.  submit() orders the batch behind the caller's stream with an event recorded
there (the inputs were produced on it); recorded on the default stream, that event behaves as a join over every stream of the device on this runtime, and the
batches then run one after the other (measured: 3.35 against 2.11 ms per B = 32 bf16 batch, tools/exp_overlap3.py).  `feed_stream(device)` hands out a
stream for the loop; anything that produced the inputs on another stream is joined the usual way (`feed.wait_stream(producer)`).  Results are the same either way.
"""
import torch


class PipelinedEval:
    def __init__(self, model, depth=2, stages=False):
        """depth: batches in flight (independent sets of static buffers / graphs).  stages=False: whole forwards, one graph and one HIP stream per slot.
        stages=True: a batch is TWO graphs — backbones, head — and all backbone graphs replay on one HIP stream, all head graphs on another, so that the
        backbones of batch i + 1 run beside the head of batch i by construction.  Measured equal within 2 % once the loop submits from a non-default stream
        (B = 32 bf16: 2.07 / 2.11 ms per batch; both were 3.3 ms from the default stream: module docstring); the whole-forward form issues less from the host."""
        if depth < 1:
            raise ValueError("PipelinedEval: depth must be >= 1")
        self.model, self.depth, self.stages = model, int(depth), bool(stages)
        self._streams = None
        self._head_done = [None] * self.depth  # stages: the event after which a slot's static buffers may be overwritten
        self._next = 0

    def feed_stream(self, device):
        """A HIP stream for the loop that calls submit() / collect() (one per PipelinedEval and device; see the module docstring)."""
        device = torch.device(device)
        if getattr(self, "_feed", None) is None or self._feed.device != device:
            self._feed = torch.cuda.Stream(device=device)
        return self._feed

    def submit(self, img_rgb, img, pcl, loader, center, M, cube, cam_para, kernel=0.8):
        """Enqueue one batch (copy-in, graph replay(s), copy-out); returns a ticket for collect()."""
        m = self.model
        if m.training:
            raise RuntimeError("PipelinedEval is an eval-mode loop: call model.eval() first")
        m._require_gpu(img)
        if img.shape[-1] != m.crop_size:
            raise RuntimeError("PipelinedEval: this model was built for %dx%d crops (got %d)" % (m.crop_size, m.crop_size, img.shape[-1]))
        dev = img.device
        plan = m._plan(dev)
        nstreams = 2 if self.stages else self.depth
        if self._streams is None or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
            self._head_done = [None] * self.depth
        slot = self._next
        self._next = (self._next + 1) % self.depth
        cur = torch.cuda.current_stream(dev)
        if cur == torch.cuda.default_stream(dev) and not getattr(self, "_warned_default", False):
            self._warned_default = True
            import warnings
            warnings.warn("PipelinedEval.submit() called on the device's default stream: the batches in flight will run one after the other "
                          "(use `with torch.cuda.stream(pe.feed_stream(device)):` around the loop; see keypointfusion_amd/serving.py)", RuntimeWarning, stacklevel=2)
        img_size, flip = int(getattr(loader, "img_size", m.crop_size)), int(getattr(loader, "flip", 1))
        if not self.stages:
            st = self._streams[slot]
            st.wait_stream(cur)  # the inputs were produced on the caller's stream
            with torch.no_grad(), torch.cuda.device(dev), torch.cuda.stream(st):
                res, sws, _ = plan.forward_graphed(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), img_size, flip, slot=slot)
                ev = torch.cuda.Event()
                ev.record(st)
            return (res, sws, ev, st)
        sa, sb = self._streams
        with torch.no_grad(), torch.cuda.device(dev):
            (ga, gb, static, res, sws), ins = plan.staged_graphs(img_rgb, img, pcl, center, M, cube, cam_para, float(kernel), img_size, flip, slot=slot)
            sa.wait_stream(cur)  # the inputs were produced on the caller's stream
            if self._head_done[slot] is not None:
                sa.wait_event(self._head_done[slot])  # the batch that used this slot last has read its inputs and the backbones' outputs
            with torch.cuda.stream(sa):
                torch._foreach_copy_(static, ins)
                for t in ins:
                    t.record_stream(sa)
                ga.replay()
                ev_bb = torch.cuda.Event()
                ev_bb.record(sa)
            with torch.cuda.stream(sb):
                sb.wait_event(ev_bb)
                gb.replay()
                outs = [torch.empty_like(t) for t in res + sws]
                torch._foreach_copy_(outs, res + sws)
                res, sws = outs[:len(res)], outs[len(res):]
                ev = torch.cuda.Event()
                ev.record(sb)
            self._head_done[slot] = ev
        return (res, sws, ev, sb)

    def collect(self, ticket):
        """(list of 6 results, list of 2 spatial weights, None) of a submitted batch, ordered after it on the caller's stream."""
        res, sws, ev, st = ticket
        cur = torch.cuda.current_stream(res[0].device)
        cur.wait_event(ev)
        for t in res + sws:
            t.record_stream(cur)  # allocated on the slot's stream, consumed on the caller's
        return res, sws, None
