"""Data-parallel execution of the forward path: one process per GPU, batch shards, no data-path collective.

The reference wraps the model in single-process `torch.nn.DataParallel` (train.py:81, demo_RGBD.py:49): scatter the batch
along dim 0, replicate weights every iteration, gather outputs on GPU 0.  In eval every sample is independent (BatchNorm
uses running statistics), so the MI355X-native form is: weights resident per rank, each rank takes a
contiguous shard of the batch, and the only communication is an optional all_gather of the (small) outputs over
RCCL/xGMI — B x 1.03 MB per image.  `torch.distributed` backend "nccl" is RCCL on ROCm; tests use "gloo" on CPU.
"""
import re

import torch


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n items for `rank` (first n % world ranks get one extra item)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank, world):
    """batch: dict name -> tensor with batch dim 0.  Returns this rank's shard (views, no copy)."""
    n = next(iter(batch.values())).shape[0]
    lo, hi = shard_bounds(n, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


def gather_outputs(tensors, n_total, dist_mod=None, group=None):
    """all_gather a list of per-rank output tensors (batch dim 0, possibly ragged shards) back into full-batch tensors in
    rank order.  With dist_mod None (single process) returns the inputs."""
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size(group) == 1:
        return list(tensors)
    world = dist_mod.get_world_size(group)
    out = []
    q, r = divmod(n_total, world)
    maxn = q + (1 if r else 0)
    for t in tensors:
        pad = torch.zeros((maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist_mod.all_gather(bufs, pad, group=group)
        parts = []
        for rk in range(world):
            lo, hi = shard_bounds(n_total, rk, world)
            parts.append(bufs[rk][: hi - lo])
        out.append(torch.cat(parts, 0))
    return out


def max_over_ranks(value, device, dist_mod=None):
    """max of a python float over all ranks (bench timing contract)."""
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    return float(t.item())


class GradBucketReducer:
    """Data-parallel gradient averaging for the training step: the replacement of the reference's single-process
    DataParallel reduce (train.py:81).  One process per GPU; after `loss.backward()` every rank holds the mean gradient.

    Parameters are packed, in REVERSE registration order (the order backward produces gradients), into flat buckets of about
    `bucket_mb` megabytes; a bucket is all-reduced (sum, asynchronously — RCCL over xGMI for "nccl", tests use "gloo") the moment its
    last gradient has been accumulated (post-accumulate-grad hooks), so communication overlaps the rest of backward; `finish()` waits
    for the outstanding reductions, divides by the world size and scatters the averaged values back into `.grad`.  Parameters that
    received no gradient in a step (the dead modules of the reference: decoder layers 0-2, BERT embeddings / poolers, ConvNeXt heads —
    20-23 % of the parameters) travel as zeros in their bucket slot and KEEP `.grad = None` afterwards when no rank
    produced a gradient for them (one flag per parameter rides at the end of the bucket), so the optimiser skips them exactly as it
    does behind the reference's DataParallel (no weight decay on dead parameters); to keep them out of the payload altogether pass
    `params` = the live ones only (`live_parameters`).  BatchNorm statistics stay per replica, like DataParallel's (no SyncBN)."""

    def __init__(self, params, dist_mod, bucket_mb=64.0, group=None):
        self.dist, self.group = dist_mod, group
        self.world = dist_mod.get_world_size(group) if dist_mod is not None and dist_mod.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets = []  # (flat buffer, [(param, offset, numel)])
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > cap or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._close(cur)
        self._slot, self._index = {}, {}
        for bi, (_, slots) in enumerate(self.buckets):
            for i, (p, off, n) in enumerate(slots):
                self._slot[id(p)] = (bi, off, n)
                self._index[id(p)] = i
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.reset()

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total + len(plist), dtype=plist[0].dtype, device=plist[0].device)  # values, then one "seen" flag per parameter
        slots, off = [], 0
        for p in plist:
            slots.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append((flat, slots))

    def reset(self):
        """Call before each backward (after optimizer.zero_grad())."""
        for bi, (flat, slots) in enumerate(self.buckets):
            self._pending[bi] = len(slots)
            self._work[bi] = None
            flat.zero_()
        self._flags = [[0.0] * len(slots) for _, slots in self.buckets]
        self._seen = set()

    def _on_grad(self, p):
        if id(p) in self._seen:  # (the hook runs once per backward, after all uses of the parameter have been accumulated)
            return
        self._seen.add(id(p))
        bi, off, n = self._slot[id(p)]
        flat, slots = self.buckets[bi]
        flat[off:off + n].copy_(p.grad.reshape(-1))
        self._flags[bi][self._index[id(p)]] = 1.0  # (host side: the hook runs on the host; one copy per bucket at launch, not one fill per parameter)
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        if self.world > 1:
            flat, slots = self.buckets[bi]
            flat[flat.numel() - len(slots):].copy_(torch.tensor(self._flags[bi], dtype=flat.dtype), non_blocking=True)
            self._work[bi] = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait for every bucket, average, and write the result back into the parameters' .grad."""
        for bi, (flat, slots) in enumerate(self.buckets):
            if self._pending[bi] > 0:  # some parameters of this bucket got no gradient this step: reduce what is there (zeros for them)
                self._pending[bi] = 0
                self._launch(bi)
        for w in self._work:
            if w is not None:
                w.wait()
        seen_all = None
        if self.world > 1:  # one device -> host transfer for the flags of every bucket (> 0: some rank produced a gradient)
            seen_all = torch.cat([flat[flat.numel() - len(slots):] for flat, slots in self.buckets]).tolist()
        pos = 0
        for bi, (flat, slots) in enumerate(self.buckets):
            seen = seen_all[pos:pos + len(slots)] if seen_all is not None else None
            pos += len(slots)
            if self.world > 1:
                flat.div_(self.world)
            for i, (p, off, n) in enumerate(slots):
                if p.grad is not None or (self.world > 1 and seen[i] > 0):
                    g = flat[off:off + n].view_as(p)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)

    def payload_bytes(self):
        return sum((f.numel() - len(sl)) * f.element_size() for f, sl in self.buckets)  # (gradient values; + 4 B of flag per parameter)

    def remove(self):
        for h in self._hooks:
            h.remove()


DEAD_PREFIXES = ("crossTR.decoder.0.", "crossTR.decoder.1.", "crossTR.decoder.2.", "crossTR.decoder.3.norm1.", ".bert.embeddings.", ".bert.pooler.",
                 "backbone.head.", "backbone.norm.", ".attention_weights.", ".reduction_joint_feature.", ".reduction_joint_feature_update.",
                 ".sampling_feature_embding.", ".sampling_offsets.",
                 ".feat_emb.")  # (the backbones' unused Residual — NOT block*.joint_feat_emb / pcl_feat_emb*, which train)
DEAD_PATTERNS = (re.compile(r"^block\d+\.cls_head\."),)  # (block*.init_TR.cls_head / final_TR.cls_head are live)


def live_parameters(module):
    """Parameters that can receive a gradient in the reference's forward (decoder layers 0-2, the BERT embedding and
    pooler tables, the ConvNeXt classifier head and `feat_emb` are dead code whose parameters never reach the outputs)."""
    return [p for n, p in module.named_parameters() if not (any(d in n for d in DEAD_PREFIXES) or any(r.search(n) for r in DEAD_PATTERNS))]
