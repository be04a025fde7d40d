"""MI355X-native implementation of the KeypointFusion (ru1ven/KeypointFusion) forward hot path.

    from keypointfusion_amd.model.model import KPFusion     # drop-in for `from model.model import KPFusion`
"""
__all__ = ["spec", "weights", "lib", "engine"]

import os as _os

# HIP runtime workaround (ROCm 7.x, gfx950), read by the runtime when it initialises — i.e. it must be in the environment before the
# first HIP call of the process, which importing this package normally precedes.  With the runtime's "graph packet capture" fast path
# (second and later launches of an instantiated hipGraph replay pre-recorded AQL packets) the small hipMemsetAsync nodes that ATen's
# multi-block reductions put in front of their semaphores are not replayed in order: from the second replay on such a reduction
# returns stale data (found as wrong layer-scale gradients in the captured training iteration; tools/replay_determinism.py,
# DESIGN.md §4.5).  The product's own kernels need no memset nodes; the torch ops around them in the training step do.
# `graphs.assert_replay_is_sound()` checks the behaviour once per process before any training graph is trusted.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
