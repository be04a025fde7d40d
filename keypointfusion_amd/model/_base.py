"""Shared plumbing of the drop-in nn.Modules: parameters live in the reference's state-dict layout under the reference's dotted
names; kernel-layout copies ("plans") are built lazily per device and rebuilt when any parameter changes."""
import threading
import zlib

import numpy as np
import torch
import torch.nn as nn

from ..weights import _draw


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state-dict names."""


def _attach(root, dotted, value, is_buffer):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if is_buffer:
        mod.register_buffer(parts[-1], value)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(value))


_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


class SpecModule(nn.Module):
    """nn.Module whose parameters/buffers are given by a keypointfusion_amd.spec list; values are seeded draws (there is no
    checkpoint in this environment) unless `values` supplies them."""

    def __init__(self):
        super().__init__()
        self._plans = {}
        self._plan_lock = threading.Lock()

    def _materialise(self, spec, seed=0, prefix="", values=None, buffers=()):
        for name, shape, dtype, init in spec:
            if values is not None and name in values:
                val = torch.from_numpy(np.asarray(values[name]).copy()).reshape(shape)
            else:
                rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32((prefix + name).encode())]))
                val = torch.from_numpy(np.asarray(_draw(rng, shape, init)).copy()).reshape(shape)
            is_buffer = name.endswith(_BUFFER_SUFFIXES) or name.startswith(tuple(buffers))
            _attach(self, name, val, is_buffer)

    def _state_version(self):
        ts = self.__dict__.get("_tensor_list")  # the module tree is walked once; later calls only read the version counters
        if ts is None:
            ts = self.__dict__["_tensor_list"] = list(self.parameters()) + list(self.buffers())
        return sum(t._version for t in ts)

    # -- copy.deepcopy / pickle: the per-device caches (packed weights, captured graphs, streams, locks) are rebuilt on demand ----
    _CACHE_ATTRS = ("_plans", "_plan_lock", "_tensor_list", "_train_side_stream", "_w16_shadow")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in self._CACHE_ATTRS:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self.__dict__["_plans"] = {}
        self.__dict__["_plan_lock"] = threading.Lock()
        self.__dict__["_tensor_list"] = None

    def _apply(self, fn, *a, **k):
        self.__dict__["_tensor_list"] = None
        self._plans.clear()
        return super()._apply(fn, *a, **k)

    def _plan(self, device, build):
        """Kernel-layout weights for `device` (build(sd, device)), rebuilt when any parameter changed."""
        key = (device.type, device.index)
        ver = self._state_version()
        with self._plan_lock:
            ent = self._plans.get(key)
            if ent is None or ent[0] != ver:
                sd = {k: v.detach() for k, v in self.state_dict().items()}
                ent = (ver, build(sd, device))
                self._plans[key] = ent
            return ent[1]

    def _load_from_state_dict(self, *a, **k):
        self._plans.clear()
        self.__dict__["_tensor_list"] = None
        return super()._load_from_state_dict(*a, **k)

    @staticmethod
    def _require_gpu(t):
        if not t.is_cuda:
            raise RuntimeError("keypointfusion_amd runs on MI355X only: inputs must be on a HIP device "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")
