"""Drop-in counterpart of the reference's `model.cbam.CBAM` (model/cbam.py:84-94): same constructor, same state-dict keys, NCHW
tensors in and out; `forward` returns `x*scale` for no_spatial=True and otherwise the SpatialGate *tuple* `(x_out*s, x_out*(1-s))`
exactly as the reference does (model/cbam.py:82).  Only pool_types ['avg', 'max'] (the default, the only one used) are built."""
import torch

from ..spec import cbam_spec
from ._base import SpecModule


class CBAM(SpecModule):
    def __init__(self, gate_channels, reduction_ratio=16, pool_types=("avg", "max"), no_spatial=False, seed=0):
        super().__init__()
        if list(pool_types) != ["avg", "max"]:
            raise NotImplementedError("CBAM pool_types other than ['avg', 'max'] are not built")
        if gate_channels % 4:
            raise ValueError("gate_channels must be a multiple of 4")
        self.gate_channels, self.no_spatial = gate_channels, no_spatial
        self._materialise(cbam_spec(gate_channels, reduction_ratio, no_spatial), seed, prefix="cbam_C%d_%d." % (gate_channels, int(no_spatial)))

    def forward(self, x):
        self._require_gpu(x)
        if self.training:  # autograd-connected outputs on this module's Parameters (keypointfusion_amd/heads_train.py: convolutions / Linears on the HIP GEMM)
            from ..heads_train import cbam_train_forward
            with torch.cuda.device(x.device):
                return cbam_train_forward(self, x)
        from ..engine import nchw_to_nhwc, nhwc_to_nchw
        from ..heads import CbamPlan
        plan = self._plan(x.device, lambda sd, dev: CbamPlan(sd, dev))
        with torch.cuda.device(x.device):
            out = plan(nchw_to_nhwc(x.detach()))
            if self.no_spatial:
                return nhwc_to_nchw(out)
            return nhwc_to_nchw(out[0]), nhwc_to_nchw(out[1])
