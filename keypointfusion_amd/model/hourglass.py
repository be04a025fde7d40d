"""Drop-in counterpart of the reference's `model.hourglass.PoseNet` (model/hourglass.py:166-229): stacked 4-level hourglasses of
pre-activation Residual blocks over a 1-channel depth crop.  Same constructor arguments and state-dict keys; forward returns
`(preds [B, 5*joint_num, S/4, S/4], feature [B, inp_dim, S/4, S/4])` of the last stack, NCHW."""
import torch

from ..spec import posenet_spec
from ._base import SpecModule


class PoseNet(SpecModule):
    def __init__(self, nstack, joint_num, inp_dim=256, bn=False, increase=0, seed=0, **kwargs):
        super().__init__()
        if increase != 0:
            raise NotImplementedError("PoseNet(increase != 0) is not built (the reference never uses it)")
        if inp_dim % 8:
            raise ValueError("inp_dim must be a multiple of 8")
        self.nstack, self.joint_num, self.inp_dim = nstack, joint_num, inp_dim
        self._materialise(posenet_spec(nstack, joint_num, inp_dim), seed, prefix="posenet_n%d_d%d." % (nstack, inp_dim))

    def forward(self, img):
        self._require_gpu(img)
        if self.training:  # autograd-connected outputs on this module's Parameters (keypointfusion_amd/heads_train.py: convolutions / Linears on the HIP GEMM)
            if img.shape[-1] % 64 or img.shape[-2] % 64:
                raise RuntimeError("PoseNet needs H and W divisible by 64 (stride-4 stem + 4 pooling levels), got %s" % (tuple(img.shape),))
            from ..heads_train import posenet_train_forward
            with torch.cuda.device(img.device):
                return posenet_train_forward(self, img)
        if img.shape[-1] % 64 or img.shape[-2] % 64:
            raise RuntimeError("PoseNet needs H and W divisible by 64 (stride-4 stem + 4 pooling levels), got %s" % (tuple(img.shape),))
        from ..engine import nhwc_to_nchw
        from ..heads import PoseNetPlan
        plan = self._plan(img.device, lambda sd, dev: PoseNetPlan(sd, self.nstack, dev))
        with torch.cuda.device(img.device):
            preds, feat = plan(img.detach().float().contiguous())
            return preds, nhwc_to_nchw(feat)
