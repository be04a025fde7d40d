"""GPU box: in-kernel wall-clock stamps of the fused 21-token stack kernels (csrc/kpf_trstack.hip, workgroup 0): where a stack's time goes, phase by phase.
usage: python tools/trstack_stamps.py [B] [f32|bf16|f16]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from keypointfusion_amd import lib as L, training as T
from test_kernels_train_gpu import _bert_stack_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
prm = _bert_stack_params(gen, dev)
e = torch.randn(B, 21, 128, generator=gen).to(dev).requires_grad_(True)
pos = torch.randn(21, 128, generator=gen).to(dev).requires_grad_(True)
names = ["s.%d.%s" % (l, k) for l in range(4) for k in T.BertStack21.ORDER]
rng = torch.tensor([1, 1], dtype=torch.int64, device=dev)
st = torch.zeros(64, dtype=torch.int64, device=dev)
lib = L.load()
for it in range(3):
    L.check(lib.kpf_tr_stack_set_stamps(st.data_ptr() if it == 2 else None))
    out = T.bert_stack21(e, pos, names, None, 0.1, rng, 1, prm, PREC)
    out.sum().backward()
    torch.cuda.synchronize()
L.check(lib.kpf_tr_stack_set_stamps(None))
v = st.cpu().tolist()
f = lambda a, b: (v[b] - v[a]) / 100.0
print("forward (us): per layer [qkv gemm | attention | Wo gemm | LN1 | Wi gemm | Wo2 gemm | LN2 + tail]")
for l in range(4):
    k = l * 8
    print("  layer %d: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f   total %6.2f" % (l, f(k, k + 1), f(k + 1, k + 2), f(k + 2, k + 3), f(k + 3, k + 4), f(k + 4, k + 5), f(k + 5, k + 6), f(k + 6, k + 7), f(k, k + 7)))
print("  all four layers: %.2f us" % f(0, 31))
print("backward (us): per layer [LN2 bwd (+ requests) | Wo2, Wi gemms | LN1 bwd | Wo gemm | attention bwd | qkv gemm]")
for i in range(4):
    k = 32 + i * 8
    print("  layer %d: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f   total %6.2f" % (3 - i, f(k, k + 1), f(k + 1, k + 2), f(k + 2, k + 3), f(k + 3, k + 4), f(k + 4, k + 5), f(k + 5, k + 6), f(k, k + 6)))
print("  all four layers: %.2f us" % f(32, 32 + 3 * 8 + 6))
