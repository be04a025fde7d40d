"""Run one eager training iteration with conv_wgrad_hip's buffers inside wide sentinel bands; report any call that writes outside."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synthetic_sd
from keypointfusion_amd import training as T, lib as L
from keypointfusion_amd.model.model import KPFusion
from keypointfusion_amd.weights import synthetic_batch
net = "KPFusion-" + (sys.argv[1] if len(sys.argv) > 1 else "convnext-tiny"); B = 4; dev = torch.device("cuda:0")
GB = 1 << 20  # guard floats (4 MB) on either side
SENT = 12345.678
bad = []
seen = set()
def guarded(n):
    raw = torch.full((n + 2 * GB,), SENT, device=dev, dtype=torch.float32)
    return raw, raw[GB:GB + n]
def check(raw, n, what, shape):
    torch.cuda.synchronize()
    lo, hi = raw[:GB], raw[GB + n:]
    if not (bool((lo == SENT).all()) and bool((hi == SENT).all())):
        nz_hi = (hi != SENT).nonzero().flatten()
        nz_lo = (lo != SENT).nonzero().flatten()
        bad.append((what, shape, int(nz_lo.numel()), int(nz_hi.numel()), int(nz_hi.min()) if nz_hi.numel() else -1, int(nz_hi.max()) if nz_hi.numel() else -1,
                    hi[nz_hi[:4]].tolist() if nz_hi.numel() else []))
def conv_wgrad_hip(dy, x, wshape, stride, pad, want_db=True):
    lib = L.load()
    Bn, H, W, Cin = x.shape
    _, OH, OW, N = dy.shape
    KH, KW = int(wshape[2]), int(wshape[3])
    dy, x = dy.float().contiguous(), x.float().contiguous()
    nws = lib.kpf_conv2d_wgrad_ws_floats(Bn * OH * OW, N, KH * KW * Cin)
    rws, ws = guarded(nws)
    ndw = N * Cin * KH * KW
    rdw, dwf = guarded(ndw)
    rdb, db = guarded(N)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.kpf_conv2d_wgrad_f32(dy.data_ptr(), x.data_ptr(), dwf.data_ptr(), db.data_ptr() if want_db else None, ws.data_ptr(), nws,
                                     Bn, H, W, Cin, Cin, OH, OW, N, N, KH, KW, stride, stride, pad, pad, st), "wgrad")
    shape = (Bn, H, W, Cin, OH, OW, N, KH, stride, pad, want_db)
    if shape not in seen:
        seen.add(shape)
        check(rws, nws, "ws", shape); check(rdw, ndw, "dw", shape); check(rdb, N, "db", shape)
    return dwf.view(tuple(wshape)).clone(), (db.clone() if want_db else None)
T.conv_wgrad_hip = conv_wgrad_hip
sd = synthetic_sd(net)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_batch(B, 128, seed=5).items()}
g = torch.Generator().manual_seed(1)
uvd = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev); xyz = (torch.rand(B, 21, 3, generator=g) * 1.2 - 0.6).to(dev)
class Loader: img_size, flip = 128, 1
m = KPFusion(net, "", 21, "dexycb", ""); m.load_state_dict(sd, strict=True); m = m.to(dev).train(); m.train_dropout = 0.0
bt = batch
results, sws, _ = m(bt["img_rgb"], bt["img"], bt["pcl"], Loader(), bt["center"], bt["M"], bt["cube"], bt["cam_para"], 0.8)
loss = T.kpfusion_loss(results, sws, bt["img"], uvd, xyz, epoch=0)[0]
loss.backward()
torch.cuda.synchronize()
print("distinct wgrad shapes", len(seen), "damaged", len(bad))
for b in bad: print(b)
