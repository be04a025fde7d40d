"""The rounding order of the reference's 3x3 inverse in its PyTorch-CPU forward, and which of its two variants this host's CPU library takes.

The reference inverts the crop matrix with `torch.linalg.inv` on whatever device `uvd` lives on (dataloader/loader.py:780-781).  The
parity target of this project is the reference's PyTorch-CPU forward (BASELINE.json north_star), where that call lands in MKL's
getrf / getrs; run on a GPU the reference inverts in the GPU solver library instead, and no bit contract exists against that.  Pixel
positions — and through them the integer top-4 pixel indices and ball-query sets — depend on the last bit of that inverse, so the device
computes M^-1 in the CPU library's operation order (csrc/kpf_geom.hip, inv3x3_lapack_order).  ATen's linalg_inv_ex -> linalg_solve_ex
factors a contiguous row-major A as its transpose (no copy) and solves with trans = 'T':  getrf(A^T) = P L U;  U^T y = e_c;  L^T x = y;
rows of X un-permuted.  MKL's 3x3 path was pinned bit for bit with tools/mkl_inv_probe.py on both host types this project runs on:

  * partial-pivot LU of A^T, column 0 scaled by the RECIPROCAL of the pivot, column 1 by a true DIVISION;
  * U^T y = e_c with reciprocal diagonals;  x1 = y1 - l21*x2;  x0 = y0 - (l10*x1 + l20*x2);
  * on CPUs where MKL dispatches its FMA code path (Intel AVX2/AVX-512) every `c - a*b` is one fused operation and the last line is
    y0 - fma(l10, x1, l20*x2); on its generic path (AMD EPYC) every product and sum is rounded separately.

`host_mode()` finds out which of the two this host runs by comparing both restatements with torch.linalg.inv on a set of crop matrices
(milliseconds, once per process): 1 = fused, 0 = separately rounded.  If the host's library follows neither (a non-MKL torch build) the
device uses the separately rounded order (0) and says so once: the inverse is then a correctly pivoted fp32 LU inverse that may differ
from that host's torch.linalg.inv in the last bit, and nothing ever leaves the device (no host copy, hipGraph-capturable).
`KPF_INV3X3_MODE=0|1` pins the order whatever the host (bit-reproducible results across machines).
"""
import threading

import numpy as np

f = np.float32


def _fma(a, b, c):
    return f(np.float64(a) * np.float64(b) + np.float64(c))  # the product of two fp32 numbers is exact in fp64


def inv3x3(m, fused):
    """fp32 inverse of one 3x3 matrix in the order described above (`fused`: MKL's FMA code path)."""
    msub = (lambda c, a, b: _fma(-a, b, c)) if fused else (lambda c, a, b: f(c - f(a * b)))
    t = np.asarray(m, dtype=f).T.copy()  # rows of A^T
    p0 = int(np.argmax(np.abs(t[:, 0])))  # isamax: the first maximum
    if p0 != 0:
        t[[0, p0]] = t[[p0, 0]]
    rc = f(1) / t[0, 0]
    t[1, 0] = f(t[1, 0] * rc)
    t[2, 0] = f(t[2, 0] * rc)
    for i in (1, 2):
        for k in (1, 2):
            t[i, k] = msub(t[i, k], t[i, 0], t[0, k])
    p1 = 2 if abs(t[2, 1]) > abs(t[1, 1]) else 1
    if p1 == 2:
        t[[1, 2]] = t[[2, 1]]
    t[2, 1] = f(t[2, 1] / t[1, 1])
    t[2, 2] = msub(t[2, 2], t[2, 1], t[1, 2])
    l10, l20, l21 = t[1, 0], t[2, 0], t[2, 1]
    rd = [f(1) / t[k, k] for k in range(3)]
    x = np.zeros((3, 3), dtype=f)
    for c in range(3):
        b = np.zeros(3, dtype=f)
        b[c] = 1
        y0 = f(b[0] * rd[0])
        b1 = f(b[1] - f(y0 * t[0, 1]))
        b2 = f(b[2] - f(y0 * t[0, 2]))
        y1 = f(b1 * rd[1])
        b2 = f(b2 - f(y1 * t[1, 2]))
        y2 = f(b2 * rd[2])
        x2 = y2
        x1 = msub(y1, x2, l21)
        x0 = f(y0 - (_fma(l10, x1, f(l20 * x2)) if fused else f(f(l10 * x1) + f(l20 * x2))))
        x[:, c] = (x0, x1, x2)
    if p1 == 2:
        x[[1, 2]] = x[[2, 1]]
    if p0 != 0:
        x[[0, p0]] = x[[p0, 0]]
    return x


def crop_matrices(n, seed=0):
    """Affine crop matrices as the dataloader builds them (scale, optional in-plane rotation, translation; last row 0 0 1)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, 3), dtype=f)
    for i in range(n):
        s = rng.uniform(0.2, 1.5)
        th = 0.0 if i % 2 == 0 else rng.uniform(-np.pi, np.pi)
        tx, ty = rng.uniform(-400, 100, 2)
        out[i] = [[s * np.cos(th), -s * np.sin(th), tx], [s * np.sin(th), s * np.cos(th), ty], [0, 0, 1]]
    return out


_mode = None
_lock = threading.Lock()


def host_mode():
    """The rounding order the device uses for M^-1: 1 / 0 = this host's torch.linalg.inv follows the fused / separately rounded order;
    -1 (with a one-time warning) if it follows neither — engine.crop_inverse then inverts on the host in eager forwards (bit-exact, one
    synchronisation) and uses order 0 only under stream capture; KPF_INV3X3_MODE=0|1 overrides (see module docstring)."""
    global _mode
    if _mode is None:
        with _lock:
            if _mode is None:
                import os
                env = os.environ.get("KPF_INV3X3_MODE")
                if env is not None:
                    if env not in ("0", "1"):
                        raise ValueError("KPF_INV3X3_MODE must be 0 or 1, got %r" % (env,))
                    _mode = int(env)
                    return _mode
                mode = probe_host()
                if mode < 0:
                    import warnings
                    warnings.warn("keypointfusion_amd: this host's torch.linalg.inv follows neither known 3x3 rounding order: eager forwards invert "
                                  "the crop matrix on the host (exact, one synchronisation per forward); captured forwards use the separately "
                                  "rounded LU order on the device (last-bit differences from this host's CPU inverse are possible).  "
                                  "KPF_INV3X3_MODE=0|1 pins an order.")
                _mode = mode
    return _mode


def probe_host():
    """1 / 0 / -1: which restatement (fused / separately rounded / neither) equals this host's torch.linalg.inv bit for bit."""
    import torch
    rng = np.random.default_rng(12345)
    Ms = np.concatenate([crop_matrices(48, seed=777), rng.normal(size=(16, 3, 3)).astype(f)])
    ref = torch.linalg.inv(torch.from_numpy(Ms).view(-1, 1, 3, 3)).view(-1, 3, 3).numpy()
    for cand in (1, 0):
        if all(np.array_equal(inv3x3(m, cand), r) for m, r in zip(Ms, ref)):
            return cand
    return -1
