#!/usr/bin/env python3
"""How the rounding order of the host's torch.linalg.inv (3x3, fp32) was pinned: brute-force search over LU / triangular-solve
operation orders, compared bit for bit against torch.linalg.lu_factor / torch.linalg.inv on THIS host's CPU.  The winning order is what
oracle/inv3x3.py restates and csrc/kpf_geom.hip implements.  MKL dispatches by CPU vendor / ISA, so run it on every host type."""
import itertools
import json
import sys

import numpy as np
import torch

f = np.float32


def fma(a, b, c):
    return f(np.float64(a) * np.float64(b) + np.float64(c))


def ms(c, a, b, uf):  # c - a*b
    return fma(-a, b, c) if uf else f(c - f(a * b))


def lu(At, v):
    At = At.copy()
    pv = []
    p = int(np.argmax(np.abs(At[:, 0])))
    pv.append(p)
    if p:
        At[[0, p]] = At[[p, 0]]
    if v["s0"]:
        r = f(1) / At[0, 0]
        At[1, 0] = f(At[1, 0] * r)
        At[2, 0] = f(At[2, 0] * r)
    else:
        At[1, 0] = f(At[1, 0] / At[0, 0])
        At[2, 0] = f(At[2, 0] / At[0, 0])
    for i in (1, 2):
        for k in (1, 2):
            At[i, k] = ms(At[i, k], At[i, 0], At[0, k], v["u0"])
    p = 2 if abs(At[2, 1]) > abs(At[1, 1]) else 1
    pv.append(p)
    if p == 2:
        At[[1, 2]] = At[[2, 1]]
    At[2, 1] = f(At[2, 1] * (f(1) / At[1, 1])) if v["s1"] else f(At[2, 1] / At[1, 1])
    At[2, 2] = ms(At[2, 2], At[2, 1], At[1, 2], v["u1"])
    return At, pv


def two(b, a1, x1, a2, x2, form):
    """b - a1*x1 - a2*x2 in several rounding orders."""
    if form == 0:
        return f(f(b - f(a1 * x1)) - f(a2 * x2))
    if form == 1:
        return fma(-a2, x2, fma(-a1, x1, b))
    if form == 2:
        return f(f(b - f(a2 * x2)) - f(a1 * x1))
    if form == 3:
        return fma(-a1, x1, fma(-a2, x2, b))
    if form == 4:
        return f(b - fma(a1, x1, f(a2 * x2)))
    if form == 5:
        return f(b - fma(a2, x2, f(a1 * x1)))
    if form == 6:
        return f(b - f(f(a1 * x1) + f(a2 * x2)))
    if form == 7:
        return fma(-a2, x2, f(b - f(a1 * x1)))
    if form == 8:
        return fma(-a1, x1, f(b - f(a2 * x2)))
    raise ValueError


NF = 9


def solve(At, pv, v):
    X = np.zeros((3, 3), dtype=f)
    d = [At[k, k] for k in range(3)]
    rd = [f(1) / x for x in d]
    dv = (lambda x, k: f(x * rd[k])) if v["rd"] else (lambda x, k: f(x / d[k]))
    l10, l20, l21 = At[1, 0], At[2, 0], At[2, 1]
    for c in range(3):
        b = [f(c == 0), f(c == 1), f(c == 2)]
        y0 = dv(b[0], 0)
        y1 = dv(ms(b[1], y0, At[0, 1], v["f1"]), 1)
        y2 = dv(two(b[2], At[0, 2], y0, At[1, 2], y1, v["t1"]), 2)
        x2 = y2
        x1 = ms(y1, x2, l21, v["f2"])
        x0 = two(y0, l10, x1, l20, x2, v["t2"])
        X[:, c] = (x0, x1, x2)
    if pv[1] == 2:
        X[[1, 2]] = X[[2, 1]]
    if pv[0]:
        X[[0, pv[0]]] = X[[pv[0], 0]]
    return X


def crop_matrices(n, rng):
    out = np.zeros((n, 3, 3), dtype=f)
    for i in range(n):
        s = rng.uniform(0.2, 1.5)
        th = 0.0 if i % 2 == 0 else rng.uniform(-np.pi, np.pi)
        tx, ty = rng.uniform(-400, 100, 2)
        out[i] = [[s * np.cos(th), -s * np.sin(th), tx], [s * np.sin(th), s * np.cos(th), ty], [0, 0, 1]]
    return out


def main():
    rng = np.random.default_rng(0)
    Ms = np.concatenate([crop_matrices(120, rng), rng.normal(size=(80, 3, 3)).astype(f)])
    A = torch.from_numpy(Ms)
    LU, piv = torch.linalg.lu_factor(A.mT)
    LU, piv = LU.numpy(), piv.numpy()
    ref = torch.linalg.inv(A.view(-1, 1, 3, 3)).view(-1, 3, 3).numpy()
    print("torch", torch.__version__, "threads", torch.get_num_threads())
    res = []
    for s0, s1, u0, u1 in itertools.product((0, 1), repeat=4):
        v = dict(s0=s0, s1=s1, u0=u0, u1=u1)
        ok = sum(np.array_equal(lu(m.T.copy(), v)[0], LU[i]) for i, m in enumerate(Ms))
        res.append((ok, v))
    res.sort(key=lambda t: -t[0])
    print("LU stage (of %d):" % len(Ms), res[:4])
    res = []
    for rd, f1, f2 in itertools.product((0, 1), repeat=3):
        for t1 in range(NF):
            for t2 in range(NF):
                v = dict(rd=rd, f1=f1, f2=f2, t1=t1, t2=t2)
                ok = sum(np.array_equal(solve(LU[i], [int(piv[i][0]) - 1, int(piv[i][1]) - 1], v), ref[i]) for i in range(len(Ms)))
                res.append((ok, v))
    res.sort(key=lambda t: -t[0])
    print("solve stage on torch's LU (of %d):" % len(Ms), res[:6])
    # a few raw examples for offline analysis
    dump = [dict(M=Ms[i].view(np.uint32).tolist(), LU=LU[i].view(np.uint32).tolist(), piv=piv[i].tolist(), inv=ref[i].view(np.uint32).tolist())
            for i in (0, 1, 2, 3, 120, 121, 122, 123)]
    print("DUMP", json.dumps(dump))


if __name__ == "__main__":
    main()
