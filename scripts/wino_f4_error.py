"""CPU study: how much fp32 rounding error do Winograd F(2x2,3x3) / F(4x4,3x3) add to the embeddings?

Emulates the transform-domain arithmetic in fp32 with torch CPU ops (phase decomposition for the
dilated layers, exactly like csrc/wino.hip), runs the whole two-branch network on one bench frame with
the calibrated weights and compares the unit-norm embeddings with an fp64 direct evaluation.

Usage: python scripts/wino_f4_error.py [H W]      (test / development tool, not product code)
"""
import sys
import os
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from unseenobjectclustering_amd import synth  # noqa: E402
from oracle import backbone_oracle as bo  # noqa: E402


def cook_toom(points, m, r=3):
    """AT [m,n], G [n,r], BT [n,n] (float64) for F(m, r) with the given n-1 finite points + infinity."""
    n = m + r - 1
    a = [Fraction(p) for p in points]
    assert len(a) == n - 1
    AT = [[a[j] ** i for j in range(n - 1)] + [Fraction(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fraction(1)
        for l in range(n - 1):
            if l != j:
                N *= a[j] - a[l]
        G.append([a[j] ** k / N for k in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    AT = np.array([[float(v) for v in row] for row in AT])
    G = np.array([[float(v) for v in row] for row in G])
    # solve  sum_j AT[i,j] G[j,q] BT[j,p] = delta(p, i+q)  for BT, column by column
    BT = np.zeros((n, n))
    rows = [(i, q) for i in range(m) for q in range(r)]
    A = np.array([[AT[i, j] * G[j, q] for j in range(n)] for (i, q) in rows])
    for p in range(n):
        b = np.array([1.0 if p == i + q else 0.0 for (i, q) in rows])
        sol, res, rank, _ = np.linalg.lstsq(A, b, rcond=None)
        assert np.allclose(A @ sol, b, atol=1e-10), (p, A @ sol - b)
        BT[:, p] = sol
    return AT, G, BT


def rebalance(AT, G, BT, scale):
    """Move per-frequency scale factors between G and BT (BT row j * s_j, G row j / s_j)."""
    s = np.asarray(scale, dtype=np.float64)
    return AT, G / s[:, None], BT * s[:, None]


def split3(x):
    """x = x1 + x2 + x3 (+ ~2^-24 |x|) with bf16 terms: what a split-precision GEMM on the bf16 matrix pipe would feed."""
    x1 = x.to(torch.bfloat16).to(torch.float32)
    r = x - x1
    x2 = r.to(torch.bfloat16).to(torch.float32)
    x3 = (r - x2).to(torch.bfloat16).to(torch.float32)
    return x1, x2, x3


class Wino:
    def __init__(self, m, points, dtype=torch.float32, scale=None, bf16x3=0):
        AT, G, BT = cook_toom(points, m)
        if scale is not None:
            AT, G, BT = rebalance(AT, G, BT, scale)
        self.m, self.n = m, m + 2
        self.AT64, self.G64, self.BT64 = AT, G, BT
        self.AT = torch.tensor(AT, dtype=dtype)
        self.BT = torch.tensor(BT, dtype=dtype)
        self.dtype = dtype
        self.bf16x3 = bf16x3     # 0 = fp32 GEMM; 6 / 3 = products kept of the 3 x 3 bf16 term pairs (hh hm mh hl lh mm / hh hm mh)

    def conv(self, x, w, dil):
        """x [B,C,H,W] fp32, w [Co,C,3,3] fp32 -> [B,Co,H,W] (stride 1, padding = dilation)."""
        m, n = self.m, self.n
        # weights transformed in fp64 then rounded once (what uoc_net_finalize could do)
        U = torch.einsum("ia,ocab,jb->ijoc", torch.tensor(self.G64), w.double(), torch.tensor(self.G64)).to(self.dtype)
        B, C, H, W = x.shape
        out = torch.zeros(B, w.shape[0], H, W, dtype=self.dtype)
        for py in range(dil):
            for px in range(dil):
                xp = x[:, :, py::dil, px::dil]
                h, wd = xp.shape[2:]
                th, tw = -(-h // m), -(-wd // m)
                xpad = F.pad(xp, (1, tw * m - wd + 1, 1, th * m - h + 1))
                patches = xpad.unfold(2, n, m).unfold(3, n, m)          # [B,C,th,tw,n,n]
                V = torch.einsum("ia,bcyxae,je->ijbyxc", self.BT, patches.to(self.dtype), self.BT)
                if self.bf16x3:
                    v, u = split3(V), split3(U)
                    pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)][:self.bf16x3]
                    M = sum(torch.einsum("ijbyxc,ijoc->ijbyxo", v[a], u[b]) for a, b in reversed(pairs))   # small terms first
                else:
                    M = torch.einsum("ijbyxc,ijoc->ijbyxo", V, U)
                Y = torch.einsum("ai,ijbyxo,ej->boyaxe", self.AT, M, self.AT)   # [B,Co,th,m,tw,m]
                Y = Y.reshape(B, -1, th * m, tw * m)[:, :, :h, :wd]
                out[:, :, py::dil, px::dil] = Y
        return out


def forward(sd, x, pfx, conv3, dtype):
    """resnet34_8s with a pluggable 3x3-stride-1 convolution (BN folded AFTER the conv like the oracle)."""
    def t(k):
        return torch.as_tensor(sd[k]).to(dtype)

    def bn(p, v):
        return F.batch_norm(v, t(p + ".running_mean"), t(p + ".running_var"), t(p + ".weight"), t(p + ".bias"),
                            training=False, eps=bo.BN_EPS)
    size = x.shape[2:]
    x = F.relu(bn(pfx + "bn1", F.conv2d(x, t(pfx + "conv1.weight"), stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    inpl, cur_stride, cur_dil = 64, 4, 1
    for li, (nb, planes) in enumerate(zip(bo.BLOCKS, bo.PLANES), start=1):
        stride = 1 if li == 1 else 2
        down = stride != 1 or inpl != planes
        if down:
            if cur_stride == 8:
                cur_dil *= stride
                stride = 1
            else:
                cur_stride *= stride
        for bi in range(nb):
            p = f"{pfx}layer{li}.{bi}."
            s = stride if bi == 0 else 1

            def c3(v, key, st):
                w = t(key)
                if st == 1 and conv3 is not None and w.shape[1] >= 128:
                    return conv3(v, w, cur_dil)
                return F.conv2d(v, w, stride=st, padding=cur_dil, dilation=cur_dil)
            out = F.relu(bn(p + "bn1", c3(x, p + "conv1.weight", s)))
            out = bn(p + "bn2", c3(out, p + "conv2.weight", 1))
            res = x
            if bi == 0 and down:
                res = bn(p + "downsample.1", F.conv2d(x, t(p + "downsample.0.weight"), stride=s))
            x = F.relu(out + res)
        inpl = planes
    x = F.conv2d(x, t(pfx + "fc.weight"), t(pfx + "fc.bias"))
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def embed(sd, img, xyz, conv3, dtype):
    with torch.no_grad():
        a = forward(sd, img.to(dtype), "fcn.resnet34_8s.", conv3, dtype)
        b = forward(sd, xyz.to(dtype), "fcn_depth.resnet34_8s.", conv3, dtype)
        return F.normalize(a + b, p=2, dim=1)


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
    torch.set_num_threads(8)
    sd = synth.calibrated_state_dict()
    fr = synth.palette_frame(10000, H, W)
    img, xyz = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    ref64 = embed(sd, img, xyz, None, torch.float64)
    direct32 = embed(sd, img, xyz, None, torch.float32)
    print(f"direct fp32 (torch CPU) vs fp64: max {float((direct32.double() - ref64).abs().max()):.3e}")
    cands = {
        "F2 std (0,1,-1)": Wino(2, [0, 1, -1]),
        "F4 std (0,1,-1,2,-2)": Wino(4, [0, 1, -1, 2, -2]),
        "F4 (0,1,-1,1/2,-1/2)": Wino(4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)]),
        "F4 (0,1,-1,1/2,-2)": Wino(4, [0, 1, -1, Fraction(1, 2), -2]),
        "F4 (0,1,-1,2,-1/2)": Wino(4, [0, 1, -1, 2, Fraction(-1, 2)]),
        "F4 (0,1,-1,2,-1/2) bf16x3, 6 products": Wino(4, [0, 1, -1, 2, Fraction(-1, 2)], bf16x3=6),
        "F4 (0,1,-1,2,-1/2) bf16x3, 3 products": Wino(4, [0, 1, -1, 2, Fraction(-1, 2)], bf16x3=3),
    }
    if os.environ.get("WINO_STUDY_ONLY"):
        cands = {k: v for k, v in cands.items() if os.environ["WINO_STUDY_ONLY"] in k}
    for name, wz in cands.items():
        e = embed(sd, img, xyz, wz.conv, torch.float32)
        d64 = (e.double() - ref64).abs()
        d32 = (e - direct32).abs()
        print(f"{name:40s} vs fp64: max {float(d64.max()):.3e} mean {float(d64.mean()):.3e}   vs direct fp32: max {float(d32.max()):.3e}")


if __name__ == "__main__":
    main()
