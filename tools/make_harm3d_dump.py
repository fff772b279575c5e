#!/usr/bin/env python3
"""Synthetic HARM3D-format dump: the reference reads 3-D HARM data as a 15-number text header file
plus, per time slice, `<dfile>NNN.bin` = one text line ended by a newline followed by REAL(4) data(35, nx1*nx2*nx3),
zones ordered x3 (phi) fastest, then x2, then x1 (fluid_model_harm3d.f90:337-368, 516-584, index arithmetic :222-240).
No such file ships with the reference, so tests and bench read dumps written by this script: the same analytic
Kerr-Schild-regular torus as tools/make_harm_dump.py with m=1,2 azimuthal structure and a seeded small-amplitude
perturbation, expressed in the modified KS coordinates that reader assumes (x1 = log r, theta = pi x2, x3 = phi).

    python tools/make_harm3d_dump.py OUTDIR [--n1 48 --n2 48 --n3 24] [--name dump3d] [--indf 40]
"""
import argparse
import os
import numpy as np
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_harm_dump import ks_metric, SPIN, GAM

SEED = 20261003
DLEN = 35


def make(n1=48, n2=48, n3=24, rout=40.0, a=SPIN, seed=SEED, time=2000.0, phase=0.0, b0=0.12):
    assert n1 <= n2, "the reference's uniqx1 slice (fluid_model_harm3d.f90:139) needs nx1 <= nx2"
    rh = 1.0 + np.sqrt(1.0 - a * a)
    rin = 0.98 * rh
    startx1, dx1 = np.log(rin), (np.log(rout) - np.log(rin)) / n1
    dx2, dx3 = 1.0 / n2, 2.0 * np.pi / n3
    x1 = startx1 + (np.arange(n1) + 0.5) * dx1
    x2 = (np.arange(n2) + 0.5) * dx2
    x3 = (np.arange(n3) + 0.5) * dx3
    X1, X2, X3 = np.meshgrid(x1, x2, x3, indexing="ij")      # flattened: x3 fastest, x1 slowest
    r, th, ph = np.exp(X1), np.pi * X2, X3
    rng = np.random.default_rng(seed)
    wig = 1.0 + 0.02 * rng.standard_normal(r.shape)
    spiral = ph - 1.3 * np.sqrt(r) - phase                   # a trailing two-armed pattern that rotates with `phase`
    az = 1.0 + 0.35 * np.cos(spiral) + 0.15 * np.cos(2.0 * spiral + 0.7)

    g = ks_metric(r, th, a)
    sig = r * r + a * a * np.cos(th) ** 2
    f = 2.0 * r / sig
    lapse = 1.0 / np.sqrt(1.0 + f)
    nvec = np.zeros(r.shape + (4,))
    nvec[..., 0] = 1.0 / lapse
    nvec[..., 1] = -(f / (1.0 + f)) / lapse
    v = np.zeros_like(nvec)
    v[..., 3] = 0.6 / (r ** 1.5 + a) * lapse * (1.0 + 0.05 * np.sin(spiral))
    v[..., 2] = 0.02 * np.sin(2.0 * th) / r ** 1.5 * lapse
    v[..., 1] = 0.01 * np.cos(spiral) / r * lapse
    v2 = np.einsum("...i,...ij,...j->...", v, g, v)
    gamma = 1.0 / np.sqrt(1.0 - v2)
    u = gamma[..., None] * (nvec + v)
    B = np.zeros_like(nvec)
    B0 = b0
    B[..., 1] = B0 * np.cos(th) / r ** 2 * (1.0 + 0.3 * np.sin(th) ** 2)
    B[..., 2] = 0.2 * B0 * np.sin(th) / r ** 3 * (1.0 + 0.2 * np.sin(spiral))
    B[..., 3] = -0.8 * B0 / r ** 2.5 * (1.0 + 0.25 * np.cos(spiral + 0.4))
    ucov = np.einsum("...ij,...j->...i", g, u)
    bt = np.einsum("...i,...i->...", B[..., 1:], ucov[..., 1:])
    b = np.zeros_like(nvec)
    b[..., 0] = bt
    b[..., 1:] = (B[..., 1:] + bt[..., None] * u[..., 1:]) / u[..., 0:1]
    assert np.abs(np.einsum("...i,...i->...", u, ucov) + 1.0).max() < 1e-10
    assert np.abs(np.einsum("...i,...i->...", b, ucov)).max() < 1e-10

    rho = (r / 6.0) ** -1.5 * np.exp(-0.5 * (np.cos(th) / 0.35) ** 2) * np.exp(-(2.2 / r) ** 4) * wig * az + 1e-6
    uint = 0.06 * rho / r ** 0.9 * (1.0 + 0.3 * np.cos(th) ** 2) / (GAM - 1.0)

    def to_mks(w):                                           # KS -> the reader's MKS: u^x1 = u^r / r, u^x2 = u^th / pi
        o = w.copy()
        o[..., 1] = w[..., 1] / r
        o[..., 2] = w[..., 2] / np.pi
        return o
    um, bm = to_mks(u), to_mks(b)
    cols = np.zeros(r.shape + (DLEN,), dtype=np.float32)
    cols[..., 0], cols[..., 1], cols[..., 2] = 0, 0, 0       # zone indices (unused by the reader)
    cols[..., 3], cols[..., 4], cols[..., 5] = X1, X2, X3
    cols[..., 6], cols[..., 7], cols[..., 8] = r, th, ph
    cols[..., 9], cols[..., 10] = rho, uint
    cols[..., 11:14] = um[..., 1:] / um[..., 0:1]
    cols[..., 14:17] = B[..., 1:]
    cols[..., 18:22] = um
    cols[..., 22:26] = ucov
    cols[..., 26:30] = bm
    cols[..., 34] = sig * np.sin(th) * r * np.pi
    header = np.zeros(15)
    header[0] = time
    header[1], header[2], header[3] = n1, n2, n3
    header[4], header[5], header[6] = startx1, 0.0, 0.0
    header[7], header[8], header[9] = dx1, dx2, dx3
    header[10], header[11] = a, GAM
    header[12] = rin
    header[13] = 1.0          # h (read, never used: theta = pi x2 in this reader)
    header[14] = rout
    return header, cols.reshape(n1 * n2 * n3, DLEN)


def write(outdir, name="dump3d", indf=40, **kw):
    """-> (dfile prefix, header file).  The reference opens trim(dfile)//'NNN'//'.bin' and hfile."""
    header, cols = make(**kw)
    os.makedirs(outdir, exist_ok=True)
    hfile = os.path.join(outdir, name + "_header")
    with open(hfile, "w") as f:
        f.write(" ".join("%.16g" % v for v in header) + "\n")
    with open(os.path.join(outdir, "%s%03d.bin" % (name, indf)), "wb") as f:
        f.write((" ".join("%.16g" % v for v in header) + "\n").encode())
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())
    return os.path.join(outdir, name), hfile


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--n1", type=int, default=48)
    ap.add_argument("--n2", type=int, default=48)
    ap.add_argument("--n3", type=int, default=24)
    ap.add_argument("--name", default="dump3d")
    ap.add_argument("--indf", type=int, default=40)
    a = ap.parse_args()
    print(write(a.outdir, a.name, a.indf, n1=a.n1, n2=a.n2, n3=a.n3))
