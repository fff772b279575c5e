#!/usr/bin/env python3
"""Synthetic HARM-format 2-D dump: the reference's HARM regression needs `dump040`,
which is not shipped, so tests and bench read a dump written by this script instead -- same text format
(header of 26 numbers, then one 34-column line per zone, x2 fastest; fluid_model_harm.f90:275-352), smooth
analytic torus, seeded small-amplitude perturbation so that no two zones are equal.

The flow is built directly in Kerr-Schild coordinates so that it is regular across the horizon (the grid,
like HARM's, starts inside it): u = Gamma (n + v) with n the KS normal observer and v a sub-Keplerian
rotation; b^mu from a lab-frame field B^i by the usual HARM relations b^t = B^i u_i, b^i = (B^i + b^t u^i)/u^t,
which makes u.u = -1 and b.u = 0 to rounding.  Both are then expressed in HARM's modified KS coordinates
(x1 = log r, theta = pi x2 + (1-h)/2 sin(2 pi x2)).

    python tools/make_harm_dump.py OUTDIR [--n 256] [--name dump040]
"""
import argparse
import os
import numpy as np

SEED = 20261002
SPIN, HSLOPE, GAM = 0.9375, 0.3, 4.0 / 3.0


def ks_metric(r, th, a):
    s2 = np.sin(th) ** 2
    sig = r * r + a * a * np.cos(th) ** 2
    g = np.zeros(r.shape + (4, 4))
    f = 2.0 * r / sig
    g[..., 0, 0] = -(1.0 - f)
    g[..., 0, 1] = g[..., 1, 0] = f
    g[..., 0, 3] = g[..., 3, 0] = -a * s2 * f
    g[..., 1, 1] = 1.0 + f
    g[..., 1, 3] = g[..., 3, 1] = -a * s2 * (1.0 + f)
    g[..., 2, 2] = sig
    g[..., 3, 3] = s2 * (sig + a * a * s2 * (1.0 + f))
    return g


def make(n=256, rout=40.0, a=SPIN, h=HSLOPE, seed=SEED, time=2000.0):
    rh = 1.0 + np.sqrt(1.0 - a * a)
    rin = 0.98 * rh
    startx1, startx2 = np.log(rin), 0.0
    dx1, dx2 = (np.log(rout) - startx1) / n, 1.0 / n
    x1 = startx1 + (np.arange(n) + 0.5) * dx1
    x2 = startx2 + (np.arange(n) + 0.5) * dx2
    X1, X2 = np.meshgrid(x1, x2, indexing="ij")          # [i1][i2], x2 fastest when flattened
    r = np.exp(X1)
    th = np.pi * X2 + 0.5 * (1.0 - h) * np.sin(2.0 * np.pi * X2)
    dthdx2 = np.pi * (1.0 + (1.0 - h) * np.cos(2.0 * np.pi * X2))
    rng = np.random.default_rng(seed)
    wig = 1.0 + 0.02 * rng.standard_normal(r.shape)

    g = ks_metric(r, th, a)
    sig = r * r + a * a * np.cos(th) ** 2
    f = 2.0 * r / sig
    lapse = 1.0 / np.sqrt(1.0 + f)
    nvec = np.zeros(r.shape + (4,))
    nvec[..., 0] = 1.0 / lapse
    nvec[..., 1] = -(f / (1.0 + f)) / lapse
    v = np.zeros_like(nvec)
    v[..., 3] = 0.6 / (r ** 1.5 + a) * lapse             # sub-Keplerian rotation seen by the normal observer
    v[..., 2] = 0.02 * np.sin(2.0 * th) / r ** 1.5 * lapse
    v2 = np.einsum("...i,...ij,...j->...", v, g, v)
    gamma = 1.0 / np.sqrt(1.0 - v2)
    u = gamma[..., None] * (nvec + v)
    B = np.zeros_like(nvec)
    B0 = 0.12
    B[..., 1] = B0 * np.cos(th) / r ** 2 * (1.0 + 0.3 * np.sin(th) ** 2)
    B[..., 2] = 0.2 * B0 * np.sin(th) / r ** 3
    B[..., 3] = -0.8 * B0 / r ** 2.5
    ucov = np.einsum("...ij,...j->...i", g, u)
    bt = np.einsum("...i,...i->...", B[..., 1:], ucov[..., 1:])
    b = np.zeros_like(nvec)
    b[..., 0] = bt
    b[..., 1:] = (B[..., 1:] + bt[..., None] * u[..., 1:]) / u[..., 0:1]
    bcov = np.einsum("...ij,...j->...i", g, b)
    assert np.abs(np.einsum("...i,...i->...", u, ucov) + 1.0).max() < 1e-10
    assert np.abs(np.einsum("...i,...i->...", b, ucov)).max() < 1e-10

    rho = (r / 6.0) ** -1.5 * np.exp(-0.5 * (np.cos(th) / 0.35) ** 2) * np.exp(-(2.2 / r) ** 4) * wig + 1e-6
    uint = 0.06 * rho / r ** 0.9 * (1.0 + 0.3 * np.cos(th) ** 2) / (GAM - 1.0)

    # KS -> modified KS components
    def to_mks(w):
        o = w.copy()
        o[..., 1] = w[..., 1] / r
        o[..., 2] = w[..., 2] / dthdx2
        return o
    um, bm = to_mks(u), to_mks(b)
    gdet = sig * np.sin(th) * r * dthdx2
    cols = np.zeros(r.shape + (34,))
    cols[..., 0], cols[..., 1], cols[..., 2], cols[..., 3] = X1, X2, r, th
    cols[..., 4], cols[..., 5] = rho, uint
    cols[..., 6:9] = um[..., 1:] / um[..., 0:1]
    cols[..., 9:12] = B[..., 1:]
    cols[..., 13:17] = um
    cols[..., 17:21] = ucov
    cols[..., 21:25] = bm
    cols[..., 25:29] = bcov
    cols[..., 33] = gdet
    header = np.zeros(26)
    header[0] = time          # dump time (M): consecutive dumps of a slow-light set differ by the time step
    header[1], header[2] = n, n
    header[3], header[4], header[5], header[6] = startx1, startx2, dx1, dx2
    header[7] = 2000.0
    header[9], header[10] = a, GAM
    header[11:16] = [0.9, 40, 1.0, 1.0, 1.0]
    header[18], header[20], header[21], header[22], header[23] = 0.0, rin, rout, 0.0, 0.0
    header[24] = h
    return header, cols.reshape(n * n, 34)


def write(outdir, n=256, name="dump040", **kw):
    header, cols = make(n=n, **kw)
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, name)
    with open(path, "w") as f:
        f.write(" ".join("%.16g" % v for v in header) + "\n")
        np.savetxt(f, cols, fmt="%.10e")
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--name", default="dump040")
    a = ap.parse_args()
    print(write(a.outdir, a.n, a.name))
