#!/usr/bin/env python3
"""Synthetic iharm-format dump: the reference's IHARM reader takes a text header (tcur, nx1, nx2, nx3,
a, hslope, gam, mks_smooth, poly_xt, poly_alpha, startx1, metric, eHEAT, eRAD, n_prim, Rin, Rout;
fluid_model_iharm.f90:427-469) and, per time slice, `<dfile>NNN.bin` = a raw stream of REAL(4) data(13+eHEAT, n), zones
ordered x3 (phi) fastest (:507-571): x1, x2, x3, rho, u, u^mu(4), b^mu(4) [, kel].  This script writes the metric=0
(MKS with hslope: x1 = log r, theta = pi x2 + (1-h)/2 sin(2 pi x2), x3 = phi) form from the same analytic torus as
tools/make_harm3d_dump.py, with an electron-entropy column for the eHEAT=1 branch.

    python tools/make_iharm_dump.py OUTDIR [--n1 48 --n2 48 --n3 24] [--name dumpih] [--indf 40]
"""
import argparse
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_harm_dump import ks_metric, SPIN, GAM, HSLOPE

SEED = 20261004


MKS_SMOOTH, POLY_XT, POLY_ALPHA = 0.5, 0.82, 14.0


def theta_mmks(x1, x2, h, startx1):
    """calcthmmks (fluid_model_iharm.f90:111-122), double precision"""
    D = np.pi / (2.0 + 2.0 / (POLY_XT ** POLY_ALPHA * (1.0 + POLY_ALPHA)))
    y = 2.0 * x2 - 1.0
    thg = np.pi * x2 + 0.5 * (1.0 - h) * np.sin(2.0 * np.pi * x2)
    thj = D * y * (1.0 + (y / POLY_XT) ** POLY_ALPHA / (1.0 + POLY_ALPHA)) + np.pi / 2.0
    return thg + np.exp(-MKS_SMOOTH * (x1 - startx1)) * (thj - thg)


def make(n1=48, n2=48, n3=24, rout=40.0, a=SPIN, h=HSLOPE, seed=SEED, time=2000.0, eheat=1, b0=0.12, metric=0):
    assert n1 <= n2, "the reference's uniqx1 slice (fluid_model_iharm.f90:251) needs nx1 <= nx2"
    rh = 1.0 + np.sqrt(1.0 - a * a)
    rin = 0.98 * rh
    startx1, dx1 = np.log(rin), (np.log(rout) - np.log(rin)) / n1
    x1 = startx1 + (np.arange(n1) + 0.5) * dx1
    x2 = (np.arange(n2) + 0.5) / n2
    x3 = (np.arange(n3) + 0.5) * 2.0 * np.pi / n3
    X1, X2, X3 = np.meshgrid(x1, x2, x3, indexing="ij")
    r, ph = np.exp(X1), X3
    if metric == 0:
        th = np.pi * X2 + 0.5 * (1.0 - h) * np.sin(2.0 * np.pi * X2)
        dthdx2 = np.pi * (1.0 + (1.0 - h) * np.cos(2.0 * np.pi * X2))
        dthdx1 = np.zeros_like(th)
    else:       # MMKS: theta depends on x1 as well; derivatives by (double precision) central differences
        th = theta_mmks(X1, X2, h, startx1)
        e = 1e-6
        dthdx2 = (theta_mmks(X1, X2 + e, h, startx1) - theta_mmks(X1, X2 - e, h, startx1)) / (2 * e)
        dthdx1 = (theta_mmks(X1 + e, X2, h, startx1) - theta_mmks(X1 - e, X2, h, startx1)) / (2 * e)
    rng = np.random.default_rng(seed)
    wig = 1.0 + 0.02 * rng.standard_normal(r.shape)
    spiral = ph - 1.3 * np.sqrt(r)
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
    u = (1.0 / np.sqrt(1.0 - v2))[..., None] * (nvec + v)
    B = np.zeros_like(nvec)
    B[..., 1] = b0 * np.cos(th) / r ** 2 * (1.0 + 0.3 * np.sin(th) ** 2)
    B[..., 2] = 0.2 * b0 * np.sin(th) / r ** 3 * (1.0 + 0.2 * np.sin(spiral))
    B[..., 3] = -0.8 * b0 / r ** 2.5 * (1.0 + 0.25 * np.cos(spiral + 0.4))
    ucov = np.einsum("...ij,...j->...i", g, u)
    bt = np.einsum("...i,...i->...", B[..., 1:], ucov[..., 1:])
    b = np.zeros_like(nvec)
    b[..., 0] = bt
    b[..., 1:] = (B[..., 1:] + bt[..., None] * u[..., 1:]) / u[..., 0:1]
    rho = (r / 6.0) ** -1.5 * np.exp(-0.5 * (np.cos(th) / 0.35) ** 2) * np.exp(-(2.2 / r) ** 4) * wig * az + 1e-6
    uint = 0.06 * rho / r ** 0.9 * (1.0 + 0.3 * np.cos(th) ** 2) / (GAM - 1.0)
    # electron entropy such that theta_e = (mp/me) kel rho^(1/3) (ressler_e, fluid.f90:894-904) is ~ 3..30
    thetae = 20.0 / r ** 0.7 * (1.0 + 0.5 * np.cos(th) ** 2)
    kel = thetae / 1836.15267 / rho ** (1.0 / 3.0)

    def to_mks(w):
        o = w.copy()
        o[..., 1] = w[..., 1] / r
        o[..., 2] = (w[..., 2] - dthdx1 * o[..., 1]) / dthdx2
        return o
    um, bm = to_mks(u), to_mks(b)
    dlen = 13 + eheat
    cols = np.zeros(r.shape + (dlen,), dtype=np.float32)
    cols[..., 0], cols[..., 1], cols[..., 2] = X1, X2, X3
    cols[..., 3], cols[..., 4] = rho, uint
    cols[..., 5:9] = um
    cols[..., 9:13] = bm
    if eheat:
        cols[..., 13] = kel
    header = [time, n1, n2, n3, a, h, GAM, MKS_SMOOTH, POLY_XT, POLY_ALPHA, startx1, metric, eheat, 0, 8 + eheat, rin, rout]
    return header, cols.reshape(n1 * n2 * n3, dlen)


def write(outdir, name="dumpih", indf=40, **kw):
    """-> (dfile prefix, header file).  The reference opens trim(dfile)//'NNN'//'.bin' and hfile."""
    header, cols = make(**kw)
    os.makedirs(outdir, exist_ok=True)
    hfile = os.path.join(outdir, name + "_header")
    with open(hfile, "w") as f:
        f.write(" ".join("%.16g" % v for v in header) + "\n")
    with open(os.path.join(outdir, "%s%03d.bin" % (name, indf)), "wb") as f:
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())
    return os.path.join(outdir, name), hfile


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--n1", type=int, default=48)
    ap.add_argument("--n2", type=int, default=48)
    ap.add_argument("--n3", type=int, default=24)
    ap.add_argument("--name", default="dumpih")
    ap.add_argument("--indf", type=int, default=40)
    a = ap.parse_args()
    print(write(a.outdir, a.name, a.indf, n1=a.n1, n2=a.n2, n3=a.n3))
