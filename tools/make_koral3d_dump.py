#!/usr/bin/env python3
"""Synthetic KORAL 3-D dump: the reference's KORAL3D reader takes an 11-number text header (tcur, nx1,
nx2, nx3, a, -, r0, h, aa, bb, pp; fluid_model_koral3d.f90:357-388) and, per time slice, `<dfile>NNNN` = the same header
line followed by one 23-column text line per zone (its "shortfile" layout, :418-431), x2 (theta) fastest, then x1, then
x3: columns 4-9 grid (x1, x2, x3, r, theta, phi), 10 rho [g/cm^3], 12-15 u^mu and 16-19 b^mu in Boyer-Lindquist
components, 21 electron temperature [K], 22 Bernoulli parameter.  Coordinates: r = r0 + exp(x1),
theta = pi/2 (1 + tan(h pi (x2 - 1/2 + (1 - 2 x2)(aa + 2^pp (bb - aa)/r^pp)))/tan(h pi/2)) (transformmksh32bl :74-82).
The same analytic torus as the other synthetic dumps, in cgs, starting outside the horizon.

    python tools/make_koral3d_dump.py OUTDIR [--n1 48 --n2 48 --n3 24] [--name koral] [--indf 40]
"""
import argparse
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_harm_dump import ks_metric, SPIN

SEED = 20261005
R0, H, AA, BB, PP = 0.0, 0.9, 0.0, 0.05, 2.0


def theta_of(x2, r):
    return 0.5 * np.pi * (1.0 + np.tan(H * np.pi * (-0.5 + x2 + (1.0 - 2.0 * x2) * (AA + 2.0 ** PP * (BB - AA) / r ** PP))) / np.tan(0.5 * H * np.pi))


def make(n1=48, n2=48, n3=24, rout=40.0, a=SPIN, seed=SEED, time=2000.0, b0=0.12):
    assert n3 <= n1, "the reference's uniqx3 slice (fluid_model_koral3d.f90:113) needs nx3 <= nx1"
    rh = 1.0 + np.sqrt(1.0 - a * a)
    rin = 1.06 * rh
    x1 = np.log(rin - R0) + (np.arange(n1) + 0.5) * (np.log(rout - R0) - np.log(rin - R0)) / n1
    x2 = (np.arange(n2) + 0.5) / n2
    x3 = -np.pi + (np.arange(n3) + 0.5) * 2.0 * np.pi / n3
    X3, X1, X2 = np.meshgrid(x3, x1, x2, indexing="ij")          # flattened: x2 fastest, then x1, then x3
    r = R0 + np.exp(X1)
    th = theta_of(X2, r)
    ph = X3
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

    def ks_to_bl(w):                                          # uks2ubl (kerr.f90:131-144)
        delta = r * r - 2.0 * r + a * a
        o = w.copy()
        o[..., 0] = w[..., 0] - 2.0 * r / delta * w[..., 1]
        o[..., 3] = w[..., 3] - a / delta * w[..., 1]
        return o
    ubl, bbl = ks_to_bl(u), ks_to_bl(b)
    # torus + a tenuous wind at all latitudes (so that the disc variant's polar / Bernoulli cuts and a sigma cut remove light)
    rho_code = (r / 6.0) ** -1.5 * (np.exp(-0.5 * (np.cos(th) / 0.35) ** 2) + 0.05) * np.exp(-(2.2 / r) ** 4) * wig * az + 1e-6
    rho_cgs = 4e-17 * rho_code                                # n ~ 2e7 cm^-3 at the torus
    bscale = np.sqrt(rho_cgs.max() / rho_code.max()) * 2.998e10   # code HL field -> Gauss/sqrt(4 pi)
    te = 6e10 / r ** 0.8 * (1.0 + 0.5 * np.cos(th) ** 2)
    be = -0.08 + 0.4 * np.cos(th) ** 4 + 0.02 * np.sin(spiral)     # bound (< 0.05) in the disc, unbound towards the poles
    cols = np.zeros(r.shape + (23,))
    cols[..., 3], cols[..., 4], cols[..., 5] = X1, X2, X3
    cols[..., 6], cols[..., 7], cols[..., 8] = r, th, ph
    cols[..., 9] = rho_cgs
    cols[..., 11:15] = ubl
    cols[..., 15:19] = bbl * bscale
    cols[..., 20] = te
    cols[..., 21] = be
    header = [time, n1, n2, n3, a, 0.0, R0, H, AA, BB, PP]
    return header, cols.reshape(n1 * n2 * n3, 23)


def write(outdir, name="koral", indf=40, **kw):
    """-> (dfile prefix, header file).  The reference opens trim(dfile)//I4.4(indf) and hfile."""
    header, cols = make(**kw)
    os.makedirs(outdir, exist_ok=True)
    hline = " ".join("%.16g" % v for v in header) + "\n"
    hfile = os.path.join(outdir, name + "_header")
    open(hfile, "w").write(hline)
    with open(os.path.join(outdir, "%s%04d" % (name, indf)), "w") as f:
        f.write(hline)
        np.savetxt(f, cols, fmt="%.9e")
    return os.path.join(outdir, name), hfile


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--n1", type=int, default=48)
    ap.add_argument("--n2", type=int, default=48)
    ap.add_argument("--n3", type=int, default=24)
    ap.add_argument("--name", default="koral")
    ap.add_argument("--indf", type=int, default=40)
    a = ap.parse_args()
    print(write(a.outdir, a.name, a.indf, n1=a.n1, n2=a.n2, n3=a.n3))
