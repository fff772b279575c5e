"""Test infrastructure (lives under tests/ because it drives oracle/_ref).  Generate tests/golden/*.npz from the COMPILED, UNMODIFIED reference (oracle/_ref).

Run in the build container only (needs /root/reference built by oracle/build_ref.sh); the
fixtures it writes are data (inputs + expected outputs) and are committed.  The GPU box only
ever reads the .npz files.
"""
import os, sys, pickle
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))     # synthetic dump writers
from oracle import refbind as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
os.makedirs(G, exist_ok=True)
FREQ = 3.45e11
FFJET = dict(fname="FFJET", ename="POLSYNCHPL", spin=0.998, standard=1, uout=0.01, mbh=3.4e9, mu0=0.906,
             gridvals=(-40., 20., -20., 40.), ftscl=2.0, frscl=70.0, gmin=100., gmax=1e5, p1=3.5, p2=3.5)


def leafs():
    rng = np.random.default_rng(20261002)
    xyzp = rng.uniform(1e-3, 8.0, size=(200, 4))
    xyzp[:20, 0] = 0.0                      # complete integrals rf(0,m,1) etc.
    out = dict(xyzp=xyzp,
               rf=np.array([R.rf(*v[:3]) for v in xyzp]), rd=np.array([R.rd(*v[:3]) for v in xyzp]),
               rc=np.array([R.rc(v[1], v[2]) for v in xyzp]), rcn=np.array([R.rc(v[1] + v[2], -v[2]) for v in xyzp]),
               rj=np.array([R.rj(*v) for v in xyzp]), rjn=np.array([R.rj(v[0], v[1], v[2], -v[3]) for v in xyzp]))
    um = np.stack([rng.uniform(-6, 6, 100), rng.uniform(-0.9, 0.999, 100)], 1)
    um[:5, 1] = 0.0
    out["um"] = um
    out["sncndn"] = np.array([R.sncndn(u, m) for u, m in um])
    np.savez_compressed(os.path.join(G, "leaf_special.npz"), **out)


def coeffs():
    rng = np.random.default_rng(7)
    n = 300
    nu = 10 ** rng.uniform(9, 15, n); nnth = 10 ** rng.uniform(-2, 6, n); b = 10 ** rng.uniform(-3, 3, n)
    th = rng.uniform(0.05, np.pi - 0.05, n); p = rng.choice([2.5, 3.5, 5.5], n)
    gmin = rng.choice([1.0, 10.0, 100.0], n).astype(float); gmin[gmin == 1.0] = 1.5
    epl = R.polsynchpl(nu, nnth, b, th, p, gmin, 1e5)
    dens = 10 ** rng.uniform(2, 8, n); t = 10 ** rng.uniform(8.5, 12, n)
    eth = R.polsynchth(nu, dens, b, t, th)
    np.savez_compressed(os.path.join(G, "coeffs.npz"), nu=nu, nnth=nnth, b=b, th=th, p=p, gmin=gmin, gmax=1e5,
                        polsynchpl=epl, dens=dens, t=t, polsynchth=eth)


def integrators():
    # the closed-form cases of unit_tests_integration.py:39-150 evaluated by the Fortran integrate()
    n = 600
    x = np.linspace(3.0, 0.0, n)           # decreasing like g%lambda
    cases = {}
    def run(name, j, K):
        tau = np.zeros(n)
        tau[1:] = np.cumsum((x[1:] - x[:-1]) * 0.5 * (np.abs(K[1:, 0]) + np.abs(K[:-1, 0])))
        tau = -tau
        cases[name + "_j"] = j; cases[name + "_K"] = K; cases[name + "_tau"] = tau
        cases[name + "_delo"] = R.integrate(1, x, j, K, tau)
        cases[name + "_formal"] = R.integrate(2, x, j, K, tau)
    one = np.ones(n)
    z = np.zeros(n)
    run("faraday_rot", np.stack([one, 0.7 * one, z, z], 1), np.stack([z + 1e-30, z, z, z, z, z, 10.0 * one], 1))
    run("rot_conv", np.stack([one, 0.7 * one, z, z], 1), np.stack([z + 1e-30, z, z, z, 4.0 * one, z, 10.0 * one], 1))
    run("full_rho", np.stack([one, 0.3 * one, 0.2 * one, 0.1 * one], 1),
        np.stack([z + 1e-30, z, z, z, 4.0 * one, 2.0 * one, 10.0 * one], 1))
    run("abs_IQ", np.stack([one, 0.7 * one, z, z], 1), np.stack([2.0 * one, 0.8 * one, z, z, z, z, z], 1))
    s = np.sin(3 * x) ** 2
    run("varying", np.stack([1 + s, 0.5 * s, 0.2 * one, -0.1 * s], 1),
        np.stack([1.5 + s, 0.4 * s, 0.1 * one, 0.05 * one, 3.0 * s, one, 6.0 * one], 1))
    run("thin", 1e-3 * np.stack([1 + s, 0.5 * s, 0.2 * one, -0.1 * s], 1),
        1e-3 * np.stack([1.5 + s, 0.4 * s, 0.1 * one, 0.05 * one, 3.0 * s, one, 6.0 * one], 1))
    np.savez_compressed(os.path.join(G, "integrators.npz"), x=x, **cases)


def ffjet(nx, nup):
    out = {}
    for iname in ("formal", "delo"):
        r = R.Ref(nn=(nx, nx, nup), iname=iname, **FFJET)
        pix, npts, _ = r.trace(1, nx * nx, [FREQ], nthreads=8)
        out["ivals_" + iname] = pix[0]
        out["npts"] = npts
    cam = r.camera()
    for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf", "tpr"):
        out["cam_" + k] = cam[k]
    out["cam_scal"] = np.array([cam[k] for k in ("u0", "uout", "offset", "mu0", "a", "kext", "next", "nup")], dtype=float)
    # geodesic integers for every ray + full per-stage dumps for a few
    tpm = np.zeros((nx * nx, nup + 64), dtype=np.int8); tpr = np.zeros_like(tpm); ncase = np.zeros(nx * nx, dtype=np.int8)
    for i in range(1, nx * nx + 1):
        g = r.geokerr_ray(i)
        ncase[i - 1] = g["ncase"]
        tpm[i - 1, :g["n"]] = g["tpm"]; tpr[i - 1, :g["n"]] = g["tpr"]
    out["tpmi"] = tpm; out["tpri"] = tpr; out["ncase"] = ncase
    np.savez_compressed(os.path.join(G, "ffjet_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ, **out)
    rays = {}
    pick = [1, nx // 2 * nx + nx // 2, nx // 2 * nx + nx // 2 + 1, nx * nx, nx * (nx // 3) + 2 * nx // 3, 7, nx * nx // 2 - 3]
    cap = np.where(out["npts"] == nup)[0]
    if len(cap):
        pick += [int(cap[0]) + 1, int(cap[len(cap) // 2]) + 1]
    r = R.Ref(nn=(nx, nx, nup), iname="formal", **FFJET)
    for i in sorted(set(pick)):
        d = r.ray_dump(i, FREQ)
        g = r.geokerr_ray(i)
        for k, v in d.items():
            rays["ray%d_%s" % (i, k)] = np.asarray(v)
        for k in ("u", "mu", "dt", "dphi", "lam"):
            rays["ray%d_raw_%s" % (i, k)] = g[k]
        rays["ray%d_ncase" % i] = np.array(g["ncase"])
    rays["rays"] = np.array(sorted(set(pick)))
    np.savez_compressed(os.path.join(G, "ffjet_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ, **rays)


SPHACC = dict(fname="SPHACC", ename="POLSYNCHTH", spin=0.0, standard=1, uout=0.04, mbh=10.0, mu0=0.5,
              gridvals=(-25., 25., -25., 25.))
FREQ_SPHACC = 2.41e11


def sphacc(nx, nup):
    # BASELINE config 2 (inputs_sphacc.in) on a smaller camera.  nthreads=1 ON PURPOSE: the reference's
    # SPHACC path is not thread-safe (module-level state in fluid_model_sphacc.f90 is not threadprivate):
    # two 8-thread runs of the compiled reference differ from each other in ~3 % of the pixels by up to
    # 3 % of the Stokes peak, the 1-thread run is reproducible.
    out = {}
    for iname in ("formal", "delo"):
        r = R.Ref(nn=(nx, nx, nup), iname=iname, **SPHACC)
        pix, npts, _ = r.trace(1, nx * nx, [FREQ_SPHACC], nthreads=1)
        out["ivals_" + iname] = pix[0]
        out["npts"] = npts
    np.savez_compressed(os.path.join(G, "sphacc_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_SPHACC, **out)
    rays = {}
    pick = sorted(set([1, nx // 2 * nx + nx // 2 + 1, nx * (nx // 3) + 2 * nx // 3, nx * nx,
                       int(np.where(out["npts"] == nup)[0][0]) + 1]))
    r = R.Ref(nn=(nx, nx, nup), iname="formal", **SPHACC)
    for i in pick:
        d = r.ray_dump(i, FREQ_SPHACC)
        for k, v in d.items():
            rays["ray%d_%s" % (i, k)] = np.asarray(v)
        rays["ray%d_ncase" % i] = np.array(r.geokerr_ray(i)["ncase"])
    rays["rays"] = np.array(pick)
    np.savez_compressed(os.path.join(G, "sphacc_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_SPHACC, **rays)


HARM = dict(fname="HARM", ename="POLSYNCHTH", spin=0.9375, standard=1, uout=0.04, mbh=4e6, mdot=1.57e15, mu0=0.6428,
            gridvals=(-13., 13., -13., 13.), muval=0.25, gmin=1., dfile="dump", hfile="dump040", findf=40, fnt=1)
FREQ_HARM = 2.3e11


def harm(nx, nup, nzone=64):
    # BASELINE config 4 on the synthetic HARM-format dump of tools/make_harm_dump.py (dump040 is not shipped
    # with the reference: parity on real GRMHD data stays unpinned)
    import hashlib
    import tempfile
    import make_harm_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = make_harm_dump.write(td, n=nzone)
        out["dump_sha256"] = np.array(hashlib.sha256(open(path, "rb").read()).hexdigest())
        for iname in ("formal", "delo"):
            r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **HARM)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=8)
            out["ivals_" + iname] = pix[0]
            out["npts"] = npts
        rays = {}
        pick = sorted(set([3, nx // 2 * nx + nx // 2, nx * 10 + 30, nx * 30 + 12, nx * nx - 4,
                           int(np.where(npts == nup)[0][0]) + 1]))
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **HARM)
        for i in pick:
            d = r.ray_dump(i, FREQ_HARM)
            for k, v in d.items():
                rays["ray%d_%s" % (i, k)] = np.asarray(v)
            rays["ray%d_ncase" % i] = np.array(r.geokerr_ray(i)["ncase"])
        rays["rays"] = np.array(pick)
    np.savez_compressed(os.path.join(G, "harm_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, nzone=nzone, **out)
    np.savez_compressed(os.path.join(G, "harm_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, nzone=nzone, **rays)


def harm_config4_zones(nx=64, nup=400, nzone=256):
    # BASELINE config 4's dump size (256 x 256 zones, the size bench.py and test_config4_full_size_properties use) on a 64 x 64 x 400
    # camera of the same field of view: the compiled reference's image, so that the full-size test's flux and a 64 x 64 device image
    # are held to the REFERENCE on that dump, not to the device's own small camera
    import hashlib
    import tempfile
    import make_harm_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = make_harm_dump.write(td, n=nzone)
        out["dump_sha256"] = np.array(hashlib.sha256(open(path, "rb").read()).hexdigest())
        for iname in ("formal", "delo"):
            r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **HARM)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=8)
            out["ivals_" + iname] = pix[0]
            out["npts"] = npts
    np.savez_compressed(os.path.join(G, "harm_zones%d_%dx%dx%d.npz" % (nzone, nx, nx, nup)), freq=FREQ_HARM, nzone=nzone, **out)


def nvals1():
    # nvals=1: emissivity/absorption of Stokes I only and the quadrature integrator (radtrans_integrate.f90:878)
    out = {}
    r = R.Ref(nn=(32, 32, 100), iname="formal", nvals=1, **FFJET)
    out["ffjet_ivals"], out["ffjet_npts"], _ = r.trace(1, 1024, [FREQ], nthreads=8)
    r = R.Ref(nn=(32, 32, 100), iname="delo", nvals=1, **SPHACC)
    out["sphacc_ivals"], out["sphacc_npts"], _ = r.trace(1, 1024, [FREQ_SPHACC], nthreads=1)
    # SYNCHPL (synchpl, polsynchemis.f90:633-698): the Stokes-I half of POLSYNCHPL, an nvals = 1 emissivity
    r = R.Ref(nn=(32, 32, 100), iname="formal", nvals=1, **dict(FFJET, ename="SYNCHPL"))
    out["ffjet_synchpl_ivals"], _, _ = r.trace(1, 1024, [FREQ], nthreads=8)
    np.savez_compressed(os.path.join(G, "nvals1_32x32x100.npz"), **out)


def extra():
    # extra=1: Stokes + the 19 ray diagnostics (grtrans_driver.f90:230-292) through oracle/ref_driver.f90:trace_pixel_extra
    out = {}
    r = R.Ref(nn=(32, 32, 100), iname="formal", **FFJET)
    out["ffjet_vals"], out["ffjet_npts"] = r.trace_extra(1, 1024, FREQ, nthreads=8)
    r = R.Ref(nn=(32, 32, 100), iname="formal", **SPHACC)
    out["sphacc_vals"], out["sphacc_npts"] = r.trace_extra(1, 1024, FREQ_SPHACC, nthreads=1)
    np.savez_compressed(os.path.join(G, "extra_32x32x100.npz"), **out)


def thindisk():
    # BASELINE config 1 = the reference's harm.in.dist problem: standard=2, nn=[100,100,1], THINDISK + BB, nvals=1, fmin=1e17,
    # mbh=10, fmdot=0.1 (iname is irrelevant: one point per ray, no integrator runs); two of its ten mu0 values
    out = {}
    kw = dict(standard=2, spin=0.998, uout=1e-4, gridvals=(-25., 25., -25., 25.), nn=(100, 100, 1), fname="THINDISK",
              ename="BB", nvals=1, iname="lsoda", mbh=10., fmdot=0.1)
    for mu0 in (0.1, 0.7):
        r = R.Ref(mu0=mu0, **kw)
        pix, npts, _ = r.trace(1, 10000, [1e17], nthreads=8)
        out["ivals_mu%02d" % int(mu0 * 10)] = pix[0]
        out["npts_mu%02d" % int(mu0 * 10)] = npts
    np.savez_compressed(os.path.join(G, "thindisk_100x100x1.npz"), **out)


MOVIE_SEEDS = {40: 20261002, 41: 20261003, 43: 20261005}


def write_movie_dumps(outdir, nzone=64):
    import make_harm_dump
    return {k: make_harm_dump.write(outdir, n=nzone, name="dump%03d" % k, seed=s) for k, s in MOVIE_SEEDS.items()}


def harm_movie():
    # nt=3 movie frames, dt=30 with HARM's default tstep=20 (one slice loaded): advance_harm_timestep
    # (fluid_model_harm.f90:511-521) steps the dump index 40 -> 41 -> 43 (the 10 M left over accumulates)
    import hashlib
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_movie_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        r = R.Ref(nn=(32, 32, 100), iname="formal", datadir=td, **HARM)
        frames = []
        for l in range(3):
            pix, npts, _ = r.trace(1, 1024, [FREQ_HARM], nthreads=8)
            frames.append(pix[0])
            if l < 2:
                r.advance_timestep(30.0)
        out["ivals"] = np.stack(frames); out["npts"] = npts
    np.savez_compressed(os.path.join(G, "harm_movie_32x32x100.npz"), freq=FREQ_HARM, dt=30.0, **out)


SLOW_DUMPS = {41: (20261003, 2040.0), 40: (20261002, 2020.0), 39: (20261001, 2000.0)}


def write_slow_dumps(outdir, nzone=64):
    import make_harm_dump
    return {k: make_harm_dump.write(outdir, n=nzone, name="dump%03d" % k, seed=s, time=t) for k, (s, t) in SLOW_DUMPS.items()}


def harm_slowlight():
    # three loaded slices 20 M apart (fnt=3, findf=41): each point reads the dump its Kerr-Schild look-back time selects
    import hashlib
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_slow_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        kw = dict(HARM); kw.update(hfile="dump041", findf=41, fnt=3)
        for iname in ("formal", "delo"):
            r = R.Ref(nn=(32, 32, 100), iname=iname, datadir=td, **kw)
            pix, npts, _ = r.trace(1, 1024, [FREQ_HARM], nthreads=8)
            out["ivals_" + iname] = pix[0]; out["npts"] = npts
        kw1 = dict(kw); kw1["fnt"] = 1
        r = R.Ref(nn=(32, 32, 100), iname="formal", datadir=td, **kw1)
        out["ivals_formal_fastlight"] = r.trace(1, 1024, [FREQ_HARM], nthreads=8)[0][0]
        # nload > 1: per-ray time origin table + mu-fill off (pgrtrans.f90:172-189)
        r = R.Ref(nn=(32, 32, 100), iname="formal", datadir=td, **kw)
        out["t0_nload"] = r.nload_t0()
        pix, npts, _ = r.trace(1, 1024, [FREQ_HARM], nthreads=8)
        out["ivals_formal_nload"] = pix[0]; out["npts_nload"] = npts
    np.savez_compressed(os.path.join(G, "harm_slowlight_32x32x100.npz"), freq=FREQ_HARM, **out)


HARM3D = dict(fname="HARM3D", ename="POLSYNCHTH", spin=0.9375, standard=1, uout=0.04, mbh=4e6, mdot=1e20, mu0=0.5,
              gridvals=(-13., 13., -13., 13.), muval=0.25, gmin=1., dfile="dump3d", hfile="dump3d_header", findf=40, fnt=1)
HARM3D_ZONES = dict(n1=48, n2=48, n3=24)


def harm3d(nx=40, nup=200):
    # the 3-D HARM sampler on the synthetic dump of tools/make_harm3d_dump.py (no HARM3D data ships with
    # the reference).  Two runs: sigcut at its default (1e10: no cut) and sigcut=1e-3 (the magnetisation cut zeroes the
    # funnel), formal and delo; per-stage dumps of a few rays.
    import hashlib
    import tempfile
    import make_harm3d_dump
    out, rays = {}, {}
    with tempfile.TemporaryDirectory() as td:
        dfile, hfile = make_harm3d_dump.write(td, **HARM3D_ZONES)
        out["dump_sha256"] = np.array(hashlib.sha256(open(dfile + "040.bin", "rb").read()).hexdigest())
        for iname in ("formal", "delo"):
            r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **HARM3D)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["ivals_" + iname] = pix[0]
            out["npts"] = npts
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, sigcut=1e-3, **HARM3D)
        pix, _, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
        out["ivals_formal_sigcut"] = pix[0]
        pick = sorted(set([3, nx // 2 * nx + nx // 2, nx * 10 + 30, nx * 30 + 12, nx * nx - 4, nx * 20 + 5]))
        for i in pick:
            d = r.ray_dump(i, FREQ_HARM)
            for k, v in d.items():
                rays["ray%d_%s" % (i, k)] = np.asarray(v)
        rays["rays"] = np.array(pick)
        # extra=1 on the 3-D sampler: the phi-weighted diagnostics read the same azimuth the sampler does
        r = R.Ref(nn=(24, 24, 120), iname="formal", datadir=td, **HARM3D)
        out["extra_vals_24x24x120"], out["extra_npts_24x24x120"] = r.trace_extra(1, 576, FREQ_HARM, nthreads=1)
    zones = np.array([HARM3D_ZONES[k] for k in ("n1", "n2", "n3")])
    np.savez_compressed(os.path.join(G, "harm3d_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, zones=zones, **out)
    np.savez_compressed(os.path.join(G, "harm3d_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, zones=zones, sigcut=1e-3, **rays)


HYBRID = dict(gmin=50., gmax=1e5, p1=3.5, p2=3.5, jetalpha=0.02, mdot=1e19)


def hybrid(nx=32, nup=150):
    # thermal + power-law electron populations (HYBRIDTHPL / HYBRIDTH / HYBRIDPL, emis.f90:473-490) on the two GRMHD
    # samplers: 2-D HARM (non-thermal density = thermal density, fluid.f90:972) and a magnetised 3-D HARM dump
    # (non-thermal electrons where b^2/rho > 1, fluid.f90:914-923)
    import tempfile
    import make_harm_dump
    import make_harm3d_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        make_harm3d_dump.write(td, b0=2.5, **HARM3D_ZONES)
        make_harm_dump.write(td, n=64)
        for en in ("HYBRIDTHPL", "HYBRIDTH", "HYBRIDPL"):
            kw = dict(HARM3D); kw.update(HYBRID); kw.update(ename=en)
            r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["harm3d_" + en] = pix[0]; out["harm3d_npts"] = npts
        kw = dict(HARM3D); kw.update(HYBRID); kw.update(ename="HYBRIDTHPL")
        r = R.Ref(nn=(nx, nx, nup), iname="delo", datadir=td, **kw)
        out["harm3d_HYBRIDTHPL_delo"] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        for en in ("HYBRIDTHPL", "HYBRIDTH"):
            kw = dict(HARM); kw.update(ename=en, gmin=10., gmax=1e5, p1=3.5, p2=3.5)
            r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["harm_" + en] = pix[0]; out["harm_npts"] = npts
    np.savez_compressed(os.path.join(G, "hybrid_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, b0=2.5, **out)


# read_inputs.f90:95-96: the six tabulated weights are REAL(4) literals stored in REAL(8)
MAXJUTT_W6 = tuple(float(np.float32(v)) for v in (0.10714295, 0.4459917, 1.34219997, 2.17450422, 1.36515333, 0.52892829))


def maxjutt(nx=32, nup=150):
    # MAXJUTT (calc_maxjutt.f90): a weighted sum of thermal components; delta=2 with the six tabulated weights on the 3-D
    # dump, delta=1.5 with two weights on the 2-D dump
    import tempfile
    import make_harm_dump
    import make_harm3d_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        make_harm3d_dump.write(td, **HARM3D_ZONES)
        make_harm_dump.write(td, n=64)
        kw = dict(HARM3D); kw.update(ename="MAXJUTT", otherargs=(2.0,) + MAXJUTT_W6)
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)      # (the reference's DELO overflows on this problem)
        pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
        out["harm3d_w6_formal"] = pix[0]; out["harm3d_npts"] = npts
        # MAXCOMP (calc_maxcomp.f90): emission of the selected component, absorption / Faraday terms of all of them
        kw = dict(HARM3D); kw.update(ename="MAXCOMP", otherargs=(2.0, 2.0, 1.0, 0.5, 0.25))
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
        out["harm3d_maxcomp_sel2"] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        kw = dict(HARM); kw.update(ename="MAXJUTT", otherargs=(1.5, 1.0, 0.5))
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
        pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
        out["harm_w2_formal"] = pix[0]; out["harm_npts"] = npts
    np.savez_compressed(os.path.join(G, "maxjutt_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, **out)


IHARM = dict(fname="IHARM", ename="POLSYNCHTH", spin=0.9375, standard=1, uout=0.04, mbh=4e6, mdot=1e20, mu0=0.5,
             gridvals=(-13., 13., -13., 13.), muval=0.25, gmin=1., dfile="dumpih", hfile="dumpih_header", findf=40, fnt=1)
IHARM_ZONES = dict(n1=48, n2=48, n3=24)


def iharm(nx=36, nup=160):
    # iharm-format dumps, metric 0 (tools/make_iharm_dump.py; no iharm data ships with the reference): the three electron
    # models of convert_fluidvars_iharm (gmin = 1: constant ratio, gmin = 3: Moscibrodzka beta-dependent, gmin = -1: the
    # dump's electron entropy), delo, and the magnetisation cut; per-stage dumps of a few rays
    import hashlib
    import tempfile
    import make_iharm_dump
    out, rays = {}, {}
    with tempfile.TemporaryDirectory() as td:
        dfile, hfile = make_iharm_dump.write(td, **IHARM_ZONES)
        out["dump_sha256"] = np.array(hashlib.sha256(open(dfile + "040.bin", "rb").read()).hexdigest())
        for key, kw in (("formal_gmin1", dict(iname="formal")), ("delo_gmin1", dict(iname="delo")), ("formal_gmin3", dict(iname="formal", gmin=3.)),
                        ("formal_kel", dict(iname="formal", gmin=-1.)), ("formal_sigcut", dict(iname="formal", sigcut=1e-3))):
            r = R.Ref(nn=(nx, nx, nup), datadir=td, **dict(IHARM, **kw))
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["ivals_" + key] = pix[0]; out["npts"] = npts
        # hybrid and MAXJUTT on this sampler (non-thermal electrons where b^2/rho > 1: the scaled zones near the horizon)
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **dict(IHARM, ename="HYBRIDTHPL", gmin=2., gmax=1e5, p1=3.5, p2=3.5, jetalpha=0.02))
        out["ivals_hybridthpl"] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **dict(IHARM, ename="MAXJUTT", otherargs=(2.0, 1.0, 0.5, 0.25)))
        out["ivals_maxjutt"] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        # MMKS (header metric 1): theta depends on x1 as well; vectors go through central differences of the REAL(4) theta
        import os as _os
        tdm = _os.path.join(td, "mmks")
        dfm, _ = make_iharm_dump.write(tdm, metric=1, **IHARM_ZONES)
        out["mmks_dump_sha256"] = np.array(hashlib.sha256(open(dfm + "040.bin", "rb").read()).hexdigest())
        for key, kw in (("mmks_formal_gmin3", dict(iname="formal", gmin=3.)), ("mmks_delo_kel", dict(iname="delo", gmin=-1.))):
            r = R.Ref(nn=(nx, nx, nup), datadir=tdm, **dict(IHARM, **kw))
            out["ivals_" + key] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **dict(IHARM, gmin=3.))
        pick = sorted(set([3, nx // 2 * nx + nx // 2, nx * 10 + 30, nx * 30 + 12, nx * nx - 4, nx * 20 + 5]))
        for i in pick:
            d = r.ray_dump(i, FREQ_HARM)
            for k, v in d.items():
                rays["ray%d_%s" % (i, k)] = np.asarray(v)
        rays["rays"] = np.array(pick)
    np.savez_compressed(os.path.join(G, "iharm_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, **out)
    np.savez_compressed(os.path.join(G, "iharm_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, gmin=3.0, **rays)


KORAL3D = dict(fname="KORAL3D", ename="POLSYNCHTH", spin=0.9375, standard=1, uout=0.04, mbh=4e6, mu0=0.5, gridvals=(-13., 13., -13., 13.),
               muval=0.25, gmin=-1., dfile="koral", hfile="koral_header", findf=40, fnt=1, ftscl=1.5, sigcut=1e10)
KORAL3D_ZONES = dict(n1=48, n2=48, n3=24)


def koral3d(nx=36, nup=160):
    # KORAL 3-D dumps (tools/make_koral3d_dump.py; no KORAL data ships with the reference): electron temperature from the
    # dump (gmin = -1) and from charles_e (gmin = 3), delo, the disc variant (polar and Bernoulli cuts), a sigma cut
    import hashlib
    import tempfile
    import make_koral3d_dump
    out, rays = {}, {}
    with tempfile.TemporaryDirectory() as td:
        dfile, hfile = make_koral3d_dump.write(td, **KORAL3D_ZONES)
        out["dump_sha256"] = np.array(hashlib.sha256(open(dfile + "0040", "rb").read()).hexdigest())
        for key, kw in (("formal_te", dict(iname="formal")), ("delo_te", dict(iname="delo")), ("formal_gmin3", dict(iname="formal", gmin=3.)),
                        ("formal_disk", dict(iname="formal", fname="KORAL3D_DISK")), ("formal_sigcut", dict(iname="formal", sigcut=0.02))):
            r = R.Ref(nn=(nx, nx, nup), datadir=td, **dict(KORAL3D, **kw))
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["ivals_" + key] = pix[0]; out["npts"] = npts
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **dict(KORAL3D, fname="KORAL3D_DISK"))
        pick = sorted(set([3, nx // 2 * nx + nx // 2, nx * 10 + 30, nx * 30 + 12, nx * nx - 4, nx * 20 + 5, nx * 17 + 18]))
        for i in pick:
            d = r.ray_dump(i, FREQ_HARM)
            for k, v in d.items():
                rays["ray%d_%s" % (i, k)] = np.asarray(v)
        rays["rays"] = np.array(pick)
    np.savez_compressed(os.path.join(G, "koral3d_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, **out)
    np.savez_compressed(os.path.join(G, "koral3d_rays_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, **rays)


def powerlaw_grmhd(nx=28, nup=120):
    # r05 (VERDICT r04 missing #3): the reference dispatches the fluid model and the emissivity independently (grtrans_driver.f90:122-159,
    # emis.f90:461-571): POLSYNCHPL on the GRMHD samplers -- non-thermal density = the thermal one (HARM 2-D, fluid.f90:972), from
    # nonthermale_b2 (HARM3D / IHARM: where b^2/rho > 1) or from the dump (KORAL3D).  formal, delo and the nvals = 1 quadrature each.
    import tempfile
    import make_harm_dump
    import make_harm3d_dump
    import make_iharm_dump
    out = {}
    PL = dict(ename="POLSYNCHPL", gmin=50., gmax=1e5, p1=3.5, p2=3.5)
    with tempfile.TemporaryDirectory() as td:
        make_harm_dump.write(td, n=64)
        make_harm3d_dump.write(td, b0=2.5, **HARM3D_ZONES)
        make_iharm_dump.write(td, b0=2.5, **IHARM_ZONES)
        # (KORAL3D + POLSYNCHPL, SPHACC + POLSYNCHPL, FFJET + POLSYNCHTH: the reference's unit conversions set the density the emissivity
        #  needs to zero there -- fluid.f90:1150, :1257, :1171 -- its image is identically zero: no kernel, refused)
        for tag, base, extra in (("harm", HARM, {}), ("harm3d", HARM3D, dict(jetalpha=0.02, mdot=1e19)), ("iharm", IHARM, dict(jetalpha=0.02, mdot=1e19))):
            for iname, nvals in (("formal", 4), ("delo", 4), ("formal", 1)):
                kw = dict(base); kw.update(PL); kw.update(extra)
                if nvals == 1:
                    kw["nvals"] = 1
                r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **kw)
                pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
                out["%s_%s_nvals%d" % (tag, iname, nvals)] = pix[0]; out[tag + "_npts"] = npts
    np.savez_compressed(os.path.join(G, "powerlaw_grmhd_%dx%dx%d.npz" % (nx, nx, nup)), freq=FREQ_HARM, b0=2.5, **out)


def unpolarized(nx=32, nup=120):
    # the scalar emissivities of emis.f90:532-548: angle-averaged thermal synchrotron (SYNCHTHAV, with and without its LTE
    # absorption), free-free (BREMS, SYNCHTHBREMS: nvals = 1 only -- the reference leaves nine of their eleven columns
    # unassigned) and RHO, on the 2-D HARM dump, the 3-D one and the Bondi flow.  nvals = 1 throughout: with nvals = 4 the
    # reference returns NaN (formal) or 1e9-sized garbage (delo) for these, its polarized branch never being set up.
    import tempfile
    import make_harm_dump
    import make_harm3d_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        make_harm_dump.write(td, n=64)
        make_harm3d_dump.write(td, **HARM3D_ZONES)
        for en, nv, iname in (("SYNCHTHAV", 1, "formal"), ("SYNCHTHAVNOABS", 1, "formal"),
                              ("BREMS", 1, "formal"), ("SYNCHTHBREMS", 1, "formal"), ("RHO", 1, "formal")):
            kw = dict(HARM); kw.update(ename=en, nvals=nv)
            r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **kw)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["harm_%s_%d_%s" % (en, nv, iname)] = pix[0]; out["harm_npts"] = npts
        kw = dict(HARM3D); kw.update(ename="SYNCHTHBREMS", nvals=1)
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
        out["harm3d_SYNCHTHBREMS_1_formal"], out["harm3d_npts"], _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
        kw = dict(SPHACC); kw.update(ename="SYNCHTHAV", nvals=1)
        r = R.Ref(nn=(nx, nx, nup), iname="formal", **kw)
        out["sphacc_SYNCHTHAV_1_formal"], out["sphacc_npts"], _ = r.trace(1, nx * nx, [FREQ_SPHACC], nthreads=1)
    np.savez_compressed(os.path.join(G, "unpolarized_%dx%dx%d.npz" % (nx, nx, nup)), **out)


def sympol(nx=32, nup=120):
    # POLSYNCHSYMTH (sympolemisth, polsynchemis.f90:915-1010): Pandya et al. fitting functions, polarized
    import tempfile
    import make_harm_dump
    import make_harm3d_dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        make_harm_dump.write(td, n=64)
        make_harm3d_dump.write(td, **HARM3D_ZONES)
        for iname in ("formal", "delo"):
            kw = dict(HARM); kw.update(ename="POLSYNCHSYMTH")
            r = R.Ref(nn=(nx, nx, nup), iname=iname, datadir=td, **kw)
            pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
            out["harm_" + iname] = pix[0]; out["harm_npts"] = npts
        kw = dict(HARM); kw.update(ename="POLSYNCHSYMTH", nvals=1)
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
        out["harm_nvals1"] = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)[0][0]
        kw = dict(HARM3D); kw.update(ename="POLSYNCHSYMTH")
        r = R.Ref(nn=(nx, nx, nup), iname="formal", datadir=td, **kw)
        pix, npts, _ = r.trace(1, nx * nx, [FREQ_HARM], nthreads=1)
        out["harm3d_formal"] = pix[0]; out["harm3d_npts"] = npts
    kw = dict(SPHACC); kw.update(ename="POLSYNCHSYMTH")
    r = R.Ref(nn=(nx, nx, nup), iname="formal", **kw)
    pix, npts, _ = r.trace(1, nx * nx, [FREQ_SPHACC], nthreads=1)
    out["sphacc_formal"] = pix[0]; out["sphacc_npts"] = npts
    np.savez_compressed(os.path.join(G, "sympol_%dx%dx%d.npz" % (nx, nx, nup)), **out)


def reference_pickle():
    g = pickle.load(open("/root/reference/test_grtrans_ffjet.p", "rb"), encoding="latin1")
    np.savez_compressed(os.path.join(G, "test_grtrans_ffjet_lsoda.npz"), ivals=g[:, :, 0].astype(np.float32))
    # the reference's first regression problem (run_grtrans_test_problems_public.py:27-52): SPHACC + SYNCHTHAV, nvals = 1,
    # 25 frequencies 1e8..1e15 Hz, a 10000 x 1 radial strip; intensity at the 15th frequency and the spectrum (LSODA)
    gi = pickle.load(open("/root/reference/test_grtrans_sphacc_intensity.p", "rb"), encoding="latin1")
    gs = pickle.load(open("/root/reference/test_grtrans_sphacc_spectrum.p", "rb"), encoding="latin1")
    np.savez_compressed(os.path.join(G, "test_grtrans_sphacc_lsoda.npz"), intensity=np.asarray(gi), spectrum=np.asarray(gs))


# ---------------------------------------------------------------------------------------------------------------
# Fixtures made by the reference's OWN driver chain: grtrans_run -> read_inputs -> grtrans_main -> grtrans_driver ->
# kwrite_raytrace_camera (oracle/refmain.py; nothing restated).  FFJET_KW / SPHACC_KW are grtrans_batch.py kwargs.
FFJET_KW = dict(fname='FFJET', nfreq=1, nmu=1, fmin=FREQ, fmax=FREQ, ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, uout=0.01,
                mbh=3.4e9, mdotmin=1.57e15, mdotmax=1.57e15, nn=[100, 100, 400], gridvals=[-40., 20., -20., 40.], mumin=.906,
                mumax=.906, ntscl=2., nrscl=70., gmin=100., gmax=1e5, p1=3.5, p2=3.5, iname='formal')
SPHACC_KW = dict(fname='SPHACC', nfreq=1, nmu=1, fmin=FREQ_SPHACC, fmax=FREQ_SPHACC, ename='POLSYNCHTH', nvals=4, spin=0., standard=1,
                 uout=0.04, mbh=10., nn=[64, 64, 200], gridvals=[-25., 25., -25., 25.], mumin=.5, mumax=.5, iname='formal')


def extra_main():
    # extra=1 through grtrans_driver.f90:230-292 itself: Stokes + 19 diagnostics (nup > 1)
    from oracle import refmain
    out = {}
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], extra=1), nthreads=8)
    out["ffjet_vals"] = r["ivals"][0]
    r = refmain.run(dict(SPHACC_KW, nn=[32, 32, 100], extra=1), nthreads=1)
    out["sphacc_vals"] = r["ivals"][0]
    old = np.load(os.path.join(G, "extra_32x32x100.npz"))
    out["ffjet_npts"], out["sphacc_npts"] = old["ffjet_npts"], old["sphacc_npts"]
    np.savez_compressed(os.path.join(G, "extra_32x32x100.npz"), **out)


def config5(nx=64, nup=400):
    # BASELINE config 5's frequency grid (README:45: nfreq=25, fmin=2.41e10, fmax=6.31e14) on the FFJET problem, formal and
    # DELO, through the reference's own main.  `cond_*`: the reference's own response, per frequency and Stokes parameter
    # (max |delta| over the image / Stokes peak), to one-ulp changes of spin, mu0 and uout and of fmin/fmax -- the floor
    # below which no other implementation of the same algorithm can be expected to agree.
    from oracle import refmain
    out = {}
    base = dict(FFJET_KW, nn=[nx, nx, nup], nfreq=25, fmin=2.41e10, fmax=6.31e14)
    up = lambda x: float(np.nextafter(x, np.inf))  # noqa: E731
    perts = [dict(spin=up(0.998)), dict(mumin=up(0.906), mumax=up(0.906)), dict(uout=up(0.01)), dict(fmin=up(2.41e10), fmax=up(6.31e14))]
    for iname in ("formal", "delo"):
        r = refmain.run(dict(base, iname=iname), nthreads=8)
        iv = r["ivals"]                                   # [25][npix][4]
        out["ivals_" + iname] = iv
        out["nu"] = r["nu"]
        peak = np.abs(iv.astype(float)).max(axis=1)       # [25][4]
        cond = np.zeros_like(peak)
        for p in perts:
            rp = refmain.run(dict(base, iname=iname, **p), nthreads=8)
            cond = np.maximum(cond, np.abs(rp["ivals"].astype(float) - iv.astype(float)).max(axis=1) / peak)
        out["cond_" + iname] = cond
        print(iname, "cond per frequency (max over Stokes):", cond.max(axis=1))
    np.savez_compressed(os.path.join(G, "ffjet_nfreq25_%dx%dx%d.npz" % (nx, nx, nup)), fmin=2.41e10, fmax=6.31e14, **out)


def thindisk_pol():
    # the reference's `thindisk` regression problem (run_grtrans_test_problems_public.py:111-129; its pickle is not shipped):
    # THINDISK + BBPOL, nvals = 4, standard = 2, nup = 1, three of its 25 frequencies; and the same with extra=1 (7 one-point
    # diagnostics, grtrans_driver.f90:298-311).  Through the reference's own main (needs ch24_vals.txt in the working directory).
    from oracle import refmain
    ch = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "grtrans_amd", "data", "ch24_vals.txt")
    kw = dict(fname='THINDISK', nfreq=3, nmu=1, fmin=2.41e16, fmax=6.31e18, ename='BBPOL', nvals=4, spin=0.9, standard=2, nn=[100, 100, 1],
              uout=0.01, mbh=10, mumin=.26, mumax=.26, gridvals=[-21, 21, -21, 21])
    out = {}
    r = refmain.run(kw, nthreads=8, links=[ch])
    out["ivals"], out["nu"] = r["ivals"], r["nu"]
    r = refmain.run(dict(kw, extra=1), nthreads=8, links=[ch])
    out["ivals_extra"] = r["ivals"]
    assert r["nvals"] == 11 and np.array_equal(r["ivals"][:, :, :4], out["ivals"], equal_nan=True)
    np.savez_compressed(os.path.join(G, "thindisk_bbpol_100x100x1.npz"), **out)


SLOWMOVIE_DUMPS = {k: (20261010 + k, 2000.0 + 10.0 * (k - 38)) for k in (40, 41, 43, 44, 47, 48)}      # 10 M apart


def write_slowmovie_dumps(outdir, nzone=48):
    import make_harm_dump
    return {k: make_harm_dump.write(outdir, n=nzone, name="dump%03d" % k, seed=s, time=t) for k, (s, t) in SLOWMOVIE_DUMPS.items()}


def harm_slowmovie():
    # nt = 3 movie frames of a SLOW-LIGHT run (hnt = 2 slices loaded) whose dumps are 10 M apart, dt = 35: load_harm_data sets
    # tstep = t(1) - t(2) = 10 (fluid_model_harm.f90:493-500), so advance_harm_timestep walks the dump index 41 -> 44 -> 48
    # (with the default 20 it would go 41 -> 42 -> 44).  Through the reference's own grtrans_main.  dt is chosen so that every
    # advance exceeds hnt: for nupdate <= hnt update_harm_data (:523-546) shifts its arrays with sections of unequal length
    # and the compiled reference aborts.
    import hashlib
    import tempfile
    from oracle import refmain
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_slowmovie_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
                  nn=[24, 24, 80], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428, mumax=.6428,
                  gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', hdfile='dump', hhfile='dump041', hindf=41, hnt=2,
                  nt=3, dt=35.)
        r = refmain.run(kw, nthreads=8, links=[paths[k] for k in sorted(paths)])
        out["ivals"] = r["ivals"]
        print(r["log"][-900:])
    np.savez_compressed(os.path.join(G, "harm_slowmovie_24x24x80.npz"), freq=FREQ_HARM, dt=35.0, **out)


KORAL_SLOW_DUMPS = {41: (20261103, 2040.0, 0.12), 40: (20261102, 2020.0, 0.2), 39: (20261101, 2000.0, 0.3),
                    44: (20261104, 2100.0, 0.16), 43: (20261105, 2080.0, 0.26), 42: (20261106, 2060.0, 0.09)}


def harm_slowlight_extra():
    # extra=1 (the 19 ray diagnostics of grtrans_driver.f90:230-292) on a SLOW-LIGHT HARM run: three loaded slices 20 M apart
    # (hnt=3, hindf=41; fluid_model_harm.f90:175-186 picks the slice per point), through the reference's own grtrans_main
    import hashlib
    import tempfile
    from oracle import refmain
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_slow_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
                  nn=[32, 32, 100], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428, mumax=.6428,
                  gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', hdfile='dump', hhfile='dump041', hindf=41, hnt=3,
                  extra=1)
        r = refmain.run(kw, nthreads=8, links=[paths[k] for k in sorted(paths)])
        assert r["nvals"] == 23
        out["vals"] = r["ivals"][0]
        g = np.load(os.path.join(G, "harm_slowlight_32x32x100.npz"))
        # the Stokes columns are the slow-light image of harm_slowlight() (the reference's own driver against ref_driver.f90)
        print("extra slow light: Stokes vs harm_slowlight golden", np.abs(out["vals"][:, :4].astype(float) - g["ivals_formal"].astype(float)).max())
    np.savez_compressed(os.path.join(G, "harm_slowlight_extra_32x32x100.npz"), freq=FREQ_HARM, **out)


def geodebug():
    # debug=1 through the reference's own grtrans_driver (grtrans_driver.f90:94-110, 341-421): geodebug.out of one pixel of the
    # FFJET, SPHACC and HARM problems (the file is re-opened per ray: i1 = i2 = the pixel).  Stored as the file's number stream.
    import tempfile
    import make_harm_dump
    from oracle import refmain
    out = {}
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], i1=500, i2=500, debug=1, iname="formal"), nthreads=1)
    out["ffjet_500"] = r["geodebug"]
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], i1=611, i2=611, debug=1, iname="delo"), nthreads=1)
    out["ffjet_611_delo"] = r["geodebug"]
    r = refmain.run(dict(SPHACC_KW, nn=[32, 32, 100], i1=400, i2=400, debug=1, iname="formal"), nthreads=1)
    out["sphacc_400"] = r["geodebug"]
    with tempfile.TemporaryDirectory() as td:
        path = make_harm_dump.write(td, n=64)
        kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
                  nn=[32, 32, 100], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428, mumax=.6428,
                  gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', hdfile='dump', hhfile='dump040', hindf=40, hnt=1,
                  i1=530, i2=530, debug=1)
        r = refmain.run(kw, nthreads=1, links=[path])
        out["harm_530"] = r["geodebug"]          # a ray that ends behind the horizon: NaN coefficient at its last point -> fac = NaN, not integrated
        r = refmain.run(dict(kw, i1=200, i2=200), nthreads=1, links=[path])
        out["harm_200"] = r["geodebug"]
    for k, v in out.items():
        print(k, len(v), v[:3])
    np.savez_compressed(os.path.join(G, "geodebug_32x32x100.npz"), **out)


def write_koral_slow_dumps(outdir, zones=None):
    import make_koral3d_dump
    zones = zones or dict(n1=32, n2=32, n3=16)
    return {k: make_koral3d_dump.write(outdir, indf=k, seed=s, time=t, b0=b, **zones)[0] + "%04d" % k for k, (s, t, b) in KORAL_SLOW_DUMPS.items()}


def koral3d_slowlight():
    # KORAL3D with several loaded slices (fnt = 3, dumps 41, 40, 39 20 M apart with different field strengths): every point reads
    # the slice its Kerr-Schild look-back time selects (fluid_model_koral3d.f90:126-128, 212-228; time weight zeroed :248).
    # Of the 3-D samplers only KORAL3D's has a defined time step: read_harm3d_data_file never assigns tcur on its BINARY branch
    # (fluid_model_harm3d.f90:383, 402-449; the compiled reference warns "Time step changes over dumps"), load_iharm_data
    # stores the HEADER's time for every slice (fluid_model_iharm.f90:745, 778: tstep = 0).  Through the reference's own
    # grtrans_main: fast light (fnt=1), slow light (fnt=3, formal and delo), nload=2 (per-ray time origin) and two movie frames
    # of a slow-light run (fnt=2, nt=2, dt=65: tstep = 20 from the dumps, advance_koral3d_timestep steps 41 -> 44 and, because
    # nupdate = 3 > fnt, reloads both slices; the shift branch of update_koral3d_data assigns sections of unequal length).
    import hashlib
    import tempfile
    from oracle import refmain
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_koral_slow_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        links = [paths[k] for k in sorted(paths)] + [os.path.join(td, "koral_header")]
        kw = dict(fname='KORAL3D', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
                  nn=[24, 24, 100], uout=0.04, mbh=4e6, nmdot=1, mumin=.5, mumax=.5, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=-1.,
                  iname='formal', hdfile='koral', hhfile='koral_header', hindf=41, hnt=3, ntscl=1.5)
        for key, extra in (("slow", {}), ("fast", dict(hnt=1)), ("nload", dict(nload=2)), ("slow_delo", dict(iname='delo')),
                           ("movie", dict(hnt=2, nt=2, dt=65.))):
            r = refmain.run(dict(kw, **extra), nthreads=8, links=links)
            out["ivals_" + key] = r["ivals"]
            print(key, r["ivals"].shape, float(np.nansum(r["ivals"][..., 0])), r["log"][-300:])
    np.savez_compressed(os.path.join(G, "koral3d_slowlight_24x24x100.npz"), freq=FREQ_HARM, **out)


def koral3d_slowlight_extra():
    # extra=1 on the KORAL3D slow-light run of koral3d_slowlight() (fnt = 3), through the reference's own grtrans_main
    import hashlib
    import tempfile
    from oracle import refmain
    out = {}
    with tempfile.TemporaryDirectory() as td:
        paths = write_koral_slow_dumps(td)
        out["dump_sha256"] = np.array([hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)])
        links = [paths[k] for k in sorted(paths)] + [os.path.join(td, "koral_header")]
        kw = dict(fname='KORAL3D', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
                  nn=[24, 24, 100], uout=0.04, mbh=4e6, nmdot=1, mumin=.5, mumax=.5, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=-1.,
                  iname='formal', hdfile='koral', hhfile='koral_header', hindf=41, hnt=3, ntscl=1.5, extra=1)
        r = refmain.run(kw, nthreads=8, links=links)
        assert r["nvals"] == 23
        out["vals"] = r["ivals"][0]
        g = np.load(os.path.join(G, "koral3d_slowlight_24x24x100.npz"))
        print("koral extra slow: Stokes vs slow golden", np.nanmax(np.abs(out["vals"][:, :4].astype(float) - g["ivals_slow"][0].astype(float))))
    np.savez_compressed(os.path.join(G, "koral3d_slowlight_extra_24x24x100.npz"), freq=FREQ_HARM, **out)


def standard2():
    # standard=2 with nup > 1 (GEOKERR's USEGEOR branch, geokerr_wrapper.f:421-461: mu steps from mu0 to the equatorial plane;
    # ends with u_f < 0 trimmed, geodesics.f90:298-323) on the FFJET problem: images through the reference's own main, trimmed
    # point counts and the raw GEOKERR arrays of a few rays through the per-stage probe.
    from oracle import refmain
    out = {}
    cams = {"a": dict(spin=0.998, mu0=0.906, gridvals=(-40., 20., -20., 40.), nn=(32, 32, 80)),
            "b": dict(spin=0.5, mu0=0.3, gridvals=(-15., 15., -15., 15.), nn=(24, 24, 64))}
    for key, cm in cams.items():
        nn = cm["nn"]
        for iname in ("formal", "delo"):
            r = refmain.run(dict(FFJET_KW, standard=2, spin=cm["spin"], mumin=cm["mu0"], mumax=cm["mu0"], gridvals=list(cm["gridvals"]),
                                 nn=list(nn), iname=iname), nthreads=8)
            out["%s_ivals_%s" % (key, iname)] = r["ivals"][0]
        rb = R.Ref(iname="formal", **dict(FFJET, standard=2, spin=cm["spin"], mu0=cm["mu0"], gridvals=cm["gridvals"], nn=nn))
        pix, npts, _ = rb.trace(1, nn[0] * nn[1], [FREQ], nthreads=8)
        assert np.array_equal(pix[0], out["%s_ivals_formal" % key], equal_nan=True)
        out[key + "_npts"] = npts
        out[key + "_cam"] = np.array([cm["spin"], cm["mu0"], *cm["gridvals"], *nn], dtype=float)
        pick = sorted(set([1, 7, nn[1] * (nn[0] // 2) + nn[1] // 2, nn[1] * (nn[0] // 3) + 2, nn[0] * nn[1]] +
                          [int(i) + 1 for i in np.where(npts < nn[2])[0][:4]]))
        for i in pick:
            g = rb.geokerr_ray(i)
            for k in ("u", "mu", "dt", "dphi", "lam", "tpm", "tpr"):
                out["%s_ray%d_%s" % (key, i, k)] = g[k]
        out[key + "_rays"] = np.array(pick)
    # an image on the synthetic HARM dump (the far points of a standard=2 ray start at r = 1/u0 >= 1e4, outside every table:
    # FFJET reads past its arrays there in the reference, the HARM sampler has a defined answer)
    import tempfile
    import make_harm_dump
    with tempfile.TemporaryDirectory() as td:
        path = make_harm_dump.write(td, n=64)
        hk = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=2,
                  nn=[24, 24, 64], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, mumin=.6428, mumax=.6428, gridvals=[-13, 13, -13, 13],
                  muval=0.25, gmin=1., hdfile='dump', hhfile='dump040', hindf=40, hnt=1)
        for iname in ("formal", "delo"):
            out["harm_ivals_" + iname] = refmain.run(dict(hk, iname=iname), nthreads=8, links=[path])["ivals"][0]
        rb = R.Ref(iname="formal", datadir=td, **dict(HARM, standard=2, nn=(24, 24, 64)))
        pix, npts, _ = rb.trace(1, 576, [FREQ_HARM], nthreads=8)
        assert np.array_equal(pix[0], out["harm_ivals_formal"], equal_nan=True)
        out["harm_npts"] = npts
    np.savez_compressed(os.path.join(G, "standard2_ffjet.npz"), freq=FREQ, **out)


def polar_camera():
    # nrotype = 1: the polar pixel grid (geokerr_wrapper.f:138-149, 263-272): ro_i = r1 (rcut/r1)**(i/nro), r1 = gridvals(1),
    # phi_j = 2 pi (j - 1/2)/nphi; FFJET problem, through the reference's own main
    from oracle import refmain
    out = {}
    kw = dict(FFJET_KW, nn=[24, 16, 100], gridvals=[2., 0., 0., 0.], rcut=30., nrotype=1)
    for iname in ("formal", "delo"):
        r = refmain.run(dict(kw, iname=iname), nthreads=8)
        out["ivals_" + iname] = r["ivals"][0]
        out["ab"] = r["ab"]
    np.savez_compressed(os.path.join(G, "ffjet_polar_24x16x100.npz"), freq=FREQ, **out)


def crosscheck_main():
    # the image goldens above were produced through oracle/ref_driver.f90 (per-stage probes + a restated pixel loop);
    # assert that the reference's own main gives the same bits, and record it
    from oracle import refmain
    import json
    rec = {}
    g = np.load(os.path.join(G, "ffjet_100x100x400.npz"))
    for iname in ("formal", "delo"):
        r = refmain.run(dict(FFJET_KW, iname=iname), nthreads=8)
        rec["ffjet_100x100x400_" + iname] = bool(np.array_equal(r["ivals"][0], g["ivals_" + iname]))
    g = np.load(os.path.join(G, "sphacc_64x64x200.npz"))
    for iname in ("formal", "delo"):
        r = refmain.run(dict(SPHACC_KW, iname=iname), nthreads=1)
        rec["sphacc_64x64x200_" + iname] = bool(np.array_equal(r["ivals"][0], g["ivals_" + iname]))
    g = np.load(os.path.join(G, "nvals1_32x32x100.npz"))
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], nvals=1), nthreads=8)
    rec["nvals1_ffjet"] = bool(np.array_equal(r["ivals"][0], g["ffjet_ivals"][0]))
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], nvals=1, ename='SYNCHPL'), nthreads=8)
    rec["nvals1_ffjet_synchpl"] = bool(np.array_equal(r["ivals"][0], g["ffjet_synchpl_ivals"][0]))
    g = np.load(os.path.join(G, "thindisk_100x100x1.npz"))
    r = refmain.run(dict(standard=2, spin=0.998, uout=1e-4, gridvals=[-25., 25., -25., 25.], nn=[100, 100, 1], fname='THINDISK', ename='BB',
                         nvals=1, iname='lsoda', mbh=10., tmdot=0.1, nmu=2, mumin=0.1, mumax=0.7, nfreq=1, fmin=1e17, fmax=1e17), nthreads=8)
    rec["thindisk_mu01"] = bool(np.array_equal(r["ivals"][0], g["ivals_mu01"]))
    rec["thindisk_mu07"] = bool(np.array_equal(r["ivals"][1], g["ivals_mu07"]))
    print(rec)
    json.dump(rec, open(os.path.join(G, "refmain_crosscheck.json"), "w"), indent=1)
    assert all(rec.values()), rec


# ---- the open parity cases of the r03 fuzz sweeps, pinned against the COMPILED reference ------------------------
# Each case = the exact camera of a sweep log (tools/fuzz_cameras.py in GRT_FUZZ_DRY mode prints it), the compiled reference's image,
# and the compiled reference's OWN response to perturbations no user can specify: every frequency moved by one ulp either way and by
# 1e-13, the camera by one ulp in inclination, spin, uout ("cond": largest change per frequency and Stokes parameter, as a fraction of
# that Stokes peak), and every frequency moved by 1e-8 ("amp": the change divided by 1e-8).  The GPU tests state their bars from these.
OPEN_CASES = {
    # profiles/r03_h_fuzz_700.log (seed 31337) camera 524: rays that start in near-vacuum zones of the KORAL3D dump
    "koral3d_delo_69ghz": dict(fname="KORAL3D", ename="POLSYNCHTH", iname="delo", spin=0.9375, mu0=0.0, uout=0.04, nn=(12, 12, 60),
                               gridvals=(-9.572477643577713, 5.75296751663922, -11.596725842453694, 3.7287193177632396),
                               dfile="koral", hfile="koral_header", findf=40, mbh=4e6, muval=0.25, gmin=-1., ftscl=1.5, p1=5.5,
                               freqs=[2.3e11, 6.9e10], i1=43, i2=132, dump="koral3d"),
    # profiles/r03_g_fuzz_700.log (seed 777) camera 240: equatorial Schwarzschild camera, lambda carries a constant of 9e8
    "sphacc_formal_equatorial": dict(fname="SPHACC", ename="POLSYNCHTH", iname="formal", spin=0.0, mu0=0.0, uout=0.04, nn=(12, 12, 60),
                                     gridvals=(-36.402300212602, 48.39734536577427, -41.53000646055861, 43.26963911781766),
                                     mbh=10., gmin=100., p1=3.0, freqs=[7.23e11, 2.41e11, 2.41e10], i1=71, i2=110, dump=None),
    # profiles/r03_f_fuzz_700.log (seed 5151) camera 303: DELO at optical depths of tens per step, alternating-sign growth
    "sphacc_delo_24ghz": dict(fname="SPHACC", ename="POLSYNCHTH", iname="delo", spin=0.0, mu0=0.0, uout=0.01, nn=(12, 12, 100),
                              gridvals=(-6.46579504122852, 4.212597494570896, -9.68891574108579, 0.9894767947136254),
                              mbh=10., gmin=100., p1=3.5, freqs=[7.23e11, 2.41e11, 2.41e10], i1=1, i2=144, dump=None),
}


# r05: the two FFJET cameras the per-pixel rules of tools/fuzz_cameras.py leave above the line: POLSYNCHPL with p = 5.5 at a tenth
# of the nominal frequency, Faraday-thick rays whose Q, U, V oscillate from step to step
OPEN_CASES_R05 = {
    # profiles/r05e_fuzz_200_seed8008_per_pixel_rules.log camera 64
    "ffjet_p55_34ghz_seed8008": dict(fname="FFJET", ename="POLSYNCHPL", iname="formal", spin=0.05213690551730877, mu0=0.0, uout=0.01, nn=(12, 12, 60),
                                     gridvals=(-9.570136125286059, 10.569561098919507, -4.811191063496265, 15.328506160709301), p1=5.5,
                                     freqs=[1.035e12, 3.45e11, 3.45e10], i1=26, i2=134, dump=None),
    # profiles/r05o_fuzz_700_seed424242.log camera 134
    "ffjet_p55_34ghz_seed424242": dict(fname="FFJET", ename="POLSYNCHPL", iname="formal", spin=0.567049935230656, mu0=-0.42378143169961224, uout=0.02,
                                       nn=(12, 12, 100), gridvals=(-10.498569236028937, 13.589957295400408, -10.355504459802308, 13.733022071627037),
                                       p1=5.5, freqs=[1.035e12, 3.45e11, 3.45e10], i1=1, i2=144, dump=None),
}


def open_cases(cases=None, fn="open_cases.npz"):
    import tempfile
    import make_koral3d_dump
    cases = OPEN_CASES if cases is None else cases
    out = {}
    with tempfile.TemporaryDirectory() as td:
        if any(c["dump"] for c in cases.values()):
            make_koral3d_dump.write(td, n1=32, n2=32, n3=16)          # the sweep's dump (tools/fuzz_cameras.py)
        for name, case in cases.items():
            kw = {k: v for k, v in case.items() if k not in ("freqs", "i1", "i2", "dump")}
            if case["dump"]:
                kw["datadir"] = td
            freqs, i1, i2 = case["freqs"], case["i1"], case["i2"]

            def run(fr=freqs, **chg):
                r = R.Ref(**dict(kw, **chg))
                pix, npts, _ = r.trace(i1, i2, list(fr), nthreads=1)
                return np.asarray(pix, dtype=np.float64), np.asarray(npts)
            ref, npts = run()
            nf = len(freqs)
            pk = np.zeros((nf, 4))
            for k in range(nf):
                okm = np.isfinite(ref[k]).all(axis=1)
                pk[k] = np.abs(ref[k][okm]).max(axis=0)
                pk[k] = np.maximum(pk[k], 1e-2 * pk[k][0])      # a Stokes parameter that vanishes by symmetry (as the sweep prices it)

            def resp(other, onp):
                d = np.zeros((nf, 4))
                if not np.array_equal(onp, npts):
                    return d
                for k in range(nf):
                    okm = np.isfinite(ref[k]).all(axis=1) & np.isfinite(other[k]).all(axis=1)
                    if okm.any():
                        d[k] = np.abs(other[k][okm] - ref[k][okm]).max(axis=0) / pk[k]
                return d
            cond = np.zeros((nf, 4))
            for fr in ([float(np.nextafter(f, np.inf)) for f in freqs], [float(np.nextafter(f, 0.0)) for f in freqs],
                       [f * (1 + 1e-13) for f in freqs], [f * (1 - 1e-13) for f in freqs]):
                cond = np.maximum(cond, resp(*run(fr=fr)))
            # (an equatorial camera has mu0 = 0, where "one ulp" is a denormal: the inclination moves by 1e-15 rad as well)
            gvp = tuple(float(np.nextafter(v, np.inf)) for v in kw["gridvals"])
            for chg in (dict(mu0=float(np.nextafter(kw["mu0"], 2.0))), dict(mu0=float(np.nextafter(kw["mu0"], -2.0))),
                        dict(mu0=kw["mu0"] + 1e-15), dict(mu0=kw["mu0"] - 1e-15), dict(gridvals=gvp),
                        dict(uout=float(np.nextafter(kw["uout"], 1.0))), dict(uout=float(np.nextafter(kw["uout"], -1.0)))) + \
                       ((dict(spin=float(np.nextafter(kw["spin"], 2.0))),) if kw["spin"] != 0.0 else ()):
                cond = np.maximum(cond, resp(*run(**chg)))
            amp = np.zeros((nf, 4))
            for rel in (1e-8, -1e-8):
                amp = np.maximum(amp, resp(*run(fr=[f * (1 + rel) for f in freqs])) / 1e-8)
            out[name + "_ivals"] = ref.astype(np.float32); out[name + "_npts"] = npts
            out[name + "_peak"] = pk; out[name + "_cond"] = cond; out[name + "_amp"] = amp
            print(name, "cond", cond.max(axis=1), "amp", amp.max(axis=1))
    np.savez_compressed(os.path.join(G, fn), **out)


def delo_vacuum_pocket_ray():
    """INPUT fixture (no expected values): the coefficient dump of one ray -- j, K as grtrans_driver hands them to integrate, lambda, tau --
    from the oracle, for tests/test_hostsim_fast.py::test_delo_through_a_vacuum_pocket... (fuzz seed 1234, camera 300, pixel 42, 23 GHz)."""
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
    import make_harm3d_dump
    from oracle import orabind
    td = tempfile.mkdtemp(prefix="grt_gold_")
    make_harm3d_dump.write(td, n1=32, n2=32, n3=16)
    gv = (-7.154324163139105, 8.484635693640907, -8.200640002803938, 7.438319853976074)
    o = orabind.Oracle(nn=(12, 12, 60), iname="delo", spin=0.9375, mu0=0.0, gridvals=gv, uout=0.04, fname="HARM3D", ename="HYBRIDTHPL", datadir=td,
                       dfile="dump3d", hfile="dump3d_header", findf=40, mbh=4e6, mdot=1e20, muval=0.25, gmin=2., p1=3.5)
    od = o.ray_dump(42, 2.3e10)
    n = od["npts"]
    np.savez_compressed(os.path.join(G, "delo_vacuum_pocket_ray.npz"), j=np.asarray(od["j"])[:n], K=np.asarray(od["K"])[:n],
                        lam=np.asarray(od["lam"])[:n], tau=np.asarray(od["tau"])[:n],
                        note=np.array("oracle ray_dump of pixel 42 at 23 GHz: HARM3D HYBRIDTHPL DELO, tools/fuzz_cameras.py 700 1234 allext camera 300"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "delo_vacuum_pocket_ray":
        delo_vacuum_pocket_ray()
        sys.exit(0)
    if len(sys.argv) > 1:            # e.g. `python tests/make_golden.py config5 extra_main crosscheck_main`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    leafs(); coeffs(); integrators(); reference_pickle()
    ffjet(100, 400)
    ffjet(32, 100)
    sphacc(64, 200)
    harm(48, 200)
    harm_config4_zones()
    nvals1()
    extra()          # ref_driver.f90's restated diagnostics (kept as the npts source) ...
    extra_main()     # ... overwritten by the reference's own grtrans_driver
    thindisk()
    harm_movie()
    harm_slowlight()
    harm_slowlight_extra()
    geodebug()
    harm3d()
    hybrid()
    maxjutt()
    iharm()
    koral3d()
    powerlaw_grmhd()
    unpolarized()
    sympol()
    config5()
    thindisk_pol()
    harm_slowmovie()
    koral3d_slowlight_extra()
    standard2()
    polar_camera()
    open_cases()
    open_cases(OPEN_CASES_R05, "open_cases_r05.npz")
    delo_vacuum_pocket_ray()
    crosscheck_main()
    print("golden fixtures written to", os.path.abspath(G))
