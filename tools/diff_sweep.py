"""tools/diff_sweep.py OUT.npz [ncams] [seed]: random small cameras through EVERY (fluid model, emissivity, integrator, frequency-batch,
extra=1, nvals=1) kernel family of the library named by GRTRANS_AMD_LIB (default: the product), results written to OUT.npz -- stored-point
counts, images, the per-step record of a few rays and GEOKERR's raw arrays (tpmi, tpri, ...) of the camera's rays.  No oracle, no
reference: the DIFFERENTIAL-COMPILATION check (tests/test_gpu_parity.py::test_differential_compilation...) runs this with the product
library and with build/diff/libgrtrans_amd_diff.so (tools/build_diff.sh: the same sources, every kernel compiled for another occupancy
= another register allocation) and compares the two files: integers bit-equal, Stokes to 1e-12 of each peak.  A live value overwritten
by a register copy or spill (defects 3 and 5) changes with the allocation; arithmetic does not."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grtrans_amd import capi
import make_harm_dump, make_harm3d_dump, make_iharm_dump, make_koral3d_dump

out = sys.argv[1]
ncams = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 20261003)
td = tempfile.mkdtemp(prefix="grt_diff_")
make_harm_dump.write(td, n=64)
d3, h3 = make_harm3d_dump.write(td, n1=32, n2=32, n3=16)
di, hi = make_iharm_dump.write(td, n1=32, n2=32, n3=16, metric=1)
dk, hk = make_koral3d_dump.write(td, n1=32, n2=32, n3=16)
dev = capi.Device(0); dev.load_tables(); dev.load_ffjet(); dev.load_sphacc()
dev.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
dev.load_harm3d(d3, h3, 40, 0.9375); dev.load_iharm(di, hi, 40, 0.9375); dev.load_koral3d(dk, hk, 40, 0.9375)
capi.geo_reuse(False)
K3 = dict(mbh=4e6, nfac=1.5, gmin=-1.); H2 = dict(mbh=4e6, gmin=1., mdot=1.57e15); H3 = dict(mbh=4e6, gmin=2., mdot=1e20); IH = dict(mbh=4e6, gmin=3., mdot=1e20)
SP = dict(mbh=10., gmin=100.); FJ = dict(mbh=3.4e9, nfac=2.0, bfac=70.0, gmin=100.)
# every (fluid model, emissivity) pair that has a kernel (grt_kernels.hip:grt_launch_transfer); scalar = nvals 1 only
PAIRS = [("KORAL3D", "POLSYNCHTH", K3), ("HARM", "POLSYNCHTH", H2), ("HARM3D", "POLSYNCHTH", H3), ("IHARM", "POLSYNCHTH", IH),
         ("HARM", "HYBRIDTHPL", H2), ("HARM3D", "HYBRIDTHPL", H3), ("IHARM", "HYBRIDTHPL", IH), ("HARM", "HYBRIDTH", H2), ("IHARM", "HYBRIDPL", IH),
         ("HARM", "MAXJUTT", H2), ("HARM3D", "MAXJUTT", H3), ("IHARM", "MAXJUTT", IH), ("HARM3D", "MAXCOMP", H3),
         ("HARM", "POLSYNCHSYMTH", H2), ("HARM3D", "POLSYNCHSYMTH", H3), ("IHARM", "POLSYNCHSYMTH", IH), ("SPHACC", "POLSYNCHSYMTH", SP),
         ("SPHACC", "POLSYNCHTH", SP), ("FFJET", "POLSYNCHPL", FJ)]
SCALAR = [("HARM", "SYNCHTHAV", H2), ("HARM3D", "BREMS", H3), ("IHARM", "SYNCHTHBREMS", IH), ("SPHACC", "SYNCHTHAVNOABS", SP), ("KORAL3D", "RHO", K3),
          ("FFJET", "SYNCHPL", FJ)]
grid = [x for x in (1.0, 0.3, 0.45, 0.9, 1.3, 1.8, 2.2, 2.6, 3.0)]
res = {}
log = []
for c in range(ncams):
    scalar = rng.random() < 0.12
    fname, ename, emk = (SCALAR if scalar else PAIRS)[int(rng.integers(0, len(SCALAR if scalar else PAIRS)))]
    a = 0.9375
    if fname == "FFJET": a = float(rng.choice([rng.uniform(0.01, 0.998), -rng.uniform(0.05, 0.99), 0.998, 0.5]))
    if fname == "SPHACC": a = float(rng.choice([0.0, 0.0, rng.uniform(0.01, 0.99)]))
    mu0 = float(rng.choice([rng.uniform(-0.99, 0.99), 0.0, 0.999, 1.0]))
    half = float(rng.uniform(4., 45.)) if fname in ("FFJET", "SPHACC") else float(rng.uniform(4., 16.))
    cx, cy = float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))
    gv = (cx - half, cx + half, cy - half, cy + half)
    uout = float(rng.choice([0.01, 0.02, 0.04])) if fname in ("FFJET", "SPHACC") else 0.04
    nn = (12, 12, int(rng.choice([40, 60, 100, 160])))
    iname = str(rng.choice(["formal", "delo"]))
    nvals = 1 if (scalar or rng.random() < 0.12) else 4
    nf = int(rng.choice([1, 1, 2, 9, 13]))
    extra = (nvals == 4 and nf == 1 and ename != "POLSYNCHSYMTH" and rng.random() < 0.3)
    rec = bool(rng.integers(0, 2)) and not extra
    f0 = {"FFJET": 3.45e11, "SPHACC": 2.41e11}.get(fname, 2.3e11) * float(rng.choice([1.0, 1.0, 0.3, 3.0]))
    freqs = [f0 * g for g in grid[:nf]]
    p1 = float(rng.choice([3.5, 3.0, 5.5]))
    cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
    em = capi.make_emis(ename, emk["mbh"], emk.get("nfac", 1.0), emk.get("bfac", 6.0), emk["gmin"], 1e5, p1, muval=0.25, mdot=emk.get("mdot", 0.0),
                        otherargs=(1.5, 1.0, 0.5) if ename == "MAXJUTT" else ((1.5, 1.0, 1.0, 0.5) if ename == "MAXCOMP" else (1.0, 1.0)))
    capi.record_steps(rec)
    npx = nn[0] * nn[1]
    pix, _, status, _ = capi.trace_tile(cam, 1, npx, fname, em, freqs, iname, nvals=nvals, nextra=19 if extra else 0)
    res["c%d_status" % c] = status; res["c%d_pix" % c] = pix
    if rec:
        res["c%d_rec" % c] = capi.download_steps(0, npx, cam.nup + cam.next)
    if c % 4 == 0:      # GEOKERR's own arrays on this camera's pixel constants (class_geokerr surface: the set-up + geodesic kernels in raw mode)
        cp = capi.camera_pixels(cam, 1, npx)
        g = capi.geokerr_rays(cam.u0, cam.uout, cam.mu0, cam.a, cam.offset, cam.nup, cp["uf"], cp["l"], cp["q2"], cp["alpha"], cp["beta"], cp["tpr"], cp["su"],
                              cp["sm"], mufill=cam.mufill, kext=cam.kext, next=cam.next)
        for k, v in g.items():
            res["c%d_gk_%s" % (c, k)] = np.asarray(v)
    log.append("cam %d %s %s %s nvals=%d nf=%d extra=%d rec=%d a=%.4f mu0=%.4f nup=%d" % (c, fname, ename, iname, nvals, nf, extra, rec, a, mu0, nn[2]))
capi.record_steps(False)
dev.close()
res["log"] = np.array(log)
np.savez_compressed(out, **res)
print("wrote %s: %d cameras, library %s" % (out, ncams, os.environ.get("GRTRANS_AMD_LIB", "product")))
