"""Per-step internals of one ray on the device against the oracle's ray_dump.  Needs the experiment builds
build/variants/lib_dbg{0..6}.so (tools/build_variant.sh dbgN "-DGRT_DEBUG_SEL=N"), one process per build:
    python tools/debug_steps.py SEL a mu0 g1 g2 g3 g4 uout n1 n2 nup iname pixel step
writes gpurun_out/dbg_SEL.npy; with SEL = "cmp" the seven files are compared with the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sel = sys.argv[1]
a, mu0 = float(sys.argv[2]), float(sys.argv[3]); gv = tuple(float(v) for v in sys.argv[4:8]); uout = float(sys.argv[8])
nn = tuple(int(v) for v in sys.argv[9:12]); iname = sys.argv[12]; pixel = int(sys.argv[13]); step = int(sys.argv[14])
freq = 3.45e11
os.makedirs("gpurun_out", exist_ok=True)
if sel != "cmp":
    os.environ["GRTRANS_AMD_LIB"] = os.path.abspath("build/variants/lib_dbg%s.so" % sel)
    from grtrans_amd import capi
    dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
    capi.record_steps(True)
    cam = capi.make_camera(1, gv, nn, a, mu0, uout, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5, muval=0.25, mdot=0.0)
    pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [freq], iname)
    n = int(status[pixel - 1])
    rec = capi.download_steps(pixel - 1, 1, cam.nup + cam.next)[:, 0, :n]
    np.save("gpurun_out/dbg_%s.npy" % sel, rec)
    dev.close()
    sys.exit(0)
from oracle import orabind
o = orabind.Oracle(nn=nn, iname=iname, spin=a, mu0=mu0, gridvals=gv, uout=uout, fname="FFJET", ename="POLSYNCHPL")
d = o.ray_dump(pixel, freq); n = d["npts"]
R = [np.load("gpurun_out/dbg_%d.npy" % s) for s in range(7)]
names = [("rshift", "ang", "s2xi", "c2xi"), ("ncgsnth", "bcgs", ("jraw", 0), ("Kraw", 0)), (("jraw", 1), ("jraw", 3), ("Kraw", 1), ("Kraw", 3)),
         (("Kraw", 4), ("Kraw", 6), ("u", 0), ("b", 0)), (("k", 0), ("k", 1), ("k", 2), ("k", 3)), (("b", 1), ("b", 2), ("b", 3), "bmag"),
         (("u", 1), ("u", 2), ("u", 3), "rho")]
for s in range(7):
    for c in range(4):
        nm = names[s][c]
        ref = np.asarray(d[nm] if isinstance(nm, str) else d[nm[0]][:, nm[1]], dtype=float)[:n]
        got = R[s][4 + c]
        rel = np.abs(got - ref) / (np.abs(ref).max() + 1e-300)
        k = int(rel.argmax())
        print("%-12s max rel diff %.3e at step %d | at step %d: dev %.17g ref %.17g" % (str(nm), rel.max(), k, step, got[step], ref[step]))
