"""Compare two library builds on the same cameras (one-off validation of a new fast path against the generic one):
python tools/compare_builds.py LIB_A LIB_B  -> per camera: stored-point counts equal, max Stokes difference / peak."""
import os, sys, subprocess, json, tempfile
import numpy as np
CAMS = [(0.998, 0.906, (-40., 20., -20., 40.), 0.01), (0.9375, 0.6428, (-13., 13., -13., 13.), 0.04), (0.5, -0.3, (-20., 20., -20., 20.), 0.02),
        (0.998, 0.1, (-15., 15., -15., 15.), 0.04), (0.2, 0.95, (-12., 12., -12., 12.), 0.04)]
if len(sys.argv) == 4:      # child: trace with one library, save
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from grtrans_amd import capi
    dev = capi.Device(0); dev.load_ffjet(); dev.load_tables()
    out = {}
    for k, (a, mu0, gv, uout) in enumerate(CAMS):
        for rec in (0, 1):
            capi.record_steps(bool(rec))
            cam = capi.make_camera(1, gv, (384, 384, 200), a, mu0, uout, -0.5)
            em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
            pix, _, status, st = capi.trace_tile(cam, 1, 384 * 384, "FFJET", em, [3.45e11], "formal")
            out["pix%d_%d" % (k, rec)] = pix[0]; out["st%d_%d" % (k, rec)] = status
    np.savez(sys.argv[3], **out)
    dev.close()
else:
    td = tempfile.mkdtemp()
    files = []
    for i, lib in enumerate(sys.argv[1:3]):
        f = os.path.join(td, "b%d.npz" % i); files.append(f)
        subprocess.check_call([sys.executable, __file__, "child", lib, f], env=dict(os.environ, GRTRANS_AMD_LIB=os.path.abspath(lib)))
    A, B = np.load(files[0]), np.load(files[1])
    for k in range(len(CAMS)):
        for rec in (0, 1):
            a, b = A["pix%d_%d" % (k, rec)], B["pix%d_%d" % (k, rec)]
            pk = np.abs(a).max(axis=0)
            print("camera", k, "record", rec, "npts equal", np.array_equal(A["st%d_%d" % (k, rec)], B["st%d_%d" % (k, rec)]),
                  "nan", int(np.isnan(a).sum()), int(np.isnan(b).sum()), "max diff / peak", (np.abs(a - b).max(axis=0) / pk).max())
