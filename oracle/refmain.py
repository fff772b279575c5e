"""oracle/refmain.py -- TEST INFRASTRUCTURE ONLY.

Runs the UNMODIFIED reference end to end: its own `grtrans_run(ifile, outfile)` (grtrans.f90:1-52 ->
read_inputs -> pgrtrans.grtrans_main -> grtrans_driver per pixel -> kwrite_raytrace_camera, cflag=0),
compiled in place from /root/reference into oracle/_ref/libgrtrans_ref.so by oracle/build_ref.sh.
Nothing of the reference's sequencing is restated on this route: the namelist goes in, the binary
camera file (camera.f90:321-340) comes out.

Each run happens in a child process (`python -m oracle.refmain job.json`): the reference keeps module
state between calls (e.g. `integer :: nextra=0` in grtrans_main is implicitly SAVEd, pgrtrans.f90:60)
and STOPs on some errors (math.f90:75-79).

Used by tests/make_golden.py (fixture generation, build container) and tests/test_ref_main.py.
"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_ref", "libgrtrans_ref.so")
DATA = os.path.join(ROOT, "grtrans_amd", "data")
REFERENCE = os.environ.get("GRT_REFERENCE", "/root/reference")


def available():
    return os.path.exists(LIB)


def _write_inputs(path, kw, writer):
    """inputs.in for kwargs `kw` (grtrans_batch.py kwargs).  writer='reference' imports the reference's own
    grtrans_batch.py (build container only; astropy and the f2py module are not needed for writing and are
    stubbed in sys.modules), writer='mirror' uses grtrans_amd.grtrans_batch (pinned against namelists the
    reference's writer produced: tests/golden/inputs_*.in)."""
    if writer == "reference":
        import types
        for name in ("astropy", "astropy.io", "astropy.io.fits", "pgrtrans"):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.modules["astropy.io"].fits = sys.modules["astropy.io.fits"]
        sys.modules["astropy"].io = sys.modules["astropy.io"]
        sys.modules["pgrtrans"].pgrtrans = None
        sys.path.insert(0, REFERENCE)
        cwd = os.getcwd()
        os.chdir(os.path.dirname(path))
        try:
            import grtrans_batch as gb          # the reference's
            x = gb.grtrans()
            x.write_grtrans_inputs(os.path.basename(path), oname="grtrans.out", **kw)
        finally:
            os.chdir(cwd)
            sys.path.remove(REFERENCE)
    else:
        sys.path.insert(0, ROOT)
        from grtrans_amd import grtrans_batch as gb
        gb.grtrans_inputs(**kw).write(path)


def read_camera_file(path):
    """cflag=0 file (camera.f90:321-340): per camera the sequential records nx,ny,nv / nkey / nkey x real(4) /
    pixloc(2,npix) / pixvals(nv,npix)."""
    ivals, keys_all = [], []
    ab = None
    with open(path, "rb") as f:
        def rec(dtype):
            n = np.fromfile(f, dtype="i4", count=1)
            if n.size == 0:
                return None
            a = np.fromfile(f, dtype=dtype, count=int(n[0]) // np.dtype(dtype).itemsize)
            np.fromfile(f, dtype="i4", count=1)
            return a
        while True:
            h = rec("i4")
            if h is None:
                break
            nx, ny, nv = [int(v) for v in h]
            nkey = int(rec("i4")[0])
            keys_all.append([float(rec("f4")[0]) for _ in range(nkey)])
            ab = rec("f4").reshape(nx * ny, 2)
            ivals.append(rec("f4").reshape(nx * ny, nv))
    return dict(ivals=np.stack(ivals, 0), ab=ab, nu=np.array([k[2] for k in keys_all]), nx=nx, ny=ny, nvals=nv)


def _child(job):
    kw = job["kw"]
    wd = job["workdir"]
    os.chdir(wd)
    kw = dict(kw)
    kw["cflag"] = 0
    _write_inputs(os.path.join(wd, "inputs.in"), kw, job["writer"])
    lib = C.CDLL(LIB, mode=os.RTLD_LAZY)      # the cfitsio entry points stay undefined; cflag=0 never calls them
    # grtrans_run(ifile, outfile): character(len=300) dummies (grtrans.f90:18) -> blank-padded 300-byte buffers;
    # flang passes the two lengths by value after the explicit arguments
    ifile = b"inputs.in".ljust(300)
    ofile = b"grtrans.out".ljust(300)
    sys.stdout.flush()
    lib.grtrans_run_(C.c_char_p(ifile), C.c_char_p(ofile), C.c_size_t(300), C.c_size_t(300))


def run(kw, nthreads=None, writer=None, links=(), keep=False, timeout=3600):
    """Run the reference's grtrans_run on kwargs `kw` (those of grtrans_batch.write_grtrans_inputs).
    links: extra files to symlink into the scratch working directory (HARM dumps etc.); the FFJET table is always there.
    Returns dict(ivals[ncams, npix, nvals+nextra] REAL(4), ab[npix,2], nu[ncams], nx, ny, nvals)."""
    if writer is None:
        writer = "reference" if os.path.isdir(REFERENCE) else "mirror"
    wd = tempfile.mkdtemp(prefix="grt_refmain_")
    for p in list(links) + glob.glob(os.path.join(DATA, "*.bin")):
        dst = os.path.join(wd, os.path.basename(p))
        if not os.path.exists(dst):
            os.symlink(os.path.abspath(p), dst)
    job = os.path.join(wd, "job.json")
    json.dump(dict(kw=kw, workdir=wd, writer=writer), open(job, "w"))
    env = dict(os.environ)
    if nthreads:
        env["OMP_NUM_THREADS"] = str(nthreads)
    env["OMP_STACKSIZE"] = env.get("OMP_STACKSIZE", "256M")
    with open(os.path.join(wd, "stdout.log"), "w") as lo:
        rc = subprocess.call([sys.executable, "-m", "oracle.refmain", job], cwd=ROOT, env=env, stdout=lo, stderr=subprocess.STDOUT,
                             timeout=timeout)
    # the flang build keeps the trailing blanks of character(len=300) in the file name
    outs = [p for p in glob.glob(os.path.join(wd, "grtrans.out*"))]
    if rc != 0 or not outs:
        raise RuntimeError("reference run failed (rc=%d), log: %s" % (rc, open(os.path.join(wd, "stdout.log")).read()[-2000:]))
    res = read_camera_file(outs[0])
    res["log"] = open(os.path.join(wd, "stdout.log")).read()
    gd = os.path.join(wd, "geodebug.out")
    if os.path.exists(gd):      # debug=1: the reference's own dump of the last ray (list-directed, wrapped lines): the number stream
        import numpy as _np
        res["geodebug"] = _np.array(open(gd).read().split(), dtype=float)
    if not keep:
        import shutil
        shutil.rmtree(wd, ignore_errors=True)
    return res


if __name__ == "__main__":
    _child(json.load(open(sys.argv[1])))
