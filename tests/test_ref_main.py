"""The compiled, unmodified reference run END TO END through its own grtrans_run / grtrans_main / grtrans_driver
(oracle/refmain.py) against (a) the vectors the reference itself ships for this path and (b) the committed goldens
that were produced through oracle/ref_driver.f90's per-stage route.  Skips when oracle/_ref is not built.

(a) pins the oracle: run_grtrans_test_problems_public.py:60-75 (FFJET, LSODA, pass mark 1e-2) and :27-52 (SPHACC
intensity profile and spectrum, pass mark 10*tol = 0.1) with the reference's own error measure.
(b) shows that the restated pixel loop of ref_driver.f90 -- kept as the per-stage probe -- gives the bits of the
reference's own driver, extra=1 diagnostics (grtrans_driver.f90:230-292) included.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import refmain  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.skipif(not refmain.available(), reason="oracle/_ref not built (needs /root/reference)")

FFJET_KW = dict(fname='FFJET', nfreq=1, nmu=1, fmin=3.45e11, fmax=3.45e11, ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, uout=0.01,
                mbh=3.4e9, mumin=.906, mumax=.906, nn=[100, 100, 400], gridvals=[-40., 20., -20., 40.], ntscl=2., nrscl=70.)
SPHACC_KW = dict(fname='SPHACC', nfreq=1, nmu=1, fmin=2.41e11, fmax=2.41e11, ename='POLSYNCHTH', nvals=4, spin=0., standard=1,
                 uout=0.04, mbh=10., nn=[32, 32, 100], gridvals=[-25., 25., -25., 25.], mumin=.5, mumax=.5, iname='formal')


def test_reference_lsoda_reproduces_its_own_ffjet_pickle():
    # run_grtrans_test_problems_public.py:60-75: iname defaults to 'lsoda'; terr = sum|x-g|/sum|g| < 1e-2
    r = refmain.run(dict(FFJET_KW, iname='lsoda'), nthreads=8, writer="mirror")
    g = np.load(os.path.join(G, "test_grtrans_ffjet_lsoda.npz"))["ivals"].astype(float)      # [npix][4]
    x = r["ivals"][0].astype(float)
    terr = np.abs(x - g).sum() / np.abs(g).sum()
    print("LSODA vs test_grtrans_ffjet.p: terr = %.5f (pass mark 1e-2)" % terr)
    assert terr < 1e-2
    assert abs(terr - 0.0054) < 5e-4          # the figure SURVEY 8c measured for the full unmodified build


def test_reference_reproduces_its_own_sphacc_pickles():
    # run_grtrans_test_problems_public.py:27-52: 25 frequencies 1e8..1e15, 10000 x 1 radial strip, SYNCHTHAV, nvals=1, LSODA
    kw = dict(fname='SPHACC', nfreq=25, nmu=1, fmin=1e8, fmax=1e15, ename='SYNCHTHAV', nvals=1, spin=0., mbh=1., standard=1,
              nn=[10000, 1, 100], gridvals=[0., 400., 0., 0.], uout=.0025)
    # one thread per run: the reference's SPHACC path is not thread-safe; eight i1..i2 strips side by side instead
    from concurrent.futures import ThreadPoolExecutor
    strips = [(1 + 1250 * k, 1250 * (k + 1)) for k in range(8)]
    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(lambda s: refmain.run(dict(kw, i1=s[0], i2=s[1]), nthreads=1, writer="mirror"), strips))
    r = parts[0]
    g = np.load(os.path.join(G, "test_grtrans_sphacc_lsoda.npz"))
    iv = np.zeros(r["ivals"].shape)                        # [25][10000][1]; pixels outside a run's i1..i2 are zero (camera.f90:60-62)
    ab = np.zeros(r["ab"].shape)                           # pixloc is NOT initialised outside i1..i2
    for (a, b), p_ in zip(strips, parts):
        iv[:, a - 1:b] = p_["ivals"][:, a - 1:b]
        ab[a - 1:b] = p_["ab"][a - 1:b]
    terr_i = np.abs(iv[14, :, 0] - g["intensity"]).sum() / np.abs(g["intensity"]).sum()
    da = ab[1, 0] - ab[0, 0]
    spec = np.array([[np.sum(iv[i, :, 0] * ab[:, 0]) * da * 2. * 3.14] for i in range(25)])     # calc_spec, ny = 1 branch
    terr_s = np.abs(spec - g["spectrum"]).sum() / np.abs(g["spectrum"]).sum()
    print("SPHACC intensity terr = %.4f, spectrum terr = %.4f (pass mark 0.1)" % (terr_i, terr_s))
    assert terr_i < 0.1 and terr_s < 0.1


def test_own_driver_gives_the_bits_of_the_committed_image_goldens():
    g = np.load(os.path.join(G, "ffjet_32x32x100.npz"))
    for iname in ("formal", "delo"):
        r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], iname=iname), nthreads=8, writer="mirror")
        assert np.array_equal(r["ivals"][0], g["ivals_" + iname]), iname
    import json
    rec = json.load(open(os.path.join(G, "refmain_crosscheck.json")))      # the full-size goldens, checked when they were made
    assert rec and all(rec.values())


def test_extra_goldens_come_from_the_references_own_driver():
    g = np.load(os.path.join(G, "extra_32x32x100.npz"))
    r = refmain.run(dict(FFJET_KW, nn=[32, 32, 100], iname='formal', extra=1), nthreads=8, writer="mirror")
    assert r["nvals"] == 23 and np.array_equal(r["ivals"][0], g["ffjet_vals"], equal_nan=True)
    r = refmain.run(dict(SPHACC_KW, extra=1), nthreads=1, writer="mirror")
    assert np.array_equal(r["ivals"][0], g["sphacc_vals"], equal_nan=True)


def test_mirror_namelist_writer_equals_the_references_writer():
    # the mirror writer (grtrans_amd.grtrans_batch) is what the tests above feed the reference with on a box without
    # /root/reference; where the reference's Python is present, both writers must produce the same run
    if not os.path.isdir(refmain.REFERENCE):
        pytest.skip("reference Python not present")
    a = refmain.run(dict(FFJET_KW, nn=[16, 16, 60], iname='delo'), nthreads=4, writer="mirror")
    b = refmain.run(dict(FFJET_KW, nn=[16, 16, 60], iname='delo'), nthreads=4, writer="reference")
    assert np.array_equal(a["ivals"], b["ivals"]) and np.array_equal(a["ab"], b["ab"])
