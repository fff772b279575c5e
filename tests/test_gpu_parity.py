"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
 (a) golden vectors produced by the compiled unmodified reference (tests/golden), and
 (b) the C++ oracle on the same inputs.
Bars: stored-point counts (npts) bit-exact; fp32 Stokes images within 2e-6 of each Stokes peak.
The reference computes the fluid variables in REAL(4) and its power-law emissivity jumps by ~0.5% where the
frequency crosses a bin of the G(x) tables, so the REAL(4) islands must be bit-identical on the device (built with
-ffp-contract=fast-honor-pragmas; test_fp32_islands_* and test_table_bin_flip_cameras hold that)."""
import os
import sys
import re
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FREQ = 3.45e11
STOKES_TOL = 2e-6      # measured on MI355X with bit-identical REAL(4) islands: <= 6.6e-7 of peak over every case below


@pytest.fixture(scope="module")
def dev():
    from grtrans_amd import capi
    d = capi.Device(0)
    d.load_ffjet()
    d.load_tables()
    yield d
    d.close()


@pytest.fixture
def fdev():
    # a context that is certainly up: the host tests (./grtrans in-process through module pgrtrans) end with grt_free
    from grtrans_amd import capi
    d = capi.Device(0)
    d.load_ffjet()
    d.load_tables()
    return d


def free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def run(nn, iname, i1=None, i2=None, freqs=(FREQ,)):
    from grtrans_amd import capi
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    i1 = i1 or 1
    i2 = i2 or nn[0] * nn[1]
    return capi.trace_tile(cam, i1, i2, "FFJET", em, list(freqs), iname)


def parity_stats(pix, ref, floor=1e-6):
    """Sharper statements than "of the image peak" about two fp32 Stokes images: the fraction of values that are the SAME BITS,
    and over the values above `floor` of their Stokes parameter's peak (a pixel at 1e-4 of the peak must not be 2 % wrong
    unnoticed) the largest distance in fp32 units in the last place and the largest relative error."""
    a = np.ascontiguousarray(pix, dtype=np.float32).reshape(-1, pix.shape[-1])
    b = np.ascontiguousarray(ref, dtype=np.float32).reshape(-1, ref.shape[-1])
    same = float(((a.view(np.int32) == b.view(np.int32)) | ((a == 0) & (b == 0))).mean())

    def ordered(x):      # fp32 bit patterns as integers that order like the floats
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7fffffff), i)
    peak = np.abs(b.astype(float)).max(axis=0)
    sig = np.abs(b.astype(float)) > floor * peak
    fin = np.isfinite(a) & np.isfinite(b) & sig
    ulp = int(np.abs(ordered(a) - ordered(b))[fin].max()) if fin.any() else 0
    rel = float((np.abs(a.astype(float) - b.astype(float))[fin] / np.abs(b.astype(float))[fin]).max()) if fin.any() else 0.0
    return dict(bit_identical=same, max_ulp=ulp, max_rel=rel, n_sig=int(fin.sum()))


def stokes_err(pix, ref, min_identical=None, max_rel=None):
    """max |device - reference| per Stokes parameter over its image peak; with min_identical / max_rel also asserts the
    sharper statements of parity_stats (floors set from the values measured on MI355X, printed in DESIGN.md 6)."""
    ps = parity_stats(pix, ref)
    ref = ref.astype(float)
    peak = np.abs(ref).reshape(-1, ref.shape[-1]).max(axis=0)
    e = (np.abs(pix.astype(float) - ref).reshape(-1, ref.shape[-1]).max(axis=0) / peak).max()
    if os.environ.get("GRT_TEST_ERRLOG"):      # measured margins, for setting the bars (tools/)
        import inspect
        with open(os.environ["GRT_TEST_ERRLOG"], "a") as f:
            f.write("%d %s %.3e identical=%.5f max_ulp=%d max_rel=%.3e n_sig=%d\n" % (inspect.stack()[1].lineno, inspect.stack()[1].function, e,
                                                                                  ps["bit_identical"], ps["max_ulp"], ps["max_rel"], ps["n_sig"]))
    if min_identical is not None:
        assert ps["bit_identical"] >= min_identical, ps
    if max_rel is not None:
        assert ps["max_rel"] <= max_rel, ps
    return e


@pytest.mark.parametrize("nx,nup", [(32, 100), (100, 400)])
@pytest.mark.parametrize("iname", ["formal", "delo"])
def test_ffjet_vs_reference_golden(dev, nx, nup, iname):
    g = np.load(os.path.join(G, "ffjet_%dx%dx%d.npz" % (nx, nx, nup)))
    pix, loc, status, st = run((nx, nx, nup), iname)
    assert np.array_equal(status, g["npts"])                 # bit-exact stored-point counts
    assert st.ray_steps == int(g["npts"].sum())
    # against the compiled reference: the floor on the fraction of fp32 values with the SAME BITS is what r04 measured minus two points
    # (100 x 100 x 400: 0.8242 formal / 0.8250 delo; 32 x 32 x 100: 0.7986 / 0.8013); no value above 1e-6 of its Stokes peak is off by
    # more than 1e-4 relative (measured <= 2.8e-5 = 333 fp32 ulp, on Q/U/V next to their zero crossings).  Where the other 18 % come
    # from (tools/attribute_fp32_flips.py, profiles/r04_c_fp32_flip_attribution.log): r and theta of the device's geodesics equal the
    # reference's to 3e-16 / 2e-15, but the running intensity of a ray differs by 2e-10 ... 5e-8 of its maximum -- median 4e-9, i.e. 7 % of
    # an fp32 rounding interval for I and more for Q, U, V, which are differences -- accumulated over 50-200 steps, none of which adds more
    # than a fifth of it: the affine parameter.  GEOKERR's lambda carries a per-ray constant of ~1e3 ... 1e5, so d(lambda) between
    # neighbouring points (1e-2 ... 1) has 1e-9 ... 1e-6 of relative rounding noise IN THE REFERENCE; the fast classes evaluate lambda by
    # other formulas (segment sums, Zeta function) that are as accurate but round differently, the literal restatement (oracle) rounds
    # like the reference and reaches 99.5 %.  (On seven of the nine dumped rays the thin-limit sum of j (dlambda_dev - dlambda_ref)
    # reproduces the final difference within a factor of a few.  The two corner pixels of the image do not fit: 4-5e-8, gathered in four
    # or five single steps of 1e-8 at r = 44 where the device's coefficient code, fed the reference's own fluid variables on the host,
    # agrees with the reference to 1e-15 -- a single-sample difference of the inputs there, not located further.)
    assert stokes_err(pix[0], g["ivals_" + iname], min_identical=0.80 if nx == 100 else 0.775, max_rel=1e-4) < STOKES_TOL
    assert np.array_equal(loc[:, 0], g["cam_alpha"].astype(np.float32))
    assert np.array_equal(loc[:, 1], g["cam_beta"].astype(np.float32))


def test_ffjet_vs_oracle_other_camera(dev):
    # a camera the fixtures do not hold: oracle on the same inputs
    from oracle import orabind
    nn = (48, 40, 160)
    o = orabind.Oracle(nn=nn, iname="formal")
    ref, rn, _ = o.trace(1, nn[0] * nn[1], [FREQ], nthreads=0)
    pix, loc, status, st = run(nn, "formal")
    assert np.array_equal(status, rn)
    assert stokes_err(pix, ref) < STOKES_TOL


CAMERA_SWEEP = [(0.1, 0.95, (-20., 20., -20., 20.), 0.02), (0.7, 0.0, (-15., 15., -15., 15.), 0.03),
                (0.998, 1.0, (-20., 20., -20., 20.), 0.02), (0.9, -0.8, (-30., 10., -10., 30.), 0.015),
                (0.3, 0.2, (-12., 12., -12., 12.), 0.05), (0.0, -0.4, (-18., 18., -18., 18.), 0.03)]


@pytest.mark.parametrize("spin,mu0,gv,uout", CAMERA_SWEEP)
def test_camera_sweep_vs_oracle(dev, spin, mu0, gv, uout):
    # slow and fast rotators, equatorial (mu0=0), pole-on (mu0=1), southern (mu0<0) and Schwarzschild cameras: the
    # fast classes, the generic path and the mu-fill logic against the oracle (itself equal to the compiled reference
    # on these cameras to <1e-8 of peak, checked with oracle/refbind.py)
    from grtrans_amd import capi
    from oracle import orabind
    nn = (24, 24, 80)
    for iname in ("formal", "delo"):
        o = orabind.Oracle(nn=nn, iname=iname, spin=spin, mu0=mu0, gridvals=gv, uout=uout)
        ref, rn, _ = o.trace(1, nn[0] * nn[1], [FREQ], nthreads=0)
        cam = capi.make_camera(1, gv, nn, spin, mu0, uout, -0.5)
        em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
        for rec in (False, True):
            capi.record_steps(rec)
            pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [FREQ], iname)
            assert np.array_equal(status, rn)
            assert stokes_err(pix, ref) < STOKES_TOL, (iname, rec)
    capi.record_steps(False)


def test_tiles_equal_full_image(dev):
    # i1..i2 semantics (read_inputs.f90:8-9): a tile reproduces exactly those pixels of the full run
    nn = (32, 32, 100)
    full, _, sfull, _ = run(nn, "formal")
    a, _, sa, _ = run(nn, "formal", 1, 300)
    b, _, sb, _ = run(nn, "formal", 301, 1024)
    assert np.array_equal(np.concatenate([a, b], axis=1), full)
    assert np.array_equal(np.concatenate([sa, sb]), sfull)


def test_trace_ranges_one_launch_equals_per_range_calls(dev, monkeypatch):
    # grt_trace_ranges: several pixel ranges (a rank's column tiles, ragged ends) in ONE set of launches give exactly the rows
    # that one grt_trace_tile call per range gives, in range order -- image, status, ray-steps, per-step record; also under the
    # multi-pass memory cap, with two frequencies, and with the resident geodesic record re-used for the next parameter set
    from grtrans_amd import capi, tiling
    nn = (40, 24, 100)
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    rng = tiling.ranges(1, 3, 7, nn[0] * nn[1] - 5, nn[1], 3)          # rank 1 of 3, tiles of 3 columns, i1 = 7, i2 = npix - 5
    assert len(rng) >= 4
    for iname, freqs in (("formal", [FREQ]), ("delo", [2.3e11, FREQ])):
        parts = [capi.trace_tile(cam, a, b, "FFJET", em, freqs, iname) for (a, b) in rng]
        pix, loc, status, st = capi.trace_ranges(cam, rng, "FFJET", em, freqs, iname)
        assert np.array_equal(pix, np.concatenate([p[0] for p in parts], axis=1))
        assert np.array_equal(loc, np.concatenate([p[1] for p in parts], axis=0))
        assert np.array_equal(status, np.concatenate([p[2] for p in parts]))
        assert st.ray_steps == sum(p[3].ray_steps for p in parts)
        monkeypatch.setenv("GRT_MAX_WORK_BYTES", str(1024 * 1024))
        pix2, _, status2, st2 = capi.trace_ranges(cam, rng, "FFJET", em, freqs, iname)
        monkeypatch.delenv("GRT_MAX_WORK_BYTES")
        assert np.array_equal(pix2, pix) and np.array_equal(status2, status) and st2.ray_steps == st.ray_steps
    # many narrow ranges (bench.py's default share-out: single image columns round-robin; 128 ranges per rank at 8 GPUs)
    cam2 = capi.make_camera(1, (-40., 20., -20., 40.), (96, 16, 60), 0.998, 0.906, 0.01, -0.5)
    rng2 = tiling.ranges(2, 4, 1, 96 * 16, 16, 1)
    assert len(rng2) == 24
    parts = [capi.trace_tile(cam2, a, b, "FFJET", em, [FREQ], "delo") for (a, b) in rng2]
    pix, _, status, st = capi.trace_ranges(cam2, rng2, "FFJET", em, [FREQ], "delo")
    assert np.array_equal(pix, np.concatenate([p[0] for p in parts], axis=1)) and np.array_equal(status, np.concatenate([p[2] for p in parts]))
    # per-step record: local ray j of the call is ray j of the record
    capi.record_steps(True)
    pix, _, status, _ = capi.trace_ranges(cam, rng, "FFJET", em, [FREQ], "formal")
    j = int(np.argmax(status > 0)); n = int(status[j])
    rec = capi.download_steps(j, 1, cam.nup + cam.next)[:, 0, :n]
    off = 0
    for (a, b) in rng:
        if j < off + b - a + 1:
            pixel = a + j - off
            break
        off += b - a + 1
    capi.trace_tile(cam, pixel, pixel, "FFJET", em, [FREQ], "formal")
    one = capi.download_steps(0, 1, cam.nup + cam.next)[:, 0, :n]
    assert np.array_equal(rec, one)
    capi.record_steps(False)
    # resident geodesics: same camera + same ranges -> only the transfer kernels run; other ranges -> geodesics again
    capi.geo_reuse(True)
    p0 = capi.lib().grt_geo_passes()
    a1 = capi.trace_ranges(cam, rng, "FFJET", em, [FREQ], "formal")[0]
    p1 = capi.lib().grt_geo_passes()
    em2 = capi.make_emis("POLSYNCHPL", 3.4e9, 4.0, 70.0, 100., 1e5, 3.5)
    a2 = capi.trace_ranges(cam, rng, "FFJET", em2, [FREQ], "formal")[0]
    p2 = capi.lib().grt_geo_passes()
    a3 = capi.trace_ranges(cam, rng[:-1], "FFJET", em2, [FREQ], "formal")[0]
    p3 = capi.lib().grt_geo_passes()
    assert p1 == p0 + 1 and p2 == p1 and p3 == p2 + 1
    assert not np.array_equal(a1, a2) and np.array_equal(a3, a2[:, :a3.shape[1]])


def test_merged_and_per_class_transfer_launches_are_bit_identical(dev):
    # r06 (grt_set_class_merge): a pass of at most 400,000 rays -- a rank's share of a multi-GPU image, and every camera of this
    # suite -- walks both ray classes in ONE transfer launch (fast class from the front of live[], generic class from its back); a whole
    # image keeps one launch per class on two streams.  Same kernel, same per-lane arithmetic: image, status, ray-steps and the per-step
    # record must be the same bits either way.  (The differential-compilation sweep also runs its second object with GRT_CLASS_MERGE=0.)
    from grtrans_amd import capi, tiling
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    cases = []
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (48, 40, 120), 0.998, 0.906, 0.01, -0.5)      # the headline camera: generic rays around alpha = 0
    cases.append((cam, [(1, 48 * 40)], "FFJET", em, [FREQ], "formal", True))
    cases.append((cam, tiling.ranges(1, 3, 7, 48 * 40 - 5, 40, 3), "FFJET", em, [2.3e11, FREQ, 9e11], "delo", False))
    cam0 = capi.make_camera(1, (-12., 12., -12., 12.), (40, 40, 100), 0.0, 0.5, 0.02, -0.5)          # Schwarzschild: cubic class + generic class
    cases.append((cam0, [(1, 1600)], "FFJET", em, [FREQ], "formal", True))
    camp = capi.make_camera(1, (-6., 6., -6., 6.), (40, 40, 100), 0.9, 0.999, 0.02, -0.5)            # nearly pole-on: many near-polar (generic) rays
    cases.append((camp, [(1, 1600)], "FFJET", em, [FREQ], "delo", True))
    try:
        for cam_, rng, fname, em_, freqs, iname, rec in cases:
            got = []
            for mode in (0, 1, -1):
                capi.class_merge(mode)
                capi.record_steps(rec)
                pix, _, status, st = capi.trace_ranges(cam_, rng, fname, em_, freqs, iname)
                r = capi.download_steps(0, min(len(status), 512), cam_.nup + cam_.next) if rec else None
                got.append((pix, status, st.ray_steps, r))
            for g in got[1:]:
                assert np.array_equal(g[0], got[0][0], equal_nan=True) and np.array_equal(g[1], got[0][1]) and g[2] == got[0][2]
                if rec:
                    assert np.array_equal(g[3], got[0][3], equal_nan=True)
    finally:
        capi.class_merge(-1); capi.record_steps(False)


def test_multi_frequency_matches_single(dev):
    nn = (24, 24, 100)
    freqs = [2.3e11, 3.45e11, 6.9e11]
    multi, _, _, _ = run(nn, "formal", freqs=freqs)
    for k, f in enumerate(freqs):
        single, _, _, _ = run(nn, "formal", freqs=[f])
        assert np.array_equal(multi[k], single[0])


@pytest.mark.parametrize("iname,nf", [("formal", 15), ("delo", 10)])
def test_frequency_batches_match_single_runs(dev, iname, nf):
    # more frequencies than one batch of the transfer kernel holds (13 formal / 8 DELO): every camera equals its own run
    nn = (16, 16, 80)
    freqs = list(2.41e10 * np.exp(np.arange(nf) * np.log(6.31e14 / 2.41e10) / (nf - 1)))
    multi, _, _, _ = run(nn, iname, freqs=freqs)
    for k in (0, 7, 8, nf - 3, nf - 1):
        single, _, _, _ = run(nn, iname, freqs=[freqs[k]])
        assert np.array_equal(multi[k], single[0]), k


def test_per_step_record_matches_reference_dump(dev):
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "ffjet_rays_32x32x100.npz"))
    nn = (32, 32, 100)
    capi.record_steps(True)
    try:
        pix, loc, status, st = run(nn, "formal")
        maxpts = 100 + 15
        for i in g["rays"]:
            n = int(g["ray%d_npts" % i])
            rec = capi.download_steps(int(i) - 1, 1, maxpts)[:, 0, :n]
            x = g["ray%d_x" % i]
            assert np.allclose(rec[0], x[:, 1], rtol=1e-9)                 # r
            assert np.allclose(rec[1], x[:, 2], rtol=1e-9, atol=1e-12)     # theta
            assert np.allclose(rec[3], g["ray%d_lam" % i], rtol=1e-7, atol=1e-7 * np.abs(g["ray%d_lam" % i]).max())
            # running intensity: reference stores it in integration order (far end first)
            I = g["ray%d_I" % i]
            scale = np.abs(I).max() + 1e-300
            assert np.abs(rec[4:8, ::-1].T[:n] - I).max() / scale < 1e-5
    finally:
        capi.record_steps(False)


FREQ_SPHACC = 2.41e11


def run_sphacc(nn, iname, freqs=(FREQ_SPHACC,)):
    from grtrans_amd import capi
    cam = capi.make_camera(1, (-25., 25., -25., 25.), nn, 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
    return capi.trace_tile(cam, 1, nn[0] * nn[1], "SPHACC", em, list(freqs), iname)


@pytest.mark.parametrize("iname", ["formal", "delo"])
def test_sphacc_polsynchth_vs_reference_golden(dev, iname):
    # BASELINE config 2 (spin 0 -> cubic root cases, thermal coefficients, 1-D Bondi tables in LDS)
    dev.load_sphacc()
    g = np.load(os.path.join(G, "sphacc_64x64x200.npz"))
    pix, loc, status, st = run_sphacc((64, 64, 200), iname)
    assert parity_stats(pix[0], g["ivals_" + iname])["bit_identical"] >= 0.90      # measured 0.936 (formal) / 0.988 (delo)
    assert parity_stats(pix[0], g["ivals_" + iname])["max_rel"] <= 3e-5             # measured 6.8e-6 over values above 1e-6 of peak
    assert np.array_equal(status, g["npts"])
    assert st.rays_dead == int((g["npts"] == 0).sum())       # image corners outside uout: compacted away
    assert stokes_err(pix[0], g["ivals_" + iname]) < STOKES_TOL


def test_sphacc_per_step_record(dev):
    from grtrans_amd import capi
    dev.load_sphacc()
    g = np.load(os.path.join(G, "sphacc_rays_64x64x200.npz"))
    capi.record_steps(True)
    try:
        run_sphacc((64, 64, 200), "formal")
        for i in g["rays"]:
            n = int(g["ray%d_npts" % i])
            if n == 0:
                continue
            rec = capi.download_steps(int(i) - 1, 1, 230)[:, 0, :n]
            x = g["ray%d_x" % i]
            assert np.allclose(rec[0], x[:, 1], rtol=1e-9)
            assert np.allclose(rec[1], x[:, 2], rtol=1e-9, atol=1e-12)
            I = g["ray%d_I" % i]
            assert np.abs(rec[4:8, ::-1].T[:n] - I).max() / (np.abs(I).max() + 1e-300) < 1e-5
    finally:
        capi.record_steps(False)


FREQ_HARM = 2.3e11


@pytest.fixture(scope="module")
def harm_dir(tmp_path_factory):
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_harm_dump
    g = np.load(os.path.join(G, "harm_48x48x200.npz"))
    d = tmp_path_factory.mktemp("harm")
    path = make_harm_dump.write(str(d), n=int(g["nzone"]))
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(g["dump_sha256"])
    return str(d)


def run_harm(dev, harm_dir, nn, iname):
    from grtrans_amd import capi
    dev.load_harm(os.path.join(harm_dir, "dump"), os.path.join(harm_dir, "dump040"), 40, 0.9375)
    cam = capi.make_camera(1, (-13., 13., -13., 13.), nn, 0.9375, 0.6428, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5, muval=0.25, mdot=1.57e15)
    return capi.trace_tile(cam, 1, nn[0] * nn[1], "HARM", em, [FREQ_HARM], iname)


@pytest.mark.parametrize("iname", ["formal", "delo"])
def test_harm_polsynchth_vs_reference_golden(dev, harm_dir, iname):
    # BASELINE config 4: 2-D GRMHD slice (synthetic HARM-format dump read by the product's own loader),
    # Brent theta->x2 inversion per point, thermal coefficients, Moscibrodzka electron temperatures
    g = np.load(os.path.join(G, "harm_48x48x200.npz"))
    pix, loc, status, st = run_harm(dev, harm_dir, (48, 48, 200), iname)
    assert np.array_equal(status, g["npts"])
    # measured: 96 % of the fp32 values bit-identical to the compiled reference, <= 1.5e-6 relative (22 ulp) above 1e-6 of peak
    assert stokes_err(pix[0], g["ivals_" + iname], min_identical=0.93, max_rel=1e-5) < STOKES_TOL


def test_record_download_across_the_64_ray_blocks_of_its_stokes_fields(dev):
    # On the device I,Q,U,V of the per-step record are stored [block of 64 rays][point][ray in block] (grt_transfer_kernel writes them
    # from the registers); grt_download_steps hands out [8][nrays][maxpts].  A 150-ray pass (two whole blocks and a ragged third one):
    # every multi-ray window equals the single-ray downloads, the last point of a ray is its pixel, a one-ray trace gives the same row.
    from grtrans_amd import capi
    dev.load_ffjet()
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (15, 10, 60), 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    capi.record_steps(True)
    try:
        for iname in ("formal", "delo"):
            pix, _, status, _ = capi.trace_tile(cam, 1, 150, "FFJET", em, [FREQ], iname)
            mp = cam.nup + cam.next
            whole = capi.download_steps(0, 150, mp)
            assert whole.shape == (8, 150, mp)
            for (r0, nr) in ((0, 64), (60, 10), (63, 2), (64, 64), (100, 50), (149, 1), (1, 148)):
                assert np.array_equal(capi.download_steps(r0, nr, mp), whole[:, r0:r0 + nr]), (iname, r0, nr)
            live = np.flatnonzero(status > 0)
            assert live.size > 100
            for j in live[::7]:
                # stored point 1 is the camera end: the running Stokes vector there is the pixel (fp32 in the image)
                assert np.array_equal(whole[4:8, j, 0].astype(np.float32), pix[0, j]), (iname, j)
            j = int(live[live.size // 2])
            capi.trace_tile(cam, j + 1, j + 1, "FFJET", em, [FREQ], iname)
            n = int(status[j])      # (slots behind npts keep what the buffer held)
            assert np.array_equal(capi.download_steps(0, 1, mp)[:, 0, :n], whole[:, j, :n]), iname
    finally:
        capi.record_steps(False)


def test_harm_per_step_record(dev, harm_dir):
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "harm_rays_48x48x200.npz"))
    capi.record_steps(True)
    try:
        run_harm(dev, harm_dir, (48, 48, 200), "formal")
        for i in g["rays"]:
            n = int(g["ray%d_npts" % i])
            rec = capi.download_steps(int(i) - 1, 1, 230)[:, 0, :n]
            x = g["ray%d_x" % i]
            assert np.allclose(rec[0], x[:, 1], rtol=1e-9)
            assert np.allclose(rec[1], x[:, 2], rtol=1e-9, atol=1e-12)
            I = np.nan_to_num(g["ray%d_I" % i])
            assert np.abs(np.nan_to_num(rec[4:8, ::-1].T[:n]) - I).max() / (np.abs(I).max() + 1e-300) < 1e-5
    finally:
        capi.record_steps(False)


def test_harm_loader_refuses_wrong_spin(dev, harm_dir):
    from grtrans_amd import capi
    with pytest.raises(capi.GrtError):
        dev.load_harm(os.path.join(harm_dir, "dump"), os.path.join(harm_dir, "dump040"), 40, 0.5)


def test_nvals1_quadrature_vs_reference_golden(dev):
    # nvals=1: Stokes I only, quadrature integrator whatever iname says (radtrans_integrate.f90:176-189)
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "nvals1_32x32x100.npz"))
    dev.load_sphacc()
    for iname in ("formal", "delo"):
        cam = capi.make_camera(1, (-40., 20., -20., 40.), (32, 32, 100), 0.998, 0.906, 0.01, -0.5)
        em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "FFJET", em, [FREQ], iname, nvals=1)
        assert pix.shape == (1, 1024, 1) and np.array_equal(status, g["ffjet_npts"])
        assert stokes_err(pix, g["ffjet_ivals"]) < STOKES_TOL
        cam = capi.make_camera(1, (-25., 25., -25., 25.), (32, 32, 100), 0.0, 0.5, 0.04, -0.5)
        em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "SPHACC", em, [FREQ_SPHACC], iname, nvals=1)
        assert np.array_equal(status, g["sphacc_npts"]) and stokes_err(pix, g["sphacc_ivals"]) < STOKES_TOL
    # SYNCHPL: the Stokes-I half of POLSYNCHPL with its own constants (same kernel instantiation, em.type selects); nvals = 1 only
    from grtrans_amd import capi
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (32, 32, 100), 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("SYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "FFJET", em, [FREQ], "formal", nvals=1)
    assert np.array_equal(status, g["ffjet_npts"])
    e = np.abs(pix - g["ffjet_synchpl_ivals"]).max() / np.abs(g["ffjet_synchpl_ivals"]).max()
    e_pol = np.abs(pix - g["ffjet_ivals"]).max() / np.abs(g["ffjet_synchpl_ivals"]).max()      # the two goldens differ by 1.2e-7
    assert e < STOKES_TOL and e < 0.6 * e_pol, (e, e_pol)
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 1024, "FFJET", em, [FREQ], "formal")
    # the reference's "polunpol" regression problem (run_grtrans_test_problems_public.py:127-143): SYNCHPL nvals = 1 at
    # 100x100x400 against Stokes I of the shipped test_grtrans_ffjet.p.  Its pass mark 1e-2 is for LSODA on both sides; the
    # fixed-step quadrature at nup = 400 is 1.2 % from the LSODA vector (formal / DELO: 2 %)
    gp = np.load(os.path.join(G, "test_grtrans_ffjet_lsoda.npz"))["ivals"].astype(float)
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (100, 100, 400), 0.998, 0.906, 0.01, -0.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, 10000, "FFJET", em, [FREQ], "formal", nvals=1)
    terr = np.abs(pix[0, :, 0] - gp[:, 0]).sum() / np.abs(gp[:, 0]).sum()
    assert terr < 0.015, terr


def test_extra_ray_diagnostics_vs_reference(dev):
    # extra=1: the 19 per-ray diagnostics (optical depths, emission- and LP-weighted means) accumulated in the
    # transfer kernel's walk; golden = compiled reference routines through oracle/ref_driver.f90
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "extra_32x32x100.npz"))
    dev.load_sphacc()

    # column tolerances (fraction of the column's largest magnitude): Stokes and optical depths 1e-5; the
    # emission-weighted means 5e-5 (ratios of two attenuated sums; beta goes through REAL(4) pow); the LP-increment
    # weighted means 1e-4: their weights are differences of neighbouring sqrt(I^2+Q^2), which amplify the few-1e-7
    # differences of the fp32 fluid islands
    tolc = np.array([1e-5] * 4 + [1e-5] * 6 + [5e-5] * 8 + [1e-4] * 5)

    def check(vals, ref, tol):
        a = np.asarray(vals, float); b = np.asarray(ref, float)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        a = np.nan_to_num(a); b = np.nan_to_num(b)
        err = np.abs(a - b).max(axis=0) / (np.abs(b).max(axis=0) + 1e-300)
        assert np.all(err < tolc), err
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (32, 32, 100), 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "FFJET", em, [FREQ], "formal", nvals=4, nextra=19)
    assert pix.shape == (1, 1024, 23) and np.array_equal(status, g["ffjet_npts"])
    check(pix[0], g["ffjet_vals"], 2e-5)
    # two frequencies: each camera carries its own diagnostics
    pix2, _, _, _ = capi.trace_tile(cam, 1, 1024, "FFJET", em, [2.3e11, FREQ], "formal", nvals=4, nextra=19)
    assert np.array_equal(pix2[1], pix[0])
    cam = capi.make_camera(1, (-25., 25., -25., 25.), (32, 32, 100), 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "SPHACC", em, [FREQ_SPHACC], "formal", nvals=4, nextra=19)
    assert np.array_equal(status, g["sphacc_npts"])
    check(pix[0], g["sphacc_vals"], 2e-5)


def test_thindisk_bb_config1_vs_reference_golden(dev):
    # BASELINE config 1 (harm.in.dist): standard=2 camera, one point per ray on a Novikov-Thorne disk, blackbody, nvals=1
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "thindisk_100x100x1.npz"))
    dev.set_thindisk(0.1, -1.0, 1e8)
    em = capi.make_emis("BB", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
    for mu0 in (0.1, 0.7):
        cam = capi.make_camera(2, (-25., 25., -25., 25.), (100, 100, 1), 0.998, mu0, 1e-4, -0.5)
        pix, _, status, st = capi.trace_tile(cam, 1, 10000, "THINDISK", em, [1e17], "lsoda", nvals=1)
        tag = "mu%02d" % int(mu0 * 10)
        assert np.array_equal(status, g["npts_" + tag])
        ref = g["ivals_" + tag].astype(float)
        assert np.abs(pix[0] - ref).max() / ref.max() < STOKES_TOL
        assert st.ray_steps == int((status > 0).sum())
    # anything else with standard=2 is refused
    cam = capi.make_camera(2, (-25., 25., -25., 25.), (10, 10, 50), 0.998, 0.5, 1e-4, -0.5)
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 100, "THINDISK", em, [1e17], "formal", nvals=1)


def test_library_refuses_unsupported(dev):
    from grtrans_amd import capi
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (8, 8, 50), 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 64, "FFJET", em, [FREQ], "lsoda")   # LSODA stays CPU-only
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 64, "SPHACC", em, [FREQ], "formal")  # no (SPHACC, POLSYNCHPL) kernel: refused, not emulated


def test_fortran_host_exe_and_harness_end_to_end(tmp_path, monkeypatch):
    # files.in -> ./grtrans (Fortran 90 host, ISO_C_BINDING -> C ABI -> kernels) -> binary output file
    # -> reader; and the same problem through run_pgrtrans.  Both against the reference golden image.
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "ffjet_32x32x100.npz"))
    kw = dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=1, nmu=1, fmin=FREQ, fmax=FREQ,
              ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[32, 32, 100], uout=0.01, mbh=3.4e9,
              mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70., iname='formal', cflag=0)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (1024, 4, 1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal"]) < STOKES_TOL
    assert np.array_equal(x.ab[:, 0], g["cam_alpha"].astype(np.float32))
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert y.ivals.shape == (4, 1024, 1) and y.ab.shape == (2, 1024)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))
    y.calc_spec_pgrtrans(1)
    x.calc_spec(1)
    assert np.allclose(y.spec[:, 0], x.spec[:, 0], rtol=1e-6)


def test_fortran_host_sphacc_extra_end_to_end(tmp_path, monkeypatch):
    # ./grtrans with fname=SPHACC, ename=POLSYNCHTH, extra=1: 4 Stokes + 19 diagnostics per pixel in the output file,
    # identical through the Fortran host and the Python harness, and equal to the reference-driver golden
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "extra_32x32x100.npz"))
    kw = dict(fname='SPHACC', nfreq=1, nmu=1, fmin=FREQ_SPHACC, fmax=FREQ_SPHACC, ename='POLSYNCHTH', nvals=4, spin=0., mbh=10.,
              standard=1, nn=[32, 32, 100], uout=0.04, mumin=.5, mumax=.5, gridvals=[-25, 25, -25, 25], iname='formal',
              cflag=0, extra=1)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (1024, 23, 1)
    ref = np.nan_to_num(g["sphacc_vals"].astype(float))
    got = np.nan_to_num(x.ivals[:, :, 0].astype(float))
    err = np.abs(got - ref).max(axis=0) / (np.abs(ref).max(axis=0) + 1e-300)
    assert err.max() < 1e-4, err
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert y.ivals.shape == (23, 1024, 1)
    assert np.array_equal(np.nan_to_num(y.ivals[:, :, 0].T).astype(np.float32), np.nan_to_num(x.ivals[:, :, 0]).astype(np.float32))


def test_edge_cases_single_pixel_dead_tile_small_nup_and_max_nfreq():
    from grtrans_amd import capi
    from oracle import orabind
    dev = capi.Device(0)        # (the harness tests above end with grt_free: bring the context up again)
    dev.load_ffjet(); dev.load_tables(); dev.load_sphacc()
    # (a) a one-pixel tile equals that pixel of the full image
    full, _, sfull, _ = run((32, 32, 100), "formal")
    one, _, sone, _ = run((32, 32, 100), "formal", 517, 517)
    assert np.array_equal(one[0, 0], full[0, 516]) and sone[0] == sfull[516]
    # (b) a tile in which no ray ever enters r < 1/uout (image corner of the SPHACC camera): zero pixels, status 0,
    #     nothing launched for it (grtrans_driver.f90:452-462)
    cam = capi.make_camera(1, (-25., 25., -25., 25.), (64, 64, 200), 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)
    pix, _, status, st = capi.trace_tile(cam, 1, 4, "SPHACC", em, [FREQ_SPHACC], "formal")
    assert not pix.any() and not status.any() and st.ray_steps == 0 and st.rays_dead == 4
    # (c) short rays: nup=10 (kext=1, next=2, 10 stored points for turning rays) against the oracle
    nn = (16, 16, 10)
    o = orabind.Oracle(nn=nn, iname="delo")
    ref, rn, _ = o.trace(1, 256, [FREQ], nthreads=0)
    pix, _, status, _ = run(nn, "delo")
    assert np.array_equal(status, rn) and stokes_err(pix, ref) < STOKES_TOL
    # (d) 32 frequencies (round 1's limit) and one more: there is no limit (pgrtrans.f90:87-91)
    freqs = list(2.41e10 * np.exp(np.arange(32) * np.log(6.31e14 / 2.41e10) / 31))
    multi, _, _, _ = run((12, 12, 60), "formal", freqs=freqs)
    single, _, _, _ = run((12, 12, 60), "formal", freqs=[freqs[17]])
    assert multi.shape == (32, 144, 4) and np.array_equal(multi[17], single[0])
    more, _, _, _ = run((12, 12, 60), "formal", freqs=freqs + [1e15])
    assert more.shape == (33, 144, 4) and np.array_equal(more[:32], multi)
    with pytest.raises(capi.GrtError):
        run((12, 12, 60), "formal", freqs=[])
    dev.close()


def test_fortran_host_harm_end_to_end(tmp_path, monkeypatch, harm_dir):
    # ./grtrans with fname=HARM: the namelist's hdfile/hhfile/hindf reach grt_load_harm_files through the Fortran host;
    # golden = compiled reference on the same synthetic dump
    import shutil
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "harm_48x48x200.npz"))
    monkeypatch.chdir(tmp_path)
    shutil.copy(os.path.join(harm_dir, "dump040"), tmp_path / "dump040")      # relative names, as the reference expects
    kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375,
              standard=1, nn=[48, 48, 200], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428,
              mumax=.6428, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='delo', cflag=0, hdfile='dump',
              hhfile='dump040', hindf=40, hnt=1)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (48 * 48, 4, 1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_delo"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))


def test_fortran_host_config1_thindisk_ten_inclinations(tmp_path, monkeypatch):
    # the reference's harm.in.dist (BASELINE config 1) through ./grtrans: ten cameras (nmu=10, mu0 = 0.1..1.0), one per
    # inclination, THINDISK + BB, standard=2, nn=[100,100,1], iname='lsoda' (no integrator runs for one point per ray)
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "thindisk_100x100x1.npz"))
    kw = dict(standard=2, spin=0.998, mumin=0.1, mumax=1.0, nmu=10, gridvals=[-25., 25., -25., 25.], nn=[100, 100, 1],
              fname='THINDISK', ename='BB', nfreq=1, fmin=1e17, fmax=1e17, mbh=10., tmdot=0.1, nvals=1, iname='lsoda',
              cflag=0, uout=1e-4)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (10000, 1, 10)
    for cam, tag in ((0, "mu01"), (6, "mu07")):           # mu0 = 0.1 + 0.1*k
        ref = g["ivals_" + tag].astype(float)
        assert np.abs(x.ivals[:, :, cam] - ref).max() / ref.max() < STOKES_TOL, tag
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert y.ivals.shape == (1, 10000, 10)
    assert np.array_equal(y.ivals[0].astype(np.float32), x.ivals[:, 0, :].astype(np.float32))


def test_harm_movie_frames_through_both_hosts(tmp_path, monkeypatch):
    # nt=3, dt=30 with one loaded slice: advance_harm_timestep walks the dump index 40 -> 41 -> 43 (tstep=20, the
    # left-over 10 M accumulates); golden frames = compiled reference with advance_fluid_timestep between traces
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "harm_movie_32x32x100.npz"))
    monkeypatch.chdir(tmp_path)
    paths = make_golden.write_movie_dumps(str(tmp_path))
    assert [hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)] == list(g["dump_sha256"])
    kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375,
              standard=1, nn=[32, 32, 100], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428,
              mumax=.6428, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', cflag=0, hdfile='dump',
              hhfile='dump040', hindf=40, hnt=1, nt=3, dt=30.)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (1024, 4, 3)
    for l in range(3):
        assert stokes_err(x.ivals[:, :, l], g["ivals"][l]) < STOKES_TOL, l
    assert stokes_err(x.ivals[:, :, 1], g["ivals"][0]) > 1e-4          # the frames do differ
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(np.transpose(y.ivals, (1, 0, 2)).astype(np.float32), x.ivals.astype(np.float32))


def test_harm_slow_light_three_slices_device_and_fortran_host(tmp_path, monkeypatch):
    # fnt=3 (hnt=3): three dumps resident, every point reads the slice its Kerr-Schild look-back time selects; the
    # geo kernels carry the coordinate time (fast path: third [-2] integral), golden = compiled reference
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "harm_slowlight_32x32x100.npz"))
    monkeypatch.chdir(tmp_path)
    paths = make_golden.write_slow_dumps(str(tmp_path))
    assert [hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)] == list(g["dump_sha256"])
    dev = capi.Device(0)
    dev.load_tables()
    dev.load_harm("dump", "dump041", 41, 0.9375, nt=3)
    cam = capi.make_camera(1, (-13., 13., -13., 13.), (32, 32, 100), 0.9375, 0.6428, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5, muval=0.25, mdot=1.57e15)
    for iname in ("formal", "delo"):
        for rec in (False, True):
            capi.record_steps(rec)
            pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["npts"])
            assert stokes_err(pix[0], g["ivals_" + iname]) < STOKES_TOL, (iname, rec)
    capi.record_steps(False)
    # nload > 1: time-origin table (raw from the device, normalised as pgrtrans does), mu-fill off
    cam.mufill, cam.kext, cam.next = 0, 0, 0
    t0 = capi.normalise_t0(capi.compute_t0(cam, 1, 1024))
    assert np.abs(t0 - g["t0_nload"]).max() < 1e-6
    capi.set_t0(t0)
    pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal")
    assert np.array_equal(status, g["npts_nload"]) and stokes_err(pix[0], g["ivals_formal_nload"]) < STOKES_TOL
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 512, "HARM", em, [FREQ_HARM], "formal")          # the table belongs to another tile
    capi.set_t0(None)
    dev.close()
    kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375,
              standard=1, nn=[32, 32, 100], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428,
              mumax=.6428, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', cflag=0, hdfile='dump',
              hhfile='dump041', hindf=41, hnt=3)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal"]) < STOKES_TOL
    # the same with nload=3 through the Fortran host and the harness
    kw["nload"] = 3
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal_nload"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))


def test_extra_diagnostics_on_a_slow_light_run_device_and_fortran_host(tmp_path, monkeypatch):
    # extra=1 with several loaded HARM slices (VERDICT r02 missing #5): the 19 ray diagnostics of grtrans_driver.f90:230-292 while
    # every point reads the slice its look-back time selects (fluid_model_harm.f90:175-186).  The EXTRA instantiation keeps phi in the
    # fourth LDS tile, so the time is read straight from the record.  Golden = the reference's own grtrans_main (hnt=3, extra=1).
    import hashlib
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    g = np.load(os.path.join(G, "harm_slowlight_extra_32x32x100.npz"))
    gs = np.load(os.path.join(G, "harm_slowlight_32x32x100.npz"))
    monkeypatch.chdir(tmp_path)
    paths = make_golden.write_slow_dumps(str(tmp_path))
    assert [hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)] == list(g["dump_sha256"])
    tolc = np.array([1e-5] * 4 + [1e-5] * 6 + [5e-5] * 8 + [1e-4] * 5)      # as test_extra_ray_diagnostics_vs_reference

    def check(vals, ref):
        a = np.asarray(vals, float); b = np.asarray(ref, float)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        a = np.nan_to_num(a); b = np.nan_to_num(b)
        err = np.abs(a - b).max(axis=0) / (np.abs(b).max(axis=0) + 1e-300)
        assert np.all(err < tolc), err
    dev = capi.Device(0)
    dev.load_tables()
    dev.load_harm("dump", "dump041", 41, 0.9375, nt=3)
    cam = capi.make_camera(1, (-13., 13., -13., 13.), (32, 32, 100), 0.9375, 0.6428, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5, muval=0.25, mdot=1.57e15)
    pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal", nvals=4, nextra=19)
    assert pix.shape == (1, 1024, 23) and np.array_equal(status, gs["npts"])
    check(pix[0], g["vals"])
    assert stokes_err(pix[0][:, :4], gs["ivals_formal"]) < STOKES_TOL
    # and the slices matter: one loaded slice gives other diagnostics
    dev.load_harm("dump", "dump041", 41, 0.9375, nt=1)
    fast, _, _, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal", nvals=4, nextra=19)
    assert np.abs(np.nan_to_num(fast[0][:, 4:10]) - np.nan_to_num(pix[0][:, 4:10])).max() > 0
    dev.close()
    kw = dict(fname='HARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375,
              standard=1, nn=[32, 32, 100], uout=0.04, mbh=4e6, mdotmin=1.57e15, mdotmax=1.57e15, nmdot=1, mumin=.6428,
              mumax=.6428, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=1., iname='formal', cflag=0, hdfile='dump',
              hhfile='dump041', hindf=41, hnt=3, extra=1)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (1024, 23, 1)
    check(x.ivals[:, :, 0], g["vals"])


def _geodebug_blocks(tokens):
    """split the number stream of a geodebug.out (header 11, 49 blocks of npts, aahat 3 npts, 12 blocks of npts, fac)"""
    t = np.asarray(tokens, dtype=float)
    n = int(t[0])
    sizes = [n] * 49 + [3 * n] + [n] * 12 + [1]
    assert len(t) == 11 + sum(sizes)
    out, off = [], 11
    for sz in sizes:
        out.append(t[off:off + sz]); off += sz
    return n, t[:11], out


@pytest.mark.parametrize("name,args", [
    ("headline", (1, (-40., 20., -20., 40.), (96, 96, 400), 0.998, 0.906, 0.01, -0.5)),
    ("nup100", (1, (-40., 20., -20., 40.), (64, 64, 100), 0.998, 0.906, 0.01, -0.5)),
    ("nup40", (1, (-40., 20., -20., 40.), (64, 64, 40), 0.998, 0.906, 0.01, -0.5)),
    ("close", (1, (-13., 13., -13., 13.), (64, 64, 400), 0.9375, 0.6428, 0.04, -0.5)),
    ("retrograde near pole-on", (1, (-8., 8., -8., 8.), (64, 64, 200), -0.9, 0.99, 0.04, -0.5)),
    ("edge-on", (1, (-20., 20., -20., 20.), (64, 64, 200), 0.5, 0.0, 0.02, -0.5)),
    ("pole-on", (1, (-20., 20., -20., 20.), (48, 48, 200), 0.7, 1.0, 0.02, -0.5)),
    # (r05m) mu-fill segments of more than one 64-point chunk (next = 60 nup / 400): the in-line sum's second phase carries across chunks
    ("nup900: three fill chunks", (1, (-40., 20., -20., 40.), (40, 40, 900), 0.998, 0.906, 0.01, -0.5)),
    ("nup512 close: two fill chunks", (1, (-13., 13., -13., 13.), (40, 40, 512), 0.9375, 0.6428, 0.04, -0.5)),
    ("nup427: a fill segment of exactly 64 + 1", (1, (-30., 30., -30., 30.), (40, 40, 427), 0.7, 0.3, 0.02, -0.5)),
    # nearly Schwarzschild (k^2 of the mu side ~ 4e-8): the theta series' prefactor lost nine digits there until r05m -- on the regular points
    # of BOTH kernels, so only the mu-fill points (Carlson's forms in the fused kernel) showed it
    ("a = -0.0014", (1, (-4.65, 4.65, -4.65, 4.65), (32, 32, 427), -0.0014, -0.6009, 0.02, -0.5)),
    ("a = -0.0013, nup40", (1, (-18., 18., -18., 18.), (32, 32, 40), -0.0013, 0.6334, 0.02, -0.5))])
def test_azimuth_kernel_agrees_with_the_fused_record_mode_kernel(name, args):
    # r05: with the azimuth wanted, the four-real-root class is traced by the image-only geodesic kernel (which leaves I_u in the record)
    # and grt_phi_kernel -- u side by a two-point Hermite rule over I, mu side from the theta series -- plus the fused kernel in its
    # phi-only form on the rays that kernel leaves over (plunging rays, no short theta series, error indicator above its bar: most rays
    # at nup = 40).  grt_set_geo_split(0) traces everything with the fused kernel <1,2> (r04's): two independent evaluations of
    # GEOPHITIME's phi (geokerr_wrapper.f:1648-2357; one of them Carlson's R_J per segment) must agree to rounding on every stored
    # point, the turning-point parities and counts exactly; r, theta, lambda come from two forms of the same arithmetic.
    from grtrans_amd import capi
    dev = capi.Device(0)
    dev.load_ffjet(); dev.load_tables()
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    cam = capi.make_camera(*args)
    npx = args[2][0] * args[2][1]
    capi.record_steps(True); capi.geo_reuse(False)
    try:
        res = {}
        for split in (1, 0):
            capi.geo_split(split)
            pix, _, status, _ = capi.trace_tile(cam, 1, npx, "FFJET", em, [FREQ], "formal")
            res[split] = (pix.copy(), status.copy(), capi.download_steps(0, npx, cam.nup + cam.next))
    finally:
        capi.geo_split(1); capi.record_steps(False); capi.geo_reuse(True)
        dev.close()
    # r05m: the Hermite sum is taken IN LINE in the image-only kernel -- regular points compacted, the mu-fill segment in a phase of its own,
    # the points below the junction corrected in place afterwards (r05a-l: grt_phi_kernel<false> in plain point order behind that kernel).
    p0, s0, r0 = res[0]
    n = np.abs(s0)
    assert (n > 0).sum() > 0.5 * npx
    mask = np.arange(r0.shape[2])[None, :] < n[:, None]
    for mode in (1,):
        p1, s1, r1 = res[mode]
        assert np.array_equal(s1, s0), (name, mode)
        for f, bar in ((0, 1e-13), (1, 1e-11), (3, 1e-9)):      # r, theta, lambda (public order: r, theta, phi, lambda)
            # (lambda is the affine distance from the ray's top stored point, where it is 0: both kernels form it as a difference of two
            #  numbers the size of the ray's largest lambda, so a point is measured against its own value or 1/10 of that largest one,
            #  whichever is larger -- at nup = 512 a plunging ray's point 24 below the top differs by 9e-10 absolutely, 4e-11 of the ray's
            #  scale and 3e-8 of its own lambda, between the image-only and the fused form: the forms' conditioning, not the azimuth's)
            floor = 0.1 * np.abs(r0[f]).max(axis=1, keepdims=True) * np.ones_like(r0[f]) if f == 3 else np.full_like(r0[f], 1e-300)
            rel = np.abs(r1[f][mask] - r0[f][mask]) / np.maximum(np.abs(r0[f][mask]), floor[mask])
            assert rel.max() <= bar, (name, mode, f, rel.max())
        dphi = np.abs(r1[2] - r0[2])[mask]
        assert dphi.max() <= 1e-9 and np.quantile(dphi, 0.999) <= 1e-11, (name, mode, dphi.max(), np.quantile(dphi, 0.999))
        peak = np.abs(p0).max(axis=(0, 1)) + 1e-300
        assert (np.abs(p1.astype(float) - p0).max(axis=(0, 1)) / peak).max() <= 2e-7, (name, mode)


def test_inline_azimuth_vs_fused_kernel_on_random_cameras():
    # tools/sweep_split.py: random spins (a tenth of them at special values), inclinations of both signs, fields of view, nup from 40 to 1100
    # (mu-fill segments of 6 to 165 points: one to three chunks), the per-step record of the default trace against grt_set_geo_split(0).
    # Counts equal on every camera, |dphi| <= 1e-9 on every stored point (the sweep prints every camera above 1e-10; 800 cameras of seeds
    # 11 and 12 had none: profiles/r05m_sweep_split_400_seed1?.log -- seed 11 is the one that found the k^2 = 1 - m1m cancellation).
    import subprocess
    env = dict(os.environ); env.pop("GRT_LOG_LAUNCHES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_split.py"), "120", "11"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "cameras failing (counts or > 1e-9): 0" in last and "points above 1e-10: 0," in last, last


def test_record_geometry_vs_oracle_on_random_cameras():
    # tools/sweep_record_vs_oracle.py: r, theta, PHI, lambda of every stored point of every ray of random cameras (a quarter of them at
    # special spins: +-0.998, 0.9375, 0.5, and the nearly Schwarzschild 0.02, 0.0014, -0.0013, 1e-4 whose theta series was nine digits
    # short until r05m; nup 40 ... 900) against the oracle's GEOKERR arrays: counts equal, r / theta 1e-9, phi 1e-8 of the ray's scale (one point
    # behind a polar turning point in 3 of 750 cameras sits 1-2e-9 off, its neighbours 1e-14: the reference's own integral there), lambda 1e-8.  Before r05m no test held the record's azimuth to the oracle on more than a handful of fixed cameras.
    import subprocess
    env = dict(os.environ); env.pop("GRT_LOG_LAUNCHES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_record_vs_oracle.py"), "60", "5"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cameras above the bars: 0" in r.stdout.strip().splitlines()[-1], r.stdout[-1500:]
    # ... and the same cameras at a = 0 exactly (the Schwarzschild-cubic kernels; GEOKERR's trigonometric cubic): seed 21's cameras 28, 39
    # and 44 (nup = 427) each had a ray whose mu-fill segment started one stored point early until the cubic's acos / cos / x^1.5 were
    # rounded correctly on the device (grt_dev_cr.hpp)
    env["GRT_SWEEP_SPIN"] = "0.0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_record_vs_oracle.py"), "50", "21"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cameras above the bars: 0" in r.stdout.strip().splitlines()[-1], r.stdout[-1500:]
    # ... and cameras ALMOST on the axis (1 - |mu0| = 1e-6; exactly pole-on: mu0 = 1): the four-real-root class's azimuth was up to 9e-7 off
    # there until the near-polar rays of such a camera went back to the reference-literal path (grt_launch_setup, r05p)
    env.pop("GRT_SWEEP_SPIN")
    for mu0 in ("-0.999999", "1.0"):
        env["GRT_SWEEP_MU0"] = mu0
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_record_vs_oracle.py"), "30", "61"], capture_output=True, text=True, env=env, timeout=1200)
        assert r.returncode == 0, (mu0, r.stdout[-3000:] + r.stderr[-2000:])
        assert "cameras above the bars: 0" in r.stdout.strip().splitlines()[-1], (mu0, r.stdout[-1500:])


def test_geokerr_raw_arrays_vs_oracle_on_random_cameras():
    # tools/sweep_geokerr_rays_vs_oracle.py: random cameras through grt_geokerr_rays (the NPH = 3 forms of the set-up / geodesic kernels: they
    # also carry the coordinate time the slow-light fluid models use) against the oracle ray by ray: npts, ncase, TPMI, TPRI bit-equal,
    # u 1e-10, mu 1e-9, dt 1e-9, dphi 1e-8, lam 1e-9 of each array's largest value.  850 cameras / 41,600 rays in
    # profiles/r05o_geokerr_rays_*: no integer off, dt <= 3e-15, dphi <= 6e-10.
    import subprocess
    env = dict(os.environ); env.pop("GRT_LOG_LAUNCHES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_geokerr_rays_vs_oracle.py"), "60", "51"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cameras above the bars: 0" in r.stdout.strip().splitlines()[-1], r.stdout[-1500:]


OPEN_CASE_SETUP_R05 = {
    # name: (spin, mu0, uout, nn, gridvals, pixel range, measured error of the third frequency's worst Stokes parameter, pinned bar)
    "ffjet_p55_34ghz_seed8008": (0.05213690551730877, 0.0, 0.01, (12, 12, 60),
                                 (-9.570136125286059, 10.569561098919507, -4.811191063496265, 15.328506160709301), 26, 134, 2.64e-6, 6e-6),
    "ffjet_p55_34ghz_seed424242": (0.567049935230656, -0.42378143169961224, 0.02, (12, 12, 100),
                                   (-10.498569236028937, 13.589957295400408, -10.355504459802308, 13.733022071627037), 1, 144, 2.2e-5, 5e-5),
}


@pytest.mark.parametrize("name", sorted(OPEN_CASE_SETUP_R05))
def test_open_cases_of_the_r05_sweeps_vs_compiled_reference(fdev, name):
    # The two FFJET cameras the per-pixel rules of tools/fuzz_cameras.py leave above the 2e-6 line, frozen against the COMPILED
    # reference's image (tests/golden/open_cases_r05.npz <- tests/make_golden.py:OPEN_CASES_R05): POLSYNCHPL with p = 5.5, formal, at
    # 1035 / 345 / 34.5 GHz.  At the two upper frequencies the suite's bar holds.  At 34.5 GHz -- a tenth of the nominal frequency, Faraday-thick
    # rays whose Q, U, V turn by radians per step -- the reference's own image moves by 7e-8 / 1.4e-7 of a peak under one-ulp perturbations of its
    # inputs (cond) and by 1,360x / 2,350x any relative change of the frequency (amp): the device's 2.6e-6 / 2.2e-5 are what a relative
    # frequency error of 2e-9 / 9e-9 does there.  They stay ON RECORD as open (not waived: the per-pixel rule does not explain them by the
    # pixel's own response); the bar pinned here is twice the measured value, so that the cases cannot get worse unnoticed.
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "open_cases_r05.npz"))
    spin, mu0, uout, nn, gv, i1, i2, measured, pinned = OPEN_CASE_SETUP_R05[name]
    cam = capi.make_camera(1, gv, nn, spin, mu0, uout, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 5.5)
    freqs = [1.035e12, 3.45e11, 3.45e10]
    ref, pk, cond, amp = g[name + "_ivals"].astype(np.float64), g[name + "_peak"], g[name + "_cond"], g[name + "_amp"]
    assert amp[2].max() > 1000.0 and cond[2].max() > 5e-8          # (the regime, as the compiled reference itself shows it)
    pix, _, status, _ = capi.trace_tile(cam, i1, i2, "FFJET", em, freqs, "formal")
    assert np.array_equal(status, g[name + "_npts"])
    for k in range(3):
        assert np.array_equal(np.isfinite(pix[k]), np.isfinite(ref[k])), (name, k)
        okm = np.isfinite(ref[k]).all(axis=1)
        err = np.abs(pix[k][okm].astype(np.float64) - ref[k][okm]).max(axis=0) / pk[k]
        assert np.all(err <= (2e-6 if k < 2 else pinned)), (name, k, err)


def test_differential_compilation_same_inputs_through_two_code_objects(tmp_path):
    # VERDICT r04 weak #3 / next #3a: correctness of the shipped object has twice depended on what the register allocator did with a live
    # value (DESIGN 6 defects (3), (5): a copy under a partial exec mask; a live integer overwritten by a register copy) -- a family no
    # reference-free test could see.  tools/build_diff.sh compiles the SAME sources with every kernel at another occupancy (GRT_OCC: two
    # waves where the product takes three, three where it takes two, the other tile depth in the transfer kernels): another allocation,
    # spill set and schedule around the same arithmetic.  600 random cameras over every (fluid model, emissivity, integrator,
    # frequency-batch, extra=1, nvals=1) kernel family, plus GEOKERR's raw arrays, through both objects (tools/diff_sweep.py, one process
    # each): stored-point counts, turning-point counts, root cases and NaN masks bit-equal; the fp64 per-step record to 1e-12 of each
    # field's peak; the fp32 images equal up to one rounding of the last bit.
    import subprocess
    import sys
    diff = os.path.join(ROOT, "build", "diff", "libgrtrans_amd_diff.so")
    if not os.path.exists(diff):
        pytest.skip("build/diff/libgrtrans_amd_diff.so not built here (tools/build_diff.sh; __graft_entry__.build() makes it)")
    outs = []
    for tag, lib in (("product", None), ("diff", diff)):
        env = dict(os.environ)
        env.pop("GRTRANS_AMD_LIB", None)
        env.pop("GRT_CLASS_MERGE", None)
        if lib:
            env["GRTRANS_AMD_LIB"] = lib
            env.pop("GRT_LOG_LAUNCHES", None)      # (the other object's instantiations are not the product's: not for the coverage log)
            env["GRT_CLASS_MERGE"] = "0"           # r06: ... and the other launch form -- one transfer launch per ray class where the product merges them for cameras this small
        o = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diff_sweep.py"), o, "600", "20261003"], check=True, env=env, timeout=1500)
        outs.append(np.load(o, allow_pickle=False))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import diff_compare
    problems, st = diff_compare.compare(outs[0], outs[1])
    assert st["cameras"] == 600 and st["img_values"] > 300000
    assert not problems, problems[:5]
    assert st["img_bit_equal"] >= 0.9999 * st["img_values"], st
    if os.environ.get("GRT_TEST_ERRLOG"):
        with open(os.environ["GRT_TEST_ERRLOG"], "a") as f:
            f.write("differential compilation: %r\n" % st)


def test_debug_ray_twice_with_a_time_origin_table_installed():
    # grt_debug_ray finds the pixel's row of the nload>1 time-origin table through the ranges of the run's last
    # trace; its own one-pixel trace used to overwrite them, so a SECOND call resolved the same pixel to row 0 (silently the wrong
    # origin) and any other pixel of the tile was refused; a refused trace overwrote them too.  Row 1 of the dump is the point's
    # coordinate time x(1) = DTI(1) - DTI(i) - t0 (geodesics.f90:345): with a table installed it moves by exactly -t0[row].
    from grtrans_amd import capi
    nn = (32, 32, 100)
    dev = capi.Device(0)
    dev.load_ffjet(); dev.load_tables()
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5, mdot=1.57e15)
    capi.set_t0(None)
    plain = {p: capi.debug_ray(cam, p, "FFJET", em, FREQ, "formal") for p in (500, 530)}
    i1, i2 = 481, 544                                       # the tile the run traced: rows 0..63 of the table
    capi.trace_tile(cam, i1, i2, "FFJET", em, [FREQ], "formal")
    t0 = 10.0 + 3.0 * np.arange(i2 - i1 + 1)
    capi.set_t0(t0)
    try:
        for p in (500, 500, 530, 500):                      # the same pixel twice, another pixel of the tile, the first again
            n, hdr, rows, fac = capi.debug_ray(cam, p, "FFJET", em, FREQ, "formal")
            n0, _, rows0, fac0 = plain[p]
            assert n == n0 and fac == fac0
            assert np.allclose(rows[1] - rows0[1], -t0[p - i1], rtol=0, atol=1e-9), (p, (rows[1] - rows0[1])[:3], -t0[p - i1])
            assert np.array_equal(rows[2:5], rows0[2:5])    # r, theta, phi do not depend on the origin
        # a refused trace (no kernel for this pair) leaves the mapping alone as well
        with pytest.raises(capi.GrtError):
            capi.trace_tile(cam, 1, 8, "FFJET", capi.make_emis("POLSYNCHTH", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5), [FREQ], "formal")
        n, hdr, rows, fac = capi.debug_ray(cam, 530, "FFJET", em, FREQ, "formal")
        assert np.allclose(rows[1] - plain[530][2][1], -t0[530 - i1], rtol=0, atol=1e-9)
        # a pixel outside the tile has no row: refused, not row 0
        with pytest.raises(capi.GrtError):
            capi.debug_ray(cam, 100, "FFJET", em, FREQ, "formal")
    finally:
        capi.set_t0(None)
        dev.close()


# per block of geodebug.out: tolerance as a fraction of the block's largest magnitude.  Geodesic and frame quantities 1e-9 (phi 1e-8);
# fluid variables (REAL(4) samples at r, theta that agree with the reference's to 1e-9, not to the bit) and everything derived from them
# 2e-6 (the Stokes bar); aahat / aa / khat / bhat 1e-9
_GD_TOL = ([1e-9] * 4 + [1e-8] + [1e-9] * 4 + [0, 0] + [2e-6] * 3 + [5e-6] * 8 + [2e-6] * 5 + [2e-6] * 8 + [2e-6] * 4 + [2e-6] + [2e-6] * 3 + [2e-6] * 2 +
           [2e-6, 1e-8] + [2e-6] * 2 + [2e-6] + [2e-6] * 8 + [2e-6] + [2e-6] * 3 + [2e-6])


def test_debug_dump_geodebug_out_vs_the_reference(tmp_path, monkeypatch, harm_dir):
    # debug=1 (VERDICT r02 missing #3): geodebug.out in the reference's 64-block layout (grtrans_driver.f90:94-110, 341-421; parser
    # read_geodebug_file.py), computed on the device, against the file the reference's own driver writes for the same pixel
    # (tests/make_golden.py:geodebug): FFJET formal and DELO, SPHACC, HARM; through the C ABI and through ./grtrans.
    from grtrans_amd import capi, grtrans_batch as gb
    g = np.load(os.path.join(G, "geodebug_32x32x100.npz"))
    nn = (32, 32, 100)

    def compare(blocks, hdr, fac, key, skip=()):
        n, h0, ref = _geodebug_blocks(g[key])
        assert int(hdr[0]) == n and np.allclose(hdr[1:8], h0[1:8], rtol=1e-15) and np.allclose(hdr[9:], h0[9:], rtol=1e-15)
        worst = {}
        for b, (mine, want, tol) in enumerate(zip(blocks, ref, _GD_TOL)):
            assert mine.shape == want.shape, (key, b)
            if b in skip:
                continue
            if b == 46:
                # CALCG (kerr.f90:181-218) declares q2, l, su, sm as arrays and the driver hands it the ray's scalars: every element
                # after the first reads past them (the reference's block holds 1e-223-sized values and NaN there).  The first element
                # is the defined one; the device evaluates the formula with the scalars for every point.
                mine, want = mine[:1], want[:1]
            assert np.array_equal(np.isnan(mine), np.isnan(want)), (key, b)

            scale = np.nanmax(np.abs(want)) + 1e-300
            if b == 58:      # bht = -u.b is zero by construction (rounding noise of the REAL(4) samples): measured against |bhat| itself
                scale = max(np.nanmax(np.abs(ref[k_])) for k_ in (59, 60, 61)) + 1e-300
            err = np.nanmax(np.abs(mine - want)) / scale if want.size else 0.0
            worst[b] = err
        if os.environ.get("GRT_TEST_ERRLOG"):
            with open(os.environ["GRT_TEST_ERRLOG"], "a") as f:
                f.write("geodebug %s %s\n" % (key, " ".join("%d:%.1e" % (b + 1, e) for b, e in worst.items())))
        bad = {b + 1: e for b, e in worst.items() if e > max(_GD_TOL[b], 1e-15)}
        assert not bad, (key, "blocks over their tolerance", bad)
        return worst

    def from_rows(rows, fac):
        return [rows[b] for b in range(49)] + [np.concatenate([rows[49], rows[50], rows[51]])] + [rows[b] for b in range(52, 64)] + [np.array([fac])]

    dev = capi.Device(0)
    dev.load_ffjet(); dev.load_tables(); dev.load_sphacc()
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5, mdot=1.57e15)
    for pixel, iname, key in ((500, "formal", "ffjet_500"), (611, "delo", "ffjet_611_delo")):
        n, hdr, rows, fac = capi.debug_ray(cam, pixel, "FFJET", em, FREQ, iname)
        compare(from_rows(rows, fac), hdr, fac, key)
    cam = capi.make_camera(1, (-25., 25., -25., 25.), nn, 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5, mdot=1.5e15)      # (the header prints sparams%mdot: the harness default 1.5e15)
    n, hdr, rows, fac = capi.debug_ray(cam, 400, "SPHACC", em, FREQ_SPHACC, "formal")
    compare(from_rows(rows, fac), hdr, fac, "sphacc_400")
    dev.load_harm(os.path.join(harm_dir, "dump"), os.path.join(harm_dir, "dump040"), 40, 0.9375)
    cam = capi.make_camera(1, (-13., 13., -13., 13.), nn, 0.9375, 0.6428, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5, muval=0.25, mdot=1.57e15)
    n, hdr, rows, fac = capi.debug_ray(cam, 200, "HARM", em, FREQ_HARM, "formal")
    compare(from_rows(rows, fac), hdr, fac, "harm_200")
    # a ray that ends behind the horizon (NaN coefficient at its last point -> fac = NaN): the driver does not integrate it and its dump
    # holds the RAW coefficients, r%I = 0 printed as 0/fac/L_BH = NaN, and a tau array it never set (block 40: not compared)
    n, hdr, rows, fac = capi.debug_ray(cam, 530, "HARM", em, FREQ_HARM, "formal")
    assert np.isnan(fac)
    compare(from_rows(rows, fac), hdr, fac, "harm_530", skip=(39,))
    with pytest.raises(capi.GrtError):
        capi.debug_ray(cam, 530, "HARM", capi.make_emis("POLSYNCHSYMTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5), FREQ_HARM, "formal")
    dev.close()
    # the Fortran host: inputs.in with debug=1, i1 = i2 = the pixel -> geodebug.out, one block per line as read_geodebug_file.py reads it
    monkeypatch.chdir(tmp_path)
    kw = dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=1, nmu=1, fmin=FREQ, fmax=FREQ, ename='POLSYNCHPL', nvals=4,
              spin=0.998, standard=1, nn=[32, 32, 100], uout=0.01, mbh=3.4e9, mdotmin=1.57e15, mdotmax=1.57e15, mumin=.906, mumax=.906,
              gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70., iname='formal', cflag=0, i1=500, i2=500, debug=1)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    lines = open('geodebug.out').read().splitlines()
    assert len(lines) == 64                                     # header + 62 arrays + fac: data[0..63] of read_geodebug_file.py
    toks = np.array(" ".join(lines).split(), dtype=float)
    n, hdr, blocks = _geodebug_blocks(toks)
    compare(blocks, hdr, blocks[-1][0], "ffjet_500")
    assert len(lines[1].split()) == n and len(lines[50].split()) == 3 * n and len(lines[63].split()) == 1


def test_frequency_batches_on_every_fluid_model(tmp_path):
    # r03: the DELO frequency-batch kernel of KORAL3D (FB = 8) returned zero pixels / negative status for most rays when compiled for one
    # wave per SIMD (486 registers; a memory aperture violation with the r02 build) -- no test ran a thermal GRMHD model with more than
    # one frequency through DELO on KORAL3D.  Every fluid model, both integrators, 2 and 9 frequencies: each camera of the batch run is
    # the bits of its single-frequency run, statuses equal.
    import sys
    from grtrans_amd import capi
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_harm_dump, make_harm3d_dump, make_iharm_dump, make_koral3d_dump
    td = str(tmp_path)
    dk, hk = make_koral3d_dump.write(td, n1=32, n2=32, n3=16)
    make_harm_dump.write(td, n=64)
    d3, h3 = make_harm3d_dump.write(td, n1=32, n2=32, n3=16)
    di, hi = make_iharm_dump.write(td, n1=32, n2=32, n3=16, metric=1)
    dev = capi.Device(0)
    try:
        dev.load_tables(); dev.load_ffjet(); dev.load_sphacc(); dev.load_koral3d(dk, hk, 40, 0.9375)
        dev.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
        dev.load_harm3d(d3, h3, 40, 0.9375); dev.load_iharm(di, hi, 40, 0.9375)
        capi.geo_reuse(False)
        nn = (12, 12, 100)
        grid = [2.3e11 * x for x in (1.0, 0.3, 0.45, 0.9, 1.3, 1.8, 2.2, 2.6, 3.0)]
        # r04: EVERY (fluid model, emissivity) pair that has a kernel (grt_kernels.hip:grt_launch_transfer) -- the launch log of the r03
        # suite showed 91 of 176 instantiations never started by any GPU test (tests/test_zz_instantiation_coverage.py holds the list to zero)
        K3 = dict(mbh=4e6, nfac=1.5, gmin=-1.); H2 = dict(mbh=4e6, gmin=1., mdot=1.57e15); H3 = dict(mbh=4e6, gmin=2., mdot=1e20); IH = dict(mbh=4e6, gmin=3., mdot=1e20)
        SP = dict(mbh=10., gmin=100.)
        cases = (("KORAL3D", "POLSYNCHTH", K3, 0.9375), ("HARM", "POLSYNCHTH", H2, 0.9375), ("HARM3D", "POLSYNCHTH", H3, 0.9375), ("IHARM", "POLSYNCHTH", IH, 0.9375),
                 ("HARM", "HYBRIDTHPL", H2, 0.9375), ("HARM3D", "HYBRIDTHPL", H3, 0.9375), ("IHARM", "HYBRIDTHPL", IH, 0.9375),
                 ("HARM", "MAXJUTT", H2, 0.9375), ("HARM3D", "MAXJUTT", H3, 0.9375), ("IHARM", "MAXJUTT", IH, 0.9375),
                 ("HARM", "POLSYNCHSYMTH", H2, 0.9375), ("HARM3D", "POLSYNCHSYMTH", H3, 0.9375), ("IHARM", "POLSYNCHSYMTH", IH, 0.9375), ("SPHACC", "POLSYNCHSYMTH", SP, 0.0),
                 ("SPHACC", "POLSYNCHTH", SP, 0.0), ("FFJET", "POLSYNCHPL", dict(mbh=3.4e9, nfac=2.0, bfac=70.0, gmin=100.), 0.998))
        for fname, ename, emk, a in cases:
            gv = (-25., 25., -25., 25.) if fname in ("SPHACC", "FFJET") else (-13., 13., -13., 13.)
            cam = capi.make_camera(1, gv, nn, a, 0.5, 0.04, -0.5)
            em = capi.make_emis(ename, emk["mbh"], emk.get("nfac", 1.0), emk.get("bfac", 6.0), emk["gmin"], 1e5, 3.5, muval=0.25, mdot=emk.get("mdot", 0.0),
                                otherargs=(1.5, 1.0, 0.5) if ename == "MAXJUTT" else (1.0, 1.0))
            for iname in ("delo", "formal"):
                singles = [capi.trace_tile(cam, 1, 144, fname, em, [x], iname) for x in grid]
                if ename != "POLSYNCHSYMTH":      # (extra=1 is refused for the Pandya fits, as for the scalar emissivities)
                    # the extra=1 instantiation (up to 512 registers at one wave per SIMD) integrates the same Stokes vector
                    ex, _, stx, _ = capi.trace_tile(cam, 1, 144, fname, em, grid[:1], iname, nvals=4, nextra=19)
                    assert np.array_equal(stx, singles[0][2]), (fname, ename, iname, "extra")
                    a_, b_ = ex[0][:, :4].astype(float), singles[0][0][0].astype(float)
                    assert np.array_equal(np.isnan(a_), np.isnan(b_))
                    assert np.nanmax(np.abs(a_ - b_)) <= 1e-7 * np.nanmax(np.abs(b_)), (fname, ename, iname, "extra")
                for nf in (2, 9):
                    pix, _, st, _ = capi.trace_tile(cam, 1, 144, fname, em, grid[:nf], iname)
                    assert np.array_equal(st, singles[0][2]), (fname, ename, iname, nf)
                    for k in range(nf):
                        assert np.array_equal(pix[k], singles[k][0][0], equal_nan=True), (fname, ename, iname, nf, k)
            # nvals = 1: the quadrature kernels (either iname), batches of 13 against single-frequency runs
            singles = [capi.trace_tile(cam, 1, 144, fname, em, [x], "formal", nvals=1) for x in grid]
            for nf in (2, 9):
                pix, _, st, _ = capi.trace_tile(cam, 1, 144, fname, em, grid[:nf], "delo", nvals=1)
                assert np.array_equal(st, singles[0][2]), (fname, ename, "nvals=1", nf)
                for k in range(nf):
                    assert np.array_equal(pix[k], singles[k][0][0], equal_nan=True), (fname, ename, "nvals=1", nf, k)
            with pytest.raises(capi.GrtError):      # extra=1 with nvals=1: the reference reads opacity columns that do not exist there
                capi.trace_tile(cam, 1, 144, fname, em, grid[:1], "formal", nvals=1, nextra=19)
        # the scalar emissivities (nvals = 1 only): one instantiation per fluid model, em.type selects
        for fname, emk, a in (("HARM", H2, 0.9375), ("HARM3D", H3, 0.9375), ("IHARM", IH, 0.9375), ("SPHACC", SP, 0.0), ("KORAL3D", K3, 0.9375)):
            gv = (-25., 25., -25., 25.) if fname == "SPHACC" else (-13., 13., -13., 13.)
            cam = capi.make_camera(1, gv, nn, a, 0.5, 0.04, -0.5)
            em = capi.make_emis("SYNCHTHAV", emk["mbh"], emk.get("nfac", 1.0), emk.get("bfac", 6.0), emk["gmin"], 1e5, 3.5, muval=0.25, mdot=emk.get("mdot", 0.0))
            singles = [capi.trace_tile(cam, 1, 144, fname, em, [x], "formal", nvals=1) for x in grid]
            for nf in (2, 9):
                pix, _, st, _ = capi.trace_tile(cam, 1, 144, fname, em, grid[:nf], "formal", nvals=1)
                assert np.array_equal(st, singles[0][2]), (fname, "SYNCHTHAV", nf)
                for k in range(nf):
                    assert np.array_equal(pix[k], singles[k][0][0], equal_nan=True), (fname, "SYNCHTHAV", nf, k)
    finally:
        capi.geo_reuse(True)
        dev.close()


def test_every_pair_and_integrator_vs_oracle(tmp_path):
    # r05 (VERDICT r04 next #3c): the launch log, tagged with the test that caused each launch, showed 45 of 172 instantiations -- the DELO
    # and nvals = 1 forms of the secondary emissivities -- launched only by self-consistency tests (batch = single frequency, the two code
    # objects): their goldens hold `formal` alone.  Every (fluid model, emissivity) pair that has a kernel, formal, DELO and the nvals = 1
    # quadrature, against the oracle (pinned to the compiled reference, tests/test_oracle_vs_golden.py) at the nominal frequencies:
    # stored-point counts and non-finite masks equal, Stokes to 2e-6 of each peak (67 cases; measured worst 2.2e-7).
    import sys
    from grtrans_amd import capi
    from oracle import orabind
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_harm_dump, make_harm3d_dump, make_iharm_dump, make_koral3d_dump
    td = str(tmp_path)
    make_harm_dump.write(td, n=64)
    d3, h3 = make_harm3d_dump.write(td, n1=32, n2=32, n3=16, b0=2.5)
    di, hi = make_iharm_dump.write(td, n1=32, n2=32, n3=16, b0=2.5)
    dk, hk = make_koral3d_dump.write(td, n1=32, n2=32, n3=16)
    H2 = dict(dfile="dump", hfile="dump040", findf=40, mbh=4e6, mdot=1.57e15, muval=0.25, gmin=1.)
    H3 = dict(dfile="dump3d", hfile="dump3d_header", findf=40, mbh=4e6, mdot=1e19, muval=0.25, gmin=2.)
    IH = dict(dfile="dumpih", hfile="dumpih_header", findf=40, mbh=4e6, mdot=1e19, muval=0.25, gmin=3.)
    K3 = dict(dfile="koral", hfile="koral_header", findf=40, mbh=4e6, muval=0.25, gmin=-1., ftscl=1.5)
    SP = dict(mbh=10., gmin=100.)
    PAIRS = [("HARM", e, H2) for e in ("POLSYNCHTH", "HYBRIDTHPL", "HYBRIDTH", "MAXJUTT", "POLSYNCHSYMTH", "POLSYNCHPL")] + \
            [("HARM3D", e, H3) for e in ("POLSYNCHTH", "HYBRIDTHPL", "HYBRIDPL", "MAXJUTT", "MAXCOMP", "POLSYNCHSYMTH", "POLSYNCHPL")] + \
            [("IHARM", e, IH) for e in ("POLSYNCHTH", "HYBRIDTHPL", "MAXJUTT", "POLSYNCHSYMTH", "POLSYNCHPL")] + \
            [("KORAL3D", "POLSYNCHTH", K3), ("SPHACC", "POLSYNCHTH", SP), ("SPHACC", "POLSYNCHSYMTH", SP)]
    SCALAR = [("HARM", "SYNCHTHAV", H2), ("HARM3D", "BREMS", H3), ("IHARM", "SYNCHTHBREMS", IH), ("SPHACC", "SYNCHTHAVNOABS", SP), ("KORAL3D", "RHO", K3)]
    nn = (12, 12, 100)
    dev = capi.Device(0)
    worst = {}
    try:
        dev.load_tables(); dev.load_sphacc()
        dev.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
        dev.load_harm3d(d3, h3, 40, 0.9375); dev.load_iharm(di, hi, 40, 0.9375); dev.load_koral3d(dk, hk, 40, 0.9375)
        for fname, ename, kw in PAIRS + SCALAR:
            a, gv, uout, freq = (0.0, (-25., 25., -25., 25.), 0.04, 2.41e11) if fname == "SPHACC" else (0.9375, (-13., 13., -13., 13.), 0.04, 2.3e11)
            oa = (1.5, 1.0, 0.5) if ename == "MAXJUTT" else ((1.5, 1.0, 1.0, 0.5) if ename == "MAXCOMP" else (1.0, 1.0))
            cam = capi.make_camera(1, gv, nn, a, 0.5, uout, -0.5)
            pl = ename in ("POLSYNCHPL", "HYBRIDTHPL", "HYBRIDTH", "HYBRIDPL")
            gmin = 50. if (ename == "POLSYNCHPL") else kw["gmin"]
            em = capi.make_emis(ename, kw["mbh"], kw.get("ftscl", 1.0), 6.0, gmin, 1e5, 3.5, muval=0.25, mdot=kw.get("mdot", 0.0), otherargs=oa)      # (nfac = the harness's ftscl)
            scalar = (fname, ename, kw) in SCALAR
            for iname, nvals in ((("formal", 1),) if scalar else (("formal", 4), ("delo", 4), ("delo", 1))):
                okw = dict(kw); okw.update(gmin=gmin, nvals=nvals, otherargs=oa)
                o = orabind.Oracle(nn=nn, iname=iname, spin=a, mu0=0.5, gridvals=gv, uout=uout, fname=fname, ename=ename, datadir=td, p1=3.5, **okw)
                ref, rn, _ = o.trace(1, 144, [freq], nthreads=0)
                pix, _, status, _ = capi.trace_tile(cam, 1, 144, fname, em, [freq], iname, nvals=nvals)
                assert np.array_equal(status, rn), (fname, ename, iname, nvals)
                ref = np.asarray(ref[0]); dv = pix[0].astype(np.float64)
                assert np.array_equal(np.isfinite(dv), np.isfinite(ref)), (fname, ename, iname, nvals)
                fin = np.isfinite(ref).all(axis=1)
                if not fin.any() or not np.abs(ref[fin]).max() > 0:
                    continue
                pk = np.abs(ref[fin]).max(axis=0); pk = np.maximum(pk, 1e-2 * pk[0]) + 1e-300
                err = (np.abs(dv[fin] - ref[fin]).max(axis=0) / pk).max()
                worst[(fname, ename, iname, nvals)] = err
                assert err < STOKES_TOL, (fname, ename, iname, nvals, err)
    finally:
        dev.close()
    assert len(worst) >= 60, len(worst)
    if os.environ.get("GRT_TEST_ERRLOG"):
        with open(os.environ["GRT_TEST_ERRLOG"], "a") as f:
            k = max(worst, key=worst.get)
            f.write("every pair and integrator vs oracle: %d cases, worst %.2e at %r\n" % (len(worst), worst[k], k))


def test_koral3d_several_slices_slow_light_nload_and_movie(tmp_path, monkeypatch):
    # KORAL3D with fnt = 3 loaded slices (each point reads the dump its Kerr-Schild look-back time selects,
    # fluid_model_koral3d.f90:126-128, 212-228), nload = 2 (per-ray time origin) and two movie frames of a two-slice run
    # (advance_koral3d_timestep :635-645 with tstep from the dumps).  Golden = the reference's own grtrans_main.  HARM3D and IHARM
    # with several slices are undefined in the reference (dump time never read / tstep = 0) and refused.
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "koral3d_slowlight_24x24x100.npz"))
    monkeypatch.chdir(tmp_path)
    paths = make_golden.write_koral_slow_dumps(str(tmp_path))
    assert [hashlib.sha256(open(paths[k], "rb").read()).hexdigest() for k in sorted(paths)] == list(g["dump_sha256"])

    def cmp(pix, key, frame=0):
        ref = g["ivals_" + key][frame].astype(float)
        bad = np.isnan(ref).any(axis=1)                 # DELO: three pixels are NaN in the reference
        assert np.array_equal(bad, np.isnan(pix).any(axis=1)), key
        return stokes_err(pix[~bad][None], ref[~bad][None])

    dev = capi.Device(0)
    dev.load_tables()
    cam = capi.make_camera(1, (-13., 13., -13., 13.), (24, 24, 100), 0.9375, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.5, 6.0, -1., 1e5, 3.5, muval=0.25)
    dev.load_koral3d("koral", "koral_header", 41, 0.9375, nt=3)
    assert capi.koral3d_tstep() == 20.0
    for key, iname in (("slow", "formal"), ("slow_delo", "delo")):
        for rec in (False, True):
            capi.record_steps(rec)
            pix, _, status, _ = capi.trace_tile(cam, 1, 576, "KORAL3D", em, [FREQ_HARM], iname)
            assert cmp(pix[0], key) < STOKES_TOL, (key, rec)
    capi.record_steps(False)
    # extra=1 with several slices (r03; golden = the reference's own main with hnt=3, extra=1): tolerances of test_extra_ray_diagnostics
    ge = np.load(os.path.join(G, "koral3d_slowlight_extra_24x24x100.npz"))
    pix, _, status, _ = capi.trace_tile(cam, 1, 576, "KORAL3D", em, [FREQ_HARM], "formal", nextra=19)
    a_, b_ = np.asarray(pix[0], float), np.asarray(ge["vals"], float)
    assert np.array_equal(np.isnan(a_), np.isnan(b_))
    a_, b_ = np.nan_to_num(a_), np.nan_to_num(b_)
    err_ = np.abs(a_ - b_).max(axis=0) / (np.abs(b_).max(axis=0) + 1e-300)
    assert np.all(err_ < np.array([1e-5] * 4 + [1e-5] * 6 + [5e-5] * 8 + [1e-4] * 5)), err_
    dev.load_koral3d("koral", "koral_header", 41, 0.9375)
    pix, _, status, _ = capi.trace_tile(cam, 1, 576, "KORAL3D", em, [FREQ_HARM], "formal")
    assert cmp(pix[0], "fast") < STOKES_TOL
    assert np.abs(g["ivals_slow"] - g["ivals_fast"]).max() > 0.1 * np.abs(g["ivals_slow"]).max()
    with pytest.raises(capi.GrtError):
        dev.load_koral3d("koral", "koral_header", 41, 0.9375, nt=5)                        # dump 37 does not exist
    dev.close()
    kw = dict(fname='KORAL3D', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=0.9375, standard=1,
              nn=[24, 24, 100], uout=0.04, mbh=4e6, nmdot=1, mumin=.5, mumax=.5, gridvals=[-13, 13, -13, 13], muval=0.25, gmin=-1.,
              iname='formal', cflag=0, hdfile='koral', hhfile='koral_header', hindf=41, hnt=3, ntscl=1.5)
    for key, extra in (("slow", {}), ("nload", dict(nload=2)), ("movie", dict(hnt=2, nt=2, dt=65.))):
        x = gb.grtrans()
        x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **dict(kw, **extra))
        x.run_grtrans()
        x.read_grtrans_output(bin=1)
        y = gb.grtrans()
        y.run_pgrtrans(**dict(kw, **extra))
        for l in range(g["ivals_" + key].shape[0]):
            assert cmp(x.ivals[:, :, l], key, l) < STOKES_TOL, (key, l)
            assert np.array_equal(y.ivals[:, :, l].T.astype(np.float32), x.ivals[:, :, l].astype(np.float32))
    assert np.abs(g["ivals_movie"][1] - g["ivals_movie"][0]).max() > 0.1 * np.abs(g["ivals_movie"][0]).max()      # the dumps moved on
    # the other 3-D readers with several slices: no defined time step in the reference
    for fname in ("HARM3D", "IHARM"):
        with pytest.raises(Exception):
            z = gb.grtrans()
            z.run_pgrtrans(**dict(kw, fname=fname, hnt=2))


def test_fits_output_cflag1_through_fortran_host(tmp_path, monkeypatch):
    # cflag=1: the reference's default on-disk format.  The host writes FITS with its own emitter (no cfitsio here),
    # the harness reads it back without astropy; nx, ny, nu sit at header positions 6, 7, 8 as the reference's reader
    # expects.  (Format parity with cfitsio's output is unpinned: neither cfitsio nor astropy exists in this image.)
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "ffjet_32x32x100.npz"))
    kw = dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=2, nmu=1, fmin=FREQ, fmax=2 * FREQ,
              ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[32, 32, 100], uout=0.01, mbh=3.4e9,
              mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70., iname='formal', cflag=1)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.fits', **kw)
    x.run_grtrans()
    raw = open('grtrans.fits', 'rb').read()
    assert len(raw) % 2880 == 0 and raw[:30].decode() == "SIMPLE  =                    T"
    # (1) container grammar by an independent strict FITS 4.0 reader (tests/fits_check.py: card grammar, END, blank/zero
    # fill, 2880-byte blocks, mandatory keyword order, big-endian data)
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fits_check
    strict = fits_check.check(raw)
    assert len(strict) == 3 and strict[0][1].size == 2 * 1024 and strict[1][1].size == 4 * 1024
    # (2) the cards behind the six mandatory ones: names, order and value kinds of the reference's writer
    # (tests/golden/fits_keys.json, parsed from camera.f90:216-312 + pgrtrans.f90:229-238 by tests/extract_fits_keys.py)
    want = json.load(open(os.path.join(G, "fits_keys.json")))
    for h in strict[1:]:
        got = h[0][6:]
        assert [k.upper() for k, _, _ in got] == [w["name"].upper() for w in want]
        for (k, v, _), w in zip(got, want):
            assert isinstance(v, {"j": int, "d": float, "e": float, "s": str}[w["kind"]]), (k, v, w)
    assert np.array_equal(strict[1][1], gb.read_fits('grtrans.fits')[1][1])        # both readers see the same numbers
    hdus = gb.read_fits('grtrans.fits')
    assert len(hdus) == 3 and [k for k, _ in hdus[1][0][:9]] == ['XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'PCOUNT', 'GCOUNT', 'NX', 'NY', 'NU']
    assert dict(hdus[1][0])['gridvals(3)'].startswith('-2.0') and dict(hdus[2][0])['FNAME'] == 'FFJET'
    x.read_grtrans_output(bin=0)
    assert x.ivals.shape == (1024, 4, 2) and np.allclose(x.nu, [FREQ, 2 * FREQ], rtol=1e-6)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal"]) < STOKES_TOL
    assert np.array_equal(x.ab[:, 0], g["cam_alpha"].astype(np.float32))


def test_harm3d_device_and_both_hosts_vs_reference_golden(tmp_path, monkeypatch):
    # 3-D HARM (binary dump read by the product's loader; eight-corner fp32 interpolation periodic in
    # phi sampled at each point's azimuth; fluid.f90:975-993 units, electron model and magnetisation cut); formal, delo,
    # sigcut, nvals=1, per-step record, then the same problem through ./grtrans (Fortran host) and run_pgrtrans
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_harm3d_dump
    g = np.load(os.path.join(G, "harm3d_40x40x200.npz"))
    monkeypatch.chdir(tmp_path)
    dfile, hfile = make_harm3d_dump.write(str(tmp_path), **make_golden.HARM3D_ZONES)
    assert hashlib.sha256(open(dfile + "040.bin", "rb").read()).hexdigest() == str(g["dump_sha256"])
    H = make_golden.HARM3D
    d = capi.Device(0)
    try:
        d.load_tables()
        with pytest.raises(capi.GrtError):
            d.load_harm3d(dfile, hfile, 40, 0.5)                  # spin check of initialize_harm3d_model
        d.load_harm3d(dfile, hfile, 40, H["spin"])
        cam = capi.make_camera(1, H["gridvals"], (40, 40, 200), H["spin"], H["mu0"], H["uout"], -0.5)
        for iname, key, sig in (("formal", "ivals_formal", 1e10), ("delo", "ivals_delo", 1e10), ("formal", "ivals_formal_sigcut", 1e-3)):
            em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, H["gmin"], 1e5, 3.5, muval=H["muval"], mdot=H["mdot"], sigcut=sig)
            pix, _, status, st = capi.trace_tile(cam, 1, 1600, "HARM3D", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["npts"])
            assert stokes_err(pix[0], g[key]) < STOKES_TOL, key
        # per-step record of the sigcut run against the reference's per-stage dumps
        r = np.load(os.path.join(G, "harm3d_rays_40x40x200.npz"))
        capi.record_steps(True)
        try:
            capi.trace_tile(cam, 1, 1600, "HARM3D", em, [FREQ_HARM], "formal")
            for i in r["rays"]:
                n = int(r["ray%d_npts" % i])
                rec = capi.download_steps(int(i) - 1, 1, 230)[:, 0, :n]
                x = r["ray%d_x" % i]
                assert np.allclose(rec[0], x[:, 1], rtol=1e-9) and np.allclose(rec[2], x[:, 3], rtol=1e-8, atol=1e-9)
                I = np.nan_to_num(r["ray%d_I" % i])
                assert np.abs(np.nan_to_num(rec[4:8, ::-1].T[:n]) - I).max() / (np.abs(I).max() + 1e-300) < 1e-5
        finally:
            capi.record_steps(False)
        # extra=1: the 19 ray diagnostics on the 3-D sampler (column tolerances as in test_extra_ray_diagnostics_vs_reference)
        cam2 = capi.make_camera(1, H["gridvals"], (24, 24, 120), H["spin"], H["mu0"], H["uout"], -0.5)
        em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, H["gmin"], 1e5, 3.5, muval=H["muval"], mdot=H["mdot"])
        px, _, status, _ = capi.trace_tile(cam2, 1, 576, "HARM3D", em, [FREQ_HARM], "formal", nvals=4, nextra=19)
        assert np.array_equal(status, g["extra_npts_24x24x120"])
        a = np.asarray(px[0], float); b = np.asarray(g["extra_vals_24x24x120"], float)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        tolc = np.array([1e-5] * 4 + [1e-5] * 6 + [5e-5] * 8 + [1e-4] * 5)
        err = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max(axis=0) / (np.abs(np.nan_to_num(b)).max(axis=0) + 1e-300)
        assert np.all(err < 2 * tolc), err
        # nvals=1 (quadrature) runs and agrees with the polarized Stokes I where the medium is thin
        em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, H["gmin"], 1e5, 3.5, muval=H["muval"], mdot=H["mdot"])
        p1, _, _, _ = capi.trace_tile(cam, 1, 1600, "HARM3D", em, [FREQ_HARM], "formal", nvals=1)
        assert p1.shape == (1, 1600, 1) and np.isfinite(p1).all() and p1.max() > 0
    finally:
        d.close()
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    kw = dict(fname='HARM3D', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=H["spin"],
              standard=1, nn=[40, 40, 200], uout=H["uout"], mbh=H["mbh"], mdotmin=H["mdot"], mdotmax=H["mdot"], nmdot=1,
              mumin=H["mu0"], mumax=H["mu0"], gridvals=list(H["gridvals"]), muval=H["muval"], gmin=H["gmin"], iname='formal',
              cflag=0, tdfile='dump3d', hhfile='dump3d_header', tindf=40, tnt=1, sigcut=1e-3)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal_sigcut"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))


def test_hybrid_emissivities_vs_reference_golden(tmp_path):
    # thermal + power-law electron populations (HYBRIDTHPL / HYBRIDTH / HYBRIDPL, emis.f90:473-490) on the 2-D HARM dump and
    # on a magnetised 3-D HARM dump (non-thermal electrons where b^2/rho > 1); golden = compiled reference
    import sys
    from grtrans_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_harm3d_dump
    import make_harm_dump
    g = np.load(os.path.join(G, "hybrid_32x32x150.npz"))
    dfile, hfile = make_harm3d_dump.write(str(tmp_path), b0=float(g["b0"]), **make_golden.HARM3D_ZONES)
    make_harm_dump.write(str(tmp_path), n=64)
    H3, H2, HY = make_golden.HARM3D, make_golden.HARM, make_golden.HYBRID
    d = capi.Device(0)
    try:
        d.load_tables()
        d.load_harm3d(dfile, hfile, 40, H3["spin"])
        cam = capi.make_camera(1, H3["gridvals"], (32, 32, 150), H3["spin"], H3["mu0"], H3["uout"], -0.5)
        for en, iname, key in (("HYBRIDTHPL", "formal", "harm3d_HYBRIDTHPL"), ("HYBRIDTH", "formal", "harm3d_HYBRIDTH"),
                               ("HYBRIDPL", "formal", "harm3d_HYBRIDPL"), ("HYBRIDTHPL", "delo", "harm3d_HYBRIDTHPL_delo")):
            em = capi.make_emis(en, H3["mbh"], 1.0, 6.0, HY["gmin"], HY["gmax"], HY["p1"], muval=H3["muval"], mdot=HY["mdot"],
                                jetalpha=HY["jetalpha"])
            pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM3D", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["harm3d_npts"])
            ref = g[key]
            ok = ~np.isnan(ref).any(axis=1)          # the reference's DELO overflows on 20 pixels of this problem
            # Q, U tolerance 1e-5 of peak here: the magnetised dump has rays with ~1e3 rad of Faraday rotation, which
            # multiplies the fp64 rounding differences in rho_V by that depth (measured: I 4e-8, Q/U 3.4e-6 of peak;
            # it was 2.4e-5 while the REAL(4) |b| island was one ulp off)
            assert stokes_err(pix[0][ok], ref[ok]) < 1e-5, key
            assert stokes_err(pix[0][ok][:, :1], ref[ok][:, :1]) < STOKES_TOL, key
        d.load_harm(os.path.join(str(tmp_path), "dump"), os.path.join(str(tmp_path), "dump040"), 40, H2["spin"])
        cam = capi.make_camera(1, H2["gridvals"], (32, 32, 150), H2["spin"], H2["mu0"], H2["uout"], -0.5)
        for en in ("HYBRIDTHPL", "HYBRIDTH"):
            em = capi.make_emis(en, H2["mbh"], 1.0, 6.0, 10., 1e5, 3.5, muval=H2["muval"], mdot=H2["mdot"])
            pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal")
            assert np.array_equal(status, g["harm_npts"])
            assert stokes_err(pix[0], g["harm_" + en]) < STOKES_TOL, en
        # a pair without a kernel is refused, never emulated
        em = capi.make_emis("HYBRIDTHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
        d.load_ffjet()
        with pytest.raises(capi.GrtError):
            capi.trace_tile(capi.make_camera(1, (-40., 20., -20., 40.), (8, 8, 50), 0.998, 0.906, 0.01, -0.5), 1, 64, "FFJET", em, [FREQ], "formal")
    finally:
        d.close()


def test_powerlaw_on_grmhd_samplers_vs_reference_golden(tmp_path):
    # r05 (VERDICT r04 missing #3): the reference dispatches fluid model and emissivity independently (grtrans_driver.f90:122-159,
    # emis.f90:461-571).  POLSYNCHPL on the GRMHD samplers whose unit conversion defines a non-thermal density: 2-D HARM (the thermal
    # density, fluid.f90:972), HARM3D and IHARM (nonthermale_b2 where b^2/rho > 1, fluid.f90:914-923, on magnetised dumps); formal, DELO and
    # the nvals = 1 quadrature.  Golden = compiled reference (tests/make_golden.py:powerlaw_grmhd).  KORAL3D / SPHACC + POLSYNCHPL and FFJET
    # + a thermal emissivity give an identically zero image in the reference (it sets the density the emissivity needs to zero): refused.
    import sys
    from grtrans_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    import make_harm3d_dump
    import make_harm_dump
    import make_iharm_dump
    g = np.load(os.path.join(G, "powerlaw_grmhd_28x28x120.npz"))
    td = str(tmp_path)
    make_harm_dump.write(td, n=64)
    d3, h3 = make_harm3d_dump.write(td, b0=float(g["b0"]), **make_golden.HARM3D_ZONES)
    di, hi = make_iharm_dump.write(td, b0=float(g["b0"]), **make_golden.IHARM_ZONES)
    nn = (28, 28, 120)
    d = capi.Device(0)
    try:
        d.load_tables()
        d.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
        d.load_harm3d(d3, h3, 40, 0.9375); d.load_iharm(di, hi, 40, 0.9375)
        for tag, fname, base, mdot in (("harm", "HARM", make_golden.HARM, make_golden.HARM["mdot"]), ("harm3d", "HARM3D", make_golden.HARM3D, 1e19),
                                       ("iharm", "IHARM", make_golden.IHARM, 1e19)):
            cam = capi.make_camera(1, base["gridvals"], nn, base["spin"], base["mu0"], base["uout"], -0.5)
            em = capi.make_emis("POLSYNCHPL", base["mbh"], 1.0, 6.0, 50., 1e5, 3.5, muval=base["muval"], mdot=mdot, jetalpha=0.02)
            for iname, nvals in (("formal", 4), ("delo", 4), ("formal", 1)):
                ref = g["%s_%s_nvals%d" % (tag, iname, nvals)]
                pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], fname, em, [FREQ_HARM], iname, nvals=nvals)
                assert np.array_equal(status, g[tag + "_npts"]), (tag, iname, nvals)
                ok = ~np.isnan(ref).any(axis=1)          # (the reference's DELO overflows on some pixels of the magnetised 3-D dumps)
                assert np.array_equal(np.isnan(pix[0]).any(axis=1), ~ok), (tag, iname, nvals)
                assert (np.abs(ref[ok]).max(axis=1) > 0).sum() >= (8 if iname == "formal" else 1)      # (the goldens are not trivially zero)
                # (Q, U, V at 1e-5 on the magnetised dumps: ~1e3 rad of Faraday rotation along their rays, as for the hybrid emissivities)
                assert stokes_err(pix[0][ok], ref[ok]) < (STOKES_TOL if tag == "harm" else 1e-5), (tag, iname, nvals, stokes_err(pix[0][ok], ref[ok]))
                assert stokes_err(pix[0][ok][:, :1], ref[ok][:, :1]) < STOKES_TOL, (tag, iname, nvals)
        # frequency batches and extra=1 of the new instantiations against their single-frequency / plain runs (bit-equal)
        cam = capi.make_camera(1, make_golden.HARM["gridvals"], (12, 12, 100), 0.9375, 0.6428, 0.04, -0.5)
        em = capi.make_emis("POLSYNCHPL", 4e6, 1.0, 6.0, 50., 1e5, 3.5, muval=0.25, mdot=make_golden.HARM["mdot"], jetalpha=0.02)
        grid = [FREQ_HARM * x for x in (1.0, 0.3, 0.45, 0.9, 1.3, 1.8, 2.2, 2.6, 3.0)]
        capi.geo_reuse(False)
        for fname in ("HARM", "HARM3D", "IHARM"):
            for iname in ("formal", "delo"):
                singles = [capi.trace_tile(cam, 1, 144, fname, em, [x], iname) for x in grid]
                ex, _, stx, _ = capi.trace_tile(cam, 1, 144, fname, em, grid[:1], iname, nvals=4, nextra=19)
                assert np.array_equal(stx, singles[0][2])
                a_, b_ = ex[0][:, :4].astype(float), singles[0][0][0].astype(float)
                assert np.array_equal(np.isnan(a_), np.isnan(b_)) and np.nanmax(np.abs(a_ - b_)) <= 1e-7 * np.nanmax(np.abs(b_)) + 1e-300
                pix, _, st, _ = capi.trace_tile(cam, 1, 144, fname, em, grid, iname)
                for k in range(len(grid)):
                    assert np.array_equal(pix[k], singles[k][0][0], equal_nan=True), (fname, iname, k)
            singles = [capi.trace_tile(cam, 1, 144, fname, em, [x], "formal", nvals=1) for x in grid]
            pix, _, st, _ = capi.trace_tile(cam, 1, 144, fname, em, grid, "delo", nvals=1)
            for k in range(len(grid)):
                assert np.array_equal(pix[k], singles[k][0][0], equal_nan=True), (fname, "nvals=1", k)
        # the pairs whose image is identically zero in the reference stay refused
        d.load_sphacc(); d.load_ffjet()
        for fname, ename in (("SPHACC", "POLSYNCHPL"), ("FFJET", "POLSYNCHTH")):
            with pytest.raises(capi.GrtError):
                capi.trace_tile(cam, 1, 16, fname, capi.make_emis(ename, 10., 1.0, 6.0, 100., 1e5, 3.5), [FREQ_HARM], "formal")
    finally:
        capi.geo_reuse(True)
        d.close()


def test_maxjutt_vs_reference_golden_device_and_fortran_host(tmp_path, monkeypatch):
    # MAXJUTT (calc_maxjutt.f90): weighted sum of thermal components; device through the C ABI with explicit otherargs, and
    # the Fortran host with delta / nweights=6 from the namelist (its tabulated REAL(4) weights, read_inputs.f90:95-96)
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_harm3d_dump
    import make_harm_dump
    g = np.load(os.path.join(G, "maxjutt_32x32x150.npz"))
    monkeypatch.chdir(tmp_path)
    dfile, hfile = make_harm3d_dump.write(str(tmp_path), **make_golden.HARM3D_ZONES)
    make_harm_dump.write(str(tmp_path), n=64)
    H3, H2 = make_golden.HARM3D, make_golden.HARM
    d = capi.Device(0)
    try:
        d.load_tables()
        d.load_harm3d(dfile, hfile, 40, H3["spin"])
        cam = capi.make_camera(1, H3["gridvals"], (32, 32, 150), H3["spin"], H3["mu0"], H3["uout"], -0.5)
        em = capi.make_emis("MAXJUTT", H3["mbh"], 1.0, 6.0, H3["gmin"], 1e5, 3.5, muval=H3["muval"], mdot=H3["mdot"],
                            otherargs=(2.0,) + make_golden.MAXJUTT_W6)
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM3D", em, [FREQ_HARM], "formal")
        # six thermal evaluations per point, each through the folded-constant coefficient routine (measured 6.5e-7 of peak)
        assert np.array_equal(status, g["harm3d_npts"]) and stokes_err(pix[0], g["harm3d_w6_formal"]) < STOKES_TOL
        em = capi.make_emis("MAXCOMP", H3["mbh"], 1.0, 6.0, H3["gmin"], 1e5, 3.5, muval=H3["muval"], mdot=H3["mdot"],
                            otherargs=(2.0, 2.0, 1.0, 0.5, 0.25))
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM3D", em, [FREQ_HARM], "formal")
        assert stokes_err(pix[0], g["harm3d_maxcomp_sel2"]) < STOKES_TOL
        d.load_harm(os.path.join(str(tmp_path), "dump"), os.path.join(str(tmp_path), "dump040"), 40, H2["spin"])
        cam = capi.make_camera(1, H2["gridvals"], (32, 32, 150), H2["spin"], H2["mu0"], H2["uout"], -0.5)
        em = capi.make_emis("MAXJUTT", H2["mbh"], 1.0, 6.0, H2["gmin"], 1e5, 3.5, muval=H2["muval"], mdot=H2["mdot"], otherargs=(1.5, 1.0, 0.5))
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal")
        assert np.array_equal(status, g["harm_npts"]) and stokes_err(pix[0], g["harm_w2_formal"]) < STOKES_TOL
        with pytest.raises(capi.GrtError):          # MAXJUTT has no SPHACC instantiation: refused
            d.load_sphacc()
            capi.trace_tile(capi.make_camera(1, (-25., 25., -25., 25.), (8, 8, 50), 0.0, 0.5, 0.04, -0.5), 1, 64, "SPHACC", em, [FREQ_SPHACC], "formal")
    finally:
        d.close()
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    kw = dict(fname='HARM3D', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='MAXJUTT', nvals=4, spin=H3["spin"],
              standard=1, nn=[32, 32, 150], uout=H3["uout"], mbh=H3["mbh"], mdotmin=H3["mdot"], mdotmax=H3["mdot"], nmdot=1,
              mumin=H3["mu0"], mumax=H3["mu0"], gridvals=list(H3["gridvals"]), muval=H3["muval"], gmin=H3["gmin"], iname='formal',
              cflag=0, tdfile='dump3d', hhfile='dump3d_header', tindf=40, tnt=1, epotherargs=[2.0] + list(make_golden.MAXJUTT_W6))
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["harm3d_w6_formal"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))


def test_harm3d_camera_sweep_and_frequency_batches(tmp_path):
    # the 3-D sampler from other viewing angles (southern, near pole-on, near equatorial) against the oracle, and the
    # frequency-batch instantiations of the thermal / hybrid / MAXJUTT kernels: every camera of a multi-frequency run
    # equals its own single-frequency run
    import sys
    from grtrans_amd import capi
    from oracle import orabind
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_harm3d_dump
    dfile, hfile = make_harm3d_dump.write(str(tmp_path), b0=1.0, n1=32, n2=32, n3=16)
    H = make_golden.HARM3D
    d = capi.Device(0)
    try:
        d.load_tables()
        d.load_harm3d(dfile, hfile, 40, H["spin"])
        nn = (20, 20, 100)
        for mu0 in (-0.6, 0.95, 0.15):
            kw = dict(H, mu0=mu0, mdot=3e19)
            o = orabind.Oracle(nn=nn, iname="formal", datadir=str(tmp_path), **kw)
            ref, rn, _ = o.trace(1, 400, [FREQ_HARM], nthreads=0)
            cam = capi.make_camera(1, H["gridvals"], nn, H["spin"], mu0, H["uout"], -0.5)
            em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, H["gmin"], 1e5, 3.5, muval=H["muval"], mdot=3e19)
            pix, _, status, _ = capi.trace_tile(cam, 1, 400, "HARM3D", em, [FREQ_HARM], "formal")
            assert np.array_equal(status, rn)
            assert stokes_err(pix[0], ref[0]) < STOKES_TOL, mu0
        cam = capi.make_camera(1, H["gridvals"], nn, H["spin"], H["mu0"], H["uout"], -0.5)
        freqs = list(2.3e10 * np.exp(np.arange(15) * np.log(2.3e13 / 2.3e10) / 14))
        for en, iname, oa in (("POLSYNCHTH", "formal", (1.0, 1.0)), ("HYBRIDTHPL", "delo", (1.0, 1.0)), ("MAXJUTT", "formal", (2.0, 1.0, 0.5, 0.25))):
            em = capi.make_emis(en, H["mbh"], 1.0, 6.0, 50. if en == "HYBRIDTHPL" else H["gmin"], 1e5, 3.5, muval=H["muval"], mdot=3e19, otherargs=oa)
            multi, _, _, _ = capi.trace_tile(cam, 1, 400, "HARM3D", em, freqs, iname)
            for k in (0, 7, 8, 12, 14):
                single, _, _, _ = capi.trace_tile(cam, 1, 400, "HARM3D", em, [freqs[k]], iname)
                # optically thick low frequencies overflow the formal propagator to NaN on a few pixels (as in the reference):
                # same pixels in both runs; elsewhere equal (separate compilations of the batch and single-frequency
                # instantiations may differ in the last bits through fp contraction)
                nanm = np.isnan(multi[k]).any(axis=1)
                assert np.array_equal(nanm, np.isnan(single[0]).any(axis=1)) and nanm.mean() < 0.2, (en, k)
                assert stokes_err(multi[k][~nanm], single[0][~nanm]) < 1e-6, (en, k)
    finally:
        d.close()


def test_iharm_device_and_both_hosts_vs_reference_golden(tmp_path, monkeypatch):
    # iharm-format 3-D dumps, metric 0: product loader, nearest-upper-zone sampler, the three electron models
    # (gmin = 1, gmin = 3, gmin = -1: electron entropy), delo, sigcut; then the Fortran host and run_pgrtrans
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_iharm_dump
    g = np.load(os.path.join(G, "iharm_36x36x160.npz"))
    monkeypatch.chdir(tmp_path)
    dfile, hfile = make_iharm_dump.write(str(tmp_path), **make_golden.IHARM_ZONES)
    assert hashlib.sha256(open(dfile + "040.bin", "rb").read()).hexdigest() == str(g["dump_sha256"])
    H = make_golden.IHARM
    d = capi.Device(0)
    try:
        d.load_tables()
        with pytest.raises(capi.GrtError):
            d.load_iharm(dfile, hfile, 40, 0.5)
        d.load_iharm(dfile, hfile, 40, H["spin"])
        cam = capi.make_camera(1, H["gridvals"], (36, 36, 160), H["spin"], H["mu0"], H["uout"], -0.5)
        for key, iname, gmin, sig in (("formal_gmin1", "formal", 1., 1e10), ("delo_gmin1", "delo", 1., 1e10), ("formal_gmin3", "formal", 3., 1e10),
                                      ("formal_kel", "formal", -1., 1e10), ("formal_sigcut", "formal", 1., 1e-3)):
            em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, gmin, 1e5, 3.5, muval=H["muval"], mdot=H["mdot"], sigcut=sig)
            pix, _, status, _ = capi.trace_tile(cam, 1, 1296, "IHARM", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["npts"])
            assert stokes_err(pix[0], g["ivals_" + key]) < STOKES_TOL, key
        em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, 1., 1e5, 3.5, muval=H["muval"], mdot=H["mdot"])
        p1, _, _, _ = capi.trace_tile(cam, 1, 1296, "IHARM", em, [FREQ_HARM], "formal", nvals=1)
        assert p1.shape == (1, 1296, 1) and np.isfinite(p1).all() and p1.max() > 0
        em = capi.make_emis("HYBRIDTHPL", H["mbh"], 1.0, 6.0, 2., 1e5, 3.5, muval=H["muval"], mdot=H["mdot"], jetalpha=0.02)
        pix, _, _, _ = capi.trace_tile(cam, 1, 1296, "IHARM", em, [FREQ_HARM], "formal")
        assert stokes_err(pix[0], g["ivals_hybridthpl"]) < STOKES_TOL
        em = capi.make_emis("MAXJUTT", H["mbh"], 1.0, 6.0, 1., 1e5, 3.5, muval=H["muval"], mdot=H["mdot"], otherargs=(2.0, 1.0, 0.5, 0.25))
        pix, _, _, _ = capi.trace_tile(cam, 1, 1296, "IHARM", em, [FREQ_HARM], "formal")
        assert stokes_err(pix[0], g["ivals_maxjutt"]) < STOKES_TOL
        # MMKS (header metric 1): theta(x1, x2), zone search on the tabulated thetag / thetaj, Brent near zone-centre angles
        dfm, hfm = make_iharm_dump.write(str(tmp_path / "mmks"), metric=1, **make_golden.IHARM_ZONES)
        assert hashlib.sha256(open(dfm + "040.bin", "rb").read()).hexdigest() == str(g["mmks_dump_sha256"])
        d.load_iharm(dfm, hfm, 40, H["spin"])
        for key, iname, gmin in (("mmks_formal_gmin3", "formal", 3.), ("mmks_delo_kel", "delo", -1.)):
            em = capi.make_emis("POLSYNCHTH", H["mbh"], 1.0, 6.0, gmin, 1e5, 3.5, muval=H["muval"], mdot=H["mdot"])
            pix, _, status, _ = capi.trace_tile(cam, 1, 1296, "IHARM", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["npts"]) and stokes_err(pix[0], g["ivals_" + key]) < STOKES_TOL, key
        with pytest.raises(capi.GrtError):      # an emissivity without a kernel is refused (POLSYNCHPL on this sampler has one since r05)
            capi.trace_tile(cam, 1, 1296, "IHARM", capi.make_emis("SYNCHPL", H["mbh"], 1.0, 6.0, 50., 1e5, 3.5), [FREQ_HARM], "formal")
    finally:
        d.close()
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    kw = dict(fname='IHARM', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=H["spin"],
              standard=1, nn=[36, 36, 160], uout=H["uout"], mbh=H["mbh"], mdotmin=H["mdot"], mdotmax=H["mdot"], nmdot=1,
              mumin=H["mu0"], mumax=H["mu0"], gridvals=list(H["gridvals"]), muval=H["muval"], gmin=3., iname='formal',
              cflag=0, tdfile='dumpih', hhfile='dumpih_header', tindf=40, tnt=1)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal_gmin3"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))


def test_bench_distributed_path_through_rccl_single_rank():
    # bench.py's N>1 code path (process group over RCCL, barrier, gather of the fp32 tiles, all_reduce of the timings)
    # launched exactly as the driver does, with one rank on this 1-GPU box (GRT_BENCH_FORCE_DIST=1)
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--nn", "256", "256", "200", "--scaling", "weak", "--no-cpu-baseline"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["metric"] == "ray-steps/sec" and j["n_gpus"] == 1 and j["value"] > 1e8 and j["scaling"] == "weak"
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert j["config"]["gather_ms"] is not None and "grt_gather_stage" in j["config"]["gather_via"], j["config"]["gather_via"]
    # strong-scaling mode (BASELINE config 5's shape: one camera shared out in column tiles) through the same launcher; with
    # GRT_BENCH_CHECK_IMAGE the gathered, un-permuted image must equal a whole-image trace bit for bit.  Three frequencies.
    env["GRT_BENCH_CHECK_IMAGE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--nn", "128", "96", "200", "--nfreq", "3", "--scaling", "strong", "--no-cpu-baseline"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["scaling"] == "strong" and "grt_gather_stage" in j["config"]["gather_via"]
    # and the fall-back (the process group's own gather) gives the same image
    env["GRT_BENCH_GATHER"] = "torch"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--nn", "128", "96", "200", "--nfreq", "3", "--scaling", "strong", "--no-cpu-baseline"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "torch.distributed.gather" in j["config"]["gather_via"]


def test_bench_two_ranks_on_one_gpu_over_gloo():
    # bench.py with TWO ranks launched as the driver launches them, on this 1-GPU box: both ranks trace on device 0
    # (GRT_BENCH_DEVICE) and the collectives run over gloo (GRT_BENCH_BACKEND; RCCL cannot put two ranks on one GPU).  Covers the
    # rank-dependent logic of the N > 1 path -- pixel strips of the widened camera (weak), round-robin column tiles of one
    # camera (strong), max-over-ranks timing, sum of ray-steps, the gather and the un-permutation: with GRT_BENCH_CHECK_IMAGE the
    # assembled image must equal a whole-image trace bit for bit.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRT_BENCH_BACKEND="gloo", GRT_BENCH_DEVICE="0", GRT_BENCH_CHECK_IMAGE="1")
    for scaling, npix in (("weak", 96 * 96), ("strong", 96 * 96 // 2)):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                              "--nn", "96", "96", "100", "--scaling", scaling, "--tile-cols", "8", "--no-cpu-baseline"], env=env, cwd=root,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1                      # rank 0 alone prints
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["value"] > 1e7 and j["config"]["pixels_per_gpu"] == npix
        assert j["config"]["gather_ms"] is not None


def test_hosts_gather_path_with_a_one_rank_communicator(tmp_path, monkeypatch):
    # The N-GPU mode of module pgrtrans (one process per GPU; pixel ranges = strips or column tiles; tiles staged device to
    # device into the rank's image; ONE RCCL gather; rank 0 assembles ivals) on this 1-GPU box: GRT_FORCE_COMM=1 runs all of it
    # with a one-rank communicator, GRT_TILE_COLS=5 makes seven ranges of the 32 columns.  Must equal the plain run exactly,
    # through ./grtrans and through run_pgrtrans (the same Fortran module), two frequencies, two mdots (geodesic re-use).
    from grtrans_amd import grtrans_batch as gb
    kw = dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=2, nmu=1, fmin=FREQ, fmax=2 * FREQ,
              ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[32, 32, 100], uout=0.01, mbh=3.4e9, nmdot=2, mdotmin=1e15,
              mdotmax=2e15, mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70., iname='delo', cflag=0,
              i1=37, i2=1000)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='plain.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    plain = x.ivals.copy()
    assert plain.shape == (1024, 4, 4) and (plain[:36] == 0).all() and (plain[1000:] == 0).all() and np.abs(plain[36:1000]).max() > 0
    monkeypatch.setenv("GRT_FORCE_COMM", "1")
    monkeypatch.setenv("GRT_COMM_FILE", str(tmp_path / "comm_id"))
    for cols in ("0", "5"):
        monkeypatch.setenv("GRT_TILE_COLS", cols)
        if os.path.exists(str(tmp_path / "comm_id")):
            os.remove(str(tmp_path / "comm_id"))
        x.write_grtrans_inputs('inputs.in', oname='comm%s.out' % cols, **kw)
        x.run_grtrans()
        x.read_grtrans_output(bin=1)
        assert np.array_equal(x.ivals, plain), cols
        assert not os.path.exists(str(tmp_path / "comm_id"))       # rank 0 removes the rendezvous file once every rank has joined
        y = gb.grtrans()
        y.run_pgrtrans(**kw)
        assert y.ivals.shape == (4, 1024, 4)
        assert np.array_equal(y.ivals.transpose(1, 0, 2).astype(np.float32), plain), cols
        assert np.array_equal(y.ab.T.astype(np.float32)[36:1000], x.ab[36:1000])


def test_grt_gather_with_a_caller_side_view_of_the_communicator():
    # grt_gather (caller-owned ncclComm_t + device buffers) on a one-rank communicator: the root's own tile lands in d_full
    import ctypes as C
    import torch
    from grtrans_amd import capi
    L = capi.lib()
    dev = capi.Device(0)
    idb = C.create_string_buffer(128)
    capi.check(L.grt_comm_unique_id(idb), "grt_comm_unique_id")
    capi.check(L.grt_comm_init(C.c_int(0), C.c_int(1), idb), "grt_comm_init")
    comm, stream = C.c_void_p(), C.c_void_p()
    capi.check(L.grt_comm_handle(C.byref(comm), C.byref(stream)), "grt_comm_handle")
    assert comm.value
    tile = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    full = torch.zeros(1000, dtype=torch.float32, device="cuda")
    cnt = (C.c_int64 * 1)(1000)
    capi.check(L.grt_gather(C.c_void_p(tile.data_ptr()), C.c_int64(1000), C.c_void_p(full.data_ptr()), cnt, C.c_int(1), C.c_int(0),
                            C.c_int(0), comm, stream), "grt_gather")
    vals = (C.c_double * 2)(3.0, -1.0)
    capi.check(L.grt_comm_allreduce(vals, C.c_int(2), C.c_int(0)), "grt_comm_allreduce")     # also synchronises the library stream
    assert torch.equal(full, tile) and list(vals) == [3.0, -1.0]
    assert L.grt_gather(C.c_void_p(tile.data_ptr()), C.c_int64(1000), C.c_void_p(full.data_ptr()), cnt, C.c_int(1), C.c_int(3),
                        C.c_int(0), comm, stream) != 0          # bad rank: refused before any RCCL call, no group left open
    capi.check(L.grt_gather(C.c_void_p(tile.data_ptr()), C.c_int64(1000), C.c_void_p(full.data_ptr()), cnt, C.c_int(1), C.c_int(0),
                            C.c_int(0), comm, stream), "grt_gather")
    L.grt_comm_free()
    dev.close()


def test_koral3d_device_and_both_hosts_vs_reference_golden(tmp_path, monkeypatch):
    # KORAL 3-D dumps (KORAL3D and KORAL3D_DISK): product loader, eight-corner sampler with the closed-form theta <-> x2 maps,
    # T_e from the dump or charles_e, delo, disc cuts, sigma cut; then the Fortran host and run_pgrtrans; TOPJET refused
    import hashlib
    import subprocess
    import sys
    from grtrans_amd import capi, grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_golden
    import make_koral3d_dump
    g = np.load(os.path.join(G, "koral3d_36x36x160.npz"))
    monkeypatch.chdir(tmp_path)
    dfile, hfile = make_koral3d_dump.write(str(tmp_path), **make_golden.KORAL3D_ZONES)
    assert hashlib.sha256(open(dfile + "0040", "rb").read()).hexdigest() == str(g["dump_sha256"])
    H = make_golden.KORAL3D
    d = capi.Device(0)
    try:
        d.load_tables()
        with pytest.raises(capi.GrtError):
            d.load_koral3d(dfile, hfile, 40, 0.5)
        d.load_koral3d(dfile, hfile, 40, H["spin"])
        cam = capi.make_camera(1, H["gridvals"], (36, 36, 160), H["spin"], H["mu0"], H["uout"], -0.5)
        for key, fname, iname, gmin, sig in (("formal_te", "KORAL3D", "formal", -1., 1e10), ("delo_te", "KORAL3D", "delo", -1., 1e10),
                                             ("formal_gmin3", "KORAL3D", "formal", 3., 1e10), ("formal_disk", "KORAL3D_DISK", "formal", -1., 1e10),
                                             ("formal_sigcut", "KORAL3D", "formal", -1., 0.02)):
            em = capi.make_emis("POLSYNCHTH", H["mbh"], H["ftscl"], 6.0, gmin, 1e5, 3.5, muval=H["muval"], sigcut=sig)
            pix, _, status, _ = capi.trace_tile(cam, 1, 1296, fname, em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["npts"])
            assert stokes_err(pix[0], g["ivals_" + key]) < STOKES_TOL, key
        with pytest.raises(capi.GrtError):      # no kernel for power-law electrons on this sampler (ncgsnth = 0 there): refused
            capi.trace_tile(cam, 1, 1296, "KORAL3D", capi.make_emis("POLSYNCHPL", H["mbh"], 1.0, 6.0, 50., 1e5, 3.5), [FREQ_HARM], "formal")
    finally:
        d.close()
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    kw = dict(fname='KORAL3D_DISK', nfreq=1, nmu=1, fmin=FREQ_HARM, fmax=FREQ_HARM, ename='POLSYNCHTH', nvals=4, spin=H["spin"],
              standard=1, nn=[36, 36, 160], uout=H["uout"], mbh=H["mbh"], nmdot=1, mumin=H["mu0"], mumax=H["mu0"],
              gridvals=list(H["gridvals"]), muval=H["muval"], gmin=-1., iname='formal', cflag=0, hdfile='koral', hhfile='koral_header',
              hindf=40, hnt=1, ntscl=H["ftscl"])
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal_disk"]) < STOKES_TOL
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0].astype(np.float32))
    with pytest.raises(Exception):          # the jet variants read an uninitialised array in the reference: refused
        z = gb.grtrans()
        z.run_pgrtrans(**dict(kw, fname='KORAL3D_TOPJET'))


def test_multi_pass_tracing_under_a_memory_cap_equals_single_pass(monkeypatch):
    # the geodesic hand-over buffers are capped (GRT_MAX_WORK_BYTES): a tile beyond the cap is traced in passes over
    # sub-ranges of its rays, with the same pixels, statuses and step counts; the per-step record forces a single pass
    from grtrans_amd import capi
    d = capi.Device(0); d.load_ffjet(); d.load_tables()      # (earlier tests closed the module's device)
    nn = (48, 48, 100)
    full, _, sfull, stfull = run(nn, "formal")
    multi1, _, _, _ = run(nn, "formal", freqs=[2.3e11, FREQ])
    monkeypatch.setenv("GRT_MAX_WORK_BYTES", str(2 * 1024 * 1024))         # ~3.7 KB per ray here -> passes of ~560 rays
    part, _, spart, stpart = run(nn, "formal")
    assert np.array_equal(part, full) and np.array_equal(spart, sfull) and stpart.ray_steps == stfull.ray_steps
    assert stpart.rays_live == stfull.rays_live
    multi2, _, _, _ = run(nn, "formal", freqs=[2.3e11, FREQ])
    assert np.array_equal(multi2, multi1)
    a, _, sa, _ = run(nn, "delo", 101, 1900)                               # a tile that does not start at pixel 1
    monkeypatch.delenv("GRT_MAX_WORK_BYTES")
    b, _, sb, _ = run(nn, "delo", 101, 1900)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)
    d.close()



def test_fp32_islands_bit_identical_on_device_and_host():
    # tests/fp32_island_check.hip runs the REAL(4) building blocks of the fluid samplers (correctly rounded
    # transcendentals, divide, sqrt, the bilinear blend, the REAL(4) metric and LNRF transform) on a million inputs on the
    # device and on the host and compares bit patterns.  With -ffp-contract=fast the SLP-packed v_pk_fma_f32 ignored the
    # contract(off) pragma and a quarter of the blends were one ulp off.
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "fp32_island_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "csrc"), "../../tests/fp32_island_check"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_device_square_root_is_the_correctly_rounded_one_and_never_below_the_side_of_a_square():
    # r06: formal_apply forms rt = fsqrt(h^2 + ap^2) and lam2 = fsqrt(rt - h); with ap = 0 -- most cells of a GRMHD dump at some frequency -- a root
    # one ulp below |h| makes lam2 NaN.  A cheaper root (one third-order step, 0.75 ulp) did exactly that for one build (tools/experiments/r06_sqrt3.patch:
    # 147 of 700 sweep cameras).  tests/sqrt_square_check.hip runs the product's fsqrt over 2.7e8 (h, ap) pairs on the device: never below |h|, always
    # the correctly rounded root.
    import subprocess
    exe = os.path.join(ROOT, "tests", "sqrt_square_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grtrans_amd", "csrc"), "-f", "Makefile", "-f", "../../tests/checks.mk", "../../tests/sqrt_square_check"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "below |h|: 0; different from the correctly rounded root: 0" in r.stdout, r.stdout + r.stderr


# cameras a random sweep (tools/fuzz_cameras.py 300 31) found: one ray each crosses a bin edge of the power-law G(x)
# tables within 1e-7, where the reference's emissivity is discontinuous (0.5% in j_I); with the REAL(4) fluid variables
# one ulp off, that pixel moved by 4e-5 .. 2e-4 of the Stokes peak
BIN_FLIP_CAMS = [
    (-0.9685100068278336, 0.999, (-24.80464485595749, 32.04212916452268, -31.650583856482534, 25.196190163997635), 0.04, 60),
    (0.5, 0.999, (-38.13646298455784, 27.794847340807493, -39.85794389728005, 26.07336642808528), 0.04, 60),
    (0.5674915055438341, 0.999, (-40.73979773709904, 42.01881718390271, -41.68701977325127, 41.07159514775048), 0.02, 100),
    (0.48720383370589426, 0.999, (-13.171512055540557, 8.30424745186863, -16.81854110666402, 4.657218400745164), 0.02, 40),
    (-0.7092107459104384, 0.0, (-16.83951670346117, 9.247214594667486, -19.710436728366844, 6.376294569761814), 0.01, 100),
]


@pytest.mark.parametrize("spin,mu0,gv,uout,nup", BIN_FLIP_CAMS)
def test_table_bin_flip_cameras(spin, mu0, gv, uout, nup):
    from grtrans_amd import capi
    from oracle import orabind
    d = capi.Device(0); d.load_ffjet(); d.load_tables()      # tests above open and close their own devices
    nn = (12, 12, nup)
    o = orabind.Oracle(nn=nn, iname="formal", spin=spin, mu0=mu0, gridvals=gv, uout=uout)
    ref, rn, _ = o.trace(1, nn[0] * nn[1], [FREQ], nthreads=0)
    cam = capi.make_camera(1, gv, nn, spin, mu0, uout, -0.5)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    pix, _, status, _ = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [FREQ], "formal")
    d.close()
    assert np.array_equal(status, rn)
    assert stokes_err(pix[0], ref[0]) < STOKES_TOL


@pytest.mark.parametrize("mode,ncam,seed", [("ff", 60, 7), ("all", 60, 77)])
def test_random_cameras_vs_oracle(mode, ncam, seed):
    # tools/fuzz_cameras.py: random spins (both signs), inclinations (incl. face-on and edge-on), fields of view, uout,
    # integrators and -- in "all" mode -- SPHACC and the four GRMHD samplers on small synthetic dumps; stored-point
    # counts equal, non-finite masks equal, Stokes within 2e-6 of each peak (this sweep, at 300 cameras, is what found
    # the one-ulp REAL(4) differences and the cosh window above)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_cameras.py"), str(ncam), str(seed), mode],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = [l for l in r.stdout.strip().splitlines() if l.startswith("MISMATCHES:")][-1]
    assert last.startswith("MISMATCHES: 0"), r.stdout[-3000:]


@pytest.mark.parametrize("kind,ncam,need", [("npts", 30, 1.0), ("pix:1e-4", 40, 0.95)])
def test_fuzz_classifier_still_flags_an_injected_device_error(kind, ncam, need):
    # VERDICT r04 weak #1 / next #3b: the sweep's "no mismatch" rests on a classifier that waives cameras on which the REFERENCE is
    # ill-conditioned; nothing measured how many real device errors it would waive with them.  GRT_FUZZ_INJECT puts a known error into the
    # device result of every camera before the comparison -- one ray's stored-point count off by one; one Stokes value of one live pixel
    # moved by 1e-4 of its peak (50x the bar) -- and counts the cameras still flagged after classification.  With the r04 rules (camera-wide
    # bars) 79 % of the 1e-4 errors were flagged (200 cameras, profiles/r05c_inject_*); with the per-pixel bars of r05 99 % are
    # (profiles/r05e_inject_pix_1e-4_per_pixel_rules.log), with 0 / 0 / 0 / 1 cameras flagged on four clean seeds of 200.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_cameras.py"), str(ncam), "424", "allext"],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, GRT_FUZZ_INJECT=kind))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("INJECTED")][-1]
    m = re.search(r"into (\d+) cameras: (\d+) still flagged", line)
    ninj, nflag = int(m.group(1)), int(m.group(2))
    assert ninj >= 0.9 * ncam and nflag >= need * ninj, line


def test_scalar_emissivities_nvals1_vs_reference_golden(tmp_path):
    # SYNCHTHAV / SYNCHTHAVNOABS / BREMS / SYNCHTHBREMS / RHO with nvals = 1 (one kernel instantiation per fluid model,
    # em.type selects) on the 2-D and 3-D HARM dumps and the Bondi flow; nvals = 4 is refused (the reference returns NaN)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_golden
    import make_harm3d_dump
    import make_harm_dump
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "unpolarized_32x32x120.npz"))
    d3, h3 = make_harm3d_dump.write(str(tmp_path), **make_golden.HARM3D_ZONES)
    make_harm_dump.write(str(tmp_path), n=64)
    H2, H3, SP = make_golden.HARM, make_golden.HARM3D, make_golden.SPHACC
    d = capi.Device(0)
    try:
        d.load_tables(); d.load_sphacc()
        d.load_harm(os.path.join(str(tmp_path), "dump"), os.path.join(str(tmp_path), "dump040"), 40, H2["spin"])
        d.load_harm3d(d3, h3, 40, H3["spin"])
        cam = capi.make_camera(1, H2["gridvals"], (32, 32, 120), H2["spin"], H2["mu0"], H2["uout"], -0.5)
        for en in ("SYNCHTHAV", "SYNCHTHAVNOABS", "BREMS", "SYNCHTHBREMS", "RHO"):
            em = capi.make_emis(en, H2["mbh"], 1.0, 6.0, H2.get("gmin", 1.), 1e5, 3.5, muval=H2["muval"], mdot=H2["mdot"])
            pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal", nvals=1)
            assert np.array_equal(status, g["harm_npts"]), en
            assert stokes_err(pix[0], g["harm_%s_1_formal" % en].reshape(1024, 1)) < STOKES_TOL, en
        with pytest.raises(capi.GrtError):
            capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal")            # nvals = 4
        cam = capi.make_camera(1, H3["gridvals"], (32, 32, 120), H3["spin"], H3["mu0"], H3["uout"], -0.5)
        em = capi.make_emis("SYNCHTHBREMS", H3["mbh"], 1.0, 6.0, H3.get("gmin", 1.), 1e5, 3.5, muval=H3["muval"], mdot=H3["mdot"])
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM3D", em, [FREQ_HARM], "formal", nvals=1)
        assert np.array_equal(status, g["harm3d_npts"]) and stokes_err(pix[0], g["harm3d_SYNCHTHBREMS_1_formal"][0].reshape(1024, 1)) < STOKES_TOL
        cam = capi.make_camera(1, SP["gridvals"], (32, 32, 120), SP["spin"], SP["mu0"], SP["uout"], -0.5)
        em = capi.make_emis("SYNCHTHAV", SP["mbh"], 1.0, 6.0, SP.get("gmin", 100.), 1e5, 3.5)
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "SPHACC", em, [FREQ_SPHACC], "delo", nvals=1)
        assert np.array_equal(status, g["sphacc_npts"]) and stokes_err(pix[0], g["sphacc_SYNCHTHAV_1_formal"][0].reshape(1024, 1)) < STOKES_TOL
    finally:
        d.close()


def test_polsynchsymth_vs_reference_golden(tmp_path):
    # POLSYNCHSYMTH (Pandya et al. fitting functions): formal / DELO / nvals = 1 on the 2-D HARM dump, formal on the 3-D one
    # and on the Bondi flow
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_golden
    import make_harm3d_dump
    import make_harm_dump
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "sympol_32x32x120.npz"))
    d3, h3 = make_harm3d_dump.write(str(tmp_path), **make_golden.HARM3D_ZONES)
    make_harm_dump.write(str(tmp_path), n=64)
    H2, H3, SP = make_golden.HARM, make_golden.HARM3D, make_golden.SPHACC
    d = capi.Device(0)
    try:
        d.load_tables(); d.load_sphacc()
        d.load_harm(os.path.join(str(tmp_path), "dump"), os.path.join(str(tmp_path), "dump040"), 40, H2["spin"])
        d.load_harm3d(d3, h3, 40, H3["spin"])
        cam = capi.make_camera(1, H2["gridvals"], (32, 32, 120), H2["spin"], H2["mu0"], H2["uout"], -0.5)
        em = capi.make_emis("POLSYNCHSYMTH", H2["mbh"], 1.0, 6.0, H2.get("gmin", 1.), 1e5, 3.5, muval=H2["muval"], mdot=H2["mdot"])
        for iname in ("formal", "delo"):
            pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], iname)
            assert np.array_equal(status, g["harm_npts"]) and stokes_err(pix[0], g["harm_" + iname]) < STOKES_TOL, iname
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM", em, [FREQ_HARM], "formal", nvals=1)
        assert stokes_err(pix[0], g["harm_nvals1"].reshape(1024, 1)) < STOKES_TOL
        cam = capi.make_camera(1, H3["gridvals"], (32, 32, 120), H3["spin"], H3["mu0"], H3["uout"], -0.5)
        em = capi.make_emis("POLSYNCHSYMTH", H3["mbh"], 1.0, 6.0, H3.get("gmin", 1.), 1e5, 3.5, muval=H3["muval"], mdot=H3["mdot"])
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "HARM3D", em, [FREQ_HARM], "formal")
        assert np.array_equal(status, g["harm3d_npts"]) and stokes_err(pix[0], g["harm3d_formal"]) < STOKES_TOL
        cam = capi.make_camera(1, SP["gridvals"], (32, 32, 120), SP["spin"], SP["mu0"], SP["uout"], -0.5)
        em = capi.make_emis("POLSYNCHSYMTH", SP["mbh"], 1.0, 6.0, SP.get("gmin", 100.), 1e5, 3.5)
        pix, _, status, _ = capi.trace_tile(cam, 1, 1024, "SPHACC", em, [FREQ_SPHACC], "formal")
        assert np.array_equal(status, g["sphacc_npts"]) and stokes_err(pix[0], g["sphacc_formal"]) < STOKES_TOL
    finally:
        d.close()


def test_reference_regression_problem_sphacc_synchthav_through_the_fortran_host(tmp_path, monkeypatch):
    # the same problem through the file-based branch of the reference's script (write_grtrans_inputs -> ./grtrans ->
    # read_grtrans_output -> calc_spec with its 2*3.14), i.e. the Fortran 90 host: 25 cameras of a 10000 x 1 strip, nvals = 1
    import subprocess
    from grtrans_amd import grtrans_batch as gb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(gb.HOST_EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "host")])
    g = np.load(os.path.join(G, "test_grtrans_sphacc_lsoda.npz"))
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', fname='SPHACC', nfreq=25, nmu=1, fmin=1e8, fmax=1e15, ename='SYNCHTHAV', nvals=1, spin=0., mbh=1.,
                           standard=1, nn=[10000, 1, 100], gridvals=[0., 400., 0., 0.], uout=.0025, oname='sphacc_abs.out', iname='formal',
                           cflag=0)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    gi = g["intensity"].astype(float)
    terr = np.sum(np.abs(x.ivals[:, 0, 14] - gi)) / np.sum(np.abs(gi))
    assert terr < 0.1 and terr < 0.03, terr
    x.calc_spec(25)
    gs = g["spectrum"].astype(float)
    terr = np.sum(np.abs(x.spec - gs)) / np.sum(np.abs(gs))
    assert terr < 0.1, terr
    y = gb.grtrans()
    y.run_pgrtrans(fname='SPHACC', nfreq=25, nmu=1, fmin=1e8, fmax=1e15, ename='SYNCHTHAV', nvals=1, spin=0., mbh=1., standard=1,
                   nn=[10000, 1, 100], gridvals=[0., 400., 0., 0.], uout=.0025, iname='formal')
    assert np.array_equal(y.ivals[0].astype(np.float32), x.ivals[:, 0, :].astype(np.float32))      # both hosts, bit for bit


def test_reference_regression_problem_sphacc_synchthav_through_the_harness():
    # the reference's own first regression problem, called the way run_grtrans_test_problems_public.py:27-36 calls it
    # (pgrtrans branch), against the golden vectors the reference ships (LSODA; its criterion sum|x-g|/sum|g| < 0.1) and
    # against the oracle on the same inputs
    from grtrans_amd import grtrans_batch as gr
    from oracle import orabind
    g = np.load(os.path.join(G, "test_grtrans_sphacc_lsoda.npz"))
    x = gr.grtrans()
    x.run_pgrtrans(fname='SPHACC', nfreq=25, nmu=1, fmin=1e8, fmax=1e15, ename='SYNCHTHAV', nvals=1, spin=0., mbh=1., standard=1,
                   nn=[10000, 1, 100], gridvals=[0., 400., 0., 0.], uout=.0025, oname='sphacc_abs.out',
                   iname='formal')        # the reference's default is LSODA, which stays CPU-only: nvals = 1 runs the quadrature
    x.calc_spec_pgrtrans((np.shape(x.ivals))[2])
    gi = g["intensity"].astype(float)
    terr = np.sum(np.abs(x.ivals[0, :, 14] - gi)) / np.sum(np.abs(gi))
    assert terr < 0.1 and terr < 0.03, terr
    gs = g["spectrum"].astype(float)
    terr = np.sum(np.abs(x.spec - gs)) / np.sum(np.abs(gs))
    assert terr < 0.1, terr
    freqs = 1e8 * np.exp(np.arange(25) * np.log(1e15 / 1e8) / 24)
    o = orabind.Oracle(nn=(10000, 1, 100), iname="formal", nvals=1, fname="SPHACC", ename="SYNCHTHAV", spin=0.0, mbh=1.0, standard=1,
                       gridvals=(0., 400., 0., 0.), uout=0.0025, mu0=float(x.inputs.mumin))
    ref, npts, _ = o.trace(1, 10000, list(freqs), nthreads=0)
    # 2e-5 here: on this camera geokerr's affine parameter carries a constant of ~4e12 for the near-radial rays, so lambda
    # is quantised at 2^-11 = 4.9e-4 and every d(lambda) of the quadrature at ~5e-6 of its value (measured: 57 of the
    # 10000 pixels above 2e-6, the largest 9e-6; stored-point counts equal)
    assert np.array_equal(x.status if hasattr(x, "status") else npts, npts)
    for k in range(25):
        assert stokes_err(x.ivals[:, :, k].T, ref[k]) < 2e-5, k


def test_near_polar_ray_azimuth_on_a_3d_dump(tmp_path):
    # a face-on camera whose pixel 90 has l = -4e-5: the ray passes within 1e-5 rad of the axis.  On the fast path its
    # azimuth came out 1.49 rad off after the equator crossing on the device (found by tools/fuzz_cameras.py 400 21 all;
    # only a 3-D fluid model sees phi): 27 % of the Stokes peak.  Such rays (l^2 < 1e-6 (q2 + a^2)) now take the generic path.
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_iharm_dump
    from grtrans_amd import capi
    from oracle import orabind
    di, hi = make_iharm_dump.write(str(tmp_path), n1=32, n2=32, n3=16, metric=1)
    gv = (-7.183576656227343, 4.3115989742937995, -11.852649562661194, -0.3574739321400511)
    nn = (12, 12, 60)
    o = orabind.Oracle(nn=nn, iname="formal", spin=0.9375, mu0=0.999, gridvals=gv, uout=0.04, fname="IHARM", ename="POLSYNCHTH", datadir=str(tmp_path),
                       dfile="dumpih", hfile="dumpih_header", findf=40, mbh=4e6, mdot=1e20, muval=0.25, gmin=3.)
    ref, rn, _ = o.trace(1, 144, [FREQ_HARM], nthreads=0)
    assert abs(o.camera()["l"][89]) < 1e-4
    d = capi.Device(0)
    try:
        d.load_tables(); d.load_iharm(di, hi, 40, 0.9375)
        cam = capi.make_camera(1, gv, nn, 0.9375, 0.999, 0.04, -0.5)
        em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 3., 1e5, 3.5, muval=0.25, mdot=1e20)
        capi.record_steps(True)
        pix, _, status, _ = capi.trace_tile(cam, 1, 144, "IHARM", em, [FREQ_HARM], "formal")
        n = int(status[89])
        rec = capi.download_steps(89, 1, cam.nup + cam.next)[:, 0, :n]
    finally:
        capi.record_steps(False)
        d.close()
    assert np.array_equal(status, rn)
    od = o.ray_dump(90, FREQ_HARM)
    # azimuth along the near-polar ray: the reference's own formulas are ill-conditioned this close to the axis (device and
    # oracle, both literal, differ by up to 2e-5 rad at l^2/(q2+a^2) ~ 1e-11: tools/polar_phi.py); the fast path was off by radians
    # r03: the reference-literal code is compiled without FMA contraction and rounds like the oracle -- 3e-10 rad over every decade
    # of l^2/(q2+a^2) (tools/polar_phi.py); contracted, the same code was 2e-5 rad off here and radians off once fully inlined
    assert np.abs(rec[2] - od["x"][:n, 3]).max() < 1e-8
    assert stokes_err(pix[0], ref[0]) < STOKES_TOL


CONFIG5_FREQS = list(2.41e10 * np.exp(np.arange(25) * np.log(6.31e14 / 2.41e10) / 24))     # pgrtrans.f90:87-91, README:45


@pytest.mark.parametrize("iname", ["formal", "delo"])
def test_config5_frequency_grid_vs_compiled_reference(fdev, iname):
    # BASELINE config 5's 25 frequencies 2.41e10 .. 6.31e14 Hz through the frequency-batch kernels (FB = 13 formal / 8 DELO)
    # against the reference's own grtrans_main (tests/make_golden.py:config5).  Bar per frequency: 2e-6 of each Stokes
    # peak, or 10x the reference's own response to one-ulp input changes at that frequency where that is larger.
    g = np.load(os.path.join(G, "ffjet_nfreq25_64x64x400.npz"))
    assert np.allclose(g["nu"], np.float32(CONFIG5_FREQS), rtol=1e-7)
    pix, _, status, st = run((64, 64, 400), iname, freqs=CONFIG5_FREQS)
    ref = g["ivals_" + iname].astype(float)                  # [25][npix][4]
    peak = np.abs(ref).max(axis=1)
    err = np.abs(pix.astype(float) - ref).max(axis=1) / peak  # [25][4]
    bar = np.maximum(STOKES_TOL, 10.0 * g["cond_" + iname])
    for k in range(25):
        print("%s %.3e Hz: err I,Q,U,V = %s  bar = %s" % (iname, CONFIG5_FREQS[k], " ".join("%.1e" % v for v in err[k]),
                                                         " ".join("%.1e" % v for v in bar[k])))
    assert np.array_equal(np.isfinite(pix), np.isfinite(g["ivals_" + iname]))
    assert (err < bar).all(), np.argwhere(err >= bar)


def test_config5_full_size_properties(fdev):
    # 1024 x 1024 x 400, 25 frequencies, image only: size-independent properties at BASELINE's full size
    from grtrans_amd import capi
    nn = (1024, 1024, 400)
    capi.record_steps(False)
    _, _, s1, st1 = run(nn, "formal")
    full, _, s25, st25 = run(nn, "formal", freqs=CONFIG5_FREQS)
    assert np.array_equal(s1, s25) and st1.ray_steps == st25.ray_steps == int(s25.sum())
    assert np.isfinite(full).all()                           # the reference's mask on this camera is empty (64 x 64 golden)
    assert (full[:, :, 0] >= 0).all() and (full[:, s25 == 0] == 0).all()
    h = nn[0] * nn[1] // 2
    a, _, sa, _ = run(nn, "formal", 1, h, freqs=CONFIG5_FREQS)
    b, _, sb, _ = run(nn, "formal", h + 1, 2 * h, freqs=CONFIG5_FREQS)
    assert np.array_equal(np.concatenate([a, b], axis=1), full)
    assert np.array_equal(np.concatenate([sa, sb]), s25)
    # the frequency nearest the nominal one equals a single-frequency run of the same frequency (batch vs FB=1 kernel)
    k = 7
    one, _, _, _ = run(nn, "formal", freqs=[CONFIG5_FREQS[k]])
    assert np.array_equal(one[0], full[k])
    # and the 64 x 64 golden camera is the 16-fold subsample of no pixel grid of this one: compare integrated fluxes instead
    g = np.load(os.path.join(G, "ffjet_nfreq25_64x64x400.npz"))
    flux_full = full[:, :, 0].astype(float).sum(axis=1) / (1024.0 * 1024.0)
    flux_gold = g["ivals_formal"][:, :, 0].astype(float).sum(axis=1) / (64.0 * 64.0)
    assert np.allclose(flux_full, flux_gold, rtol=0.02), np.abs(flux_full / flux_gold - 1).max()


def _full_size_properties(trace, nn, inames, flux_ref, flux_rtol, record_rows=4096):
    """Size-independent properties of a BASELINE config at its full size (the oracle cannot finish these in seconds):
    ray-steps = sum of the stored-point counts, finite image, zero pixels exactly where no point was stored, two half-image
    calls = the whole image bit for bit, the per-step record on or off does not change the image, the image-integrated flux
    agrees with the small golden camera of the same field of view (flux_ref per Stokes I, rtol flux_rtol).
    trace(i1, i2, iname) -> pix[1, n, 4], status, stats."""
    from grtrans_amd import capi
    npix = nn[0] * nn[1]
    out = {}
    for iname in inames:
        capi.record_steps(False)
        full, s, st = trace(1, npix, iname)
        assert st.ray_steps == int(np.abs(s).sum()) and (np.abs(s) <= nn[2] + int(np.ceil(60.0 * nn[2] / 400.0))).all()
        finite = np.isfinite(full[0]).all(axis=1)
        assert finite.mean() > 0.999, (iname, 1 - finite.mean())       # (a thermal DELO pixel may overflow in the reference too: counted, not hidden)
        assert (full[0][s <= 0] == 0).all() and (full[0][finite, 0] >= 0).all()
        h = npix // 2
        a, sa, _ = trace(1, h, iname)
        b, sb, _ = trace(h + 1, npix, iname)
        assert np.array_equal(np.concatenate([a, b], axis=1), full, equal_nan=True) and np.array_equal(np.concatenate([sa, sb]), s)
        # the per-step record (azimuth integrals + I,Q,U,V rows) does not change what the image holds: a block of rows in the middle
        lo = h - record_rows // 2
        capi.record_steps(True)
        try:
            r, sr, _ = trace(lo + 1, lo + record_rows, iname)
        finally:
            capi.record_steps(False)
        assert np.array_equal(r, full[:, lo:lo + record_rows], equal_nan=True) and np.array_equal(sr, s[lo:lo + record_rows])
        flux = np.nansum(full[0][:, 0].astype(float)) / npix
        if flux_ref is not None:
            assert abs(flux / flux_ref[iname] - 1.0) < flux_rtol, (iname, flux, flux_ref[iname])
        out[iname] = dict(ray_steps=int(st.ray_steps), nonfinite=int((~finite).sum()), flux=flux, ms=st.ms_total)
    if os.environ.get("GRT_TEST_ERRLOG"):
        with open(os.environ["GRT_TEST_ERRLOG"], "a") as f:
            f.write("full_size %s %s\n" % (list(nn), out))
    return out


def test_config2_full_size_properties(fdev):
    # BASELINE config 2 at its full size: SPHACC POLSYNCHTH nvals=4 spin=0 nn=[256,256,200], formal and DELO
    from grtrans_amd import capi
    fdev.load_sphacc()
    nn = (256, 256, 200)
    cam = capi.make_camera(1, (-25., 25., -25., 25.), nn, 0.0, 0.5, 0.04, -0.5)
    em = capi.make_emis("POLSYNCHTH", 10.0, 1.0, 6.0, 100., 1e5, 3.5)

    def trace(i1, i2, iname):
        pix, _, s, st = capi.trace_tile(cam, i1, i2, "SPHACC", em, [FREQ_SPHACC], iname)
        return pix, s, st
    g = np.load(os.path.join(G, "sphacc_64x64x200.npz"))
    ref = {k: g["ivals_" + k][:, 0].astype(float).sum() / 4096.0 for k in ("formal", "delo")}
    _full_size_properties(trace, nn, ("formal", "delo"), ref, 0.03)


def test_config3_full_size_properties(fdev):
    # BASELINE config 3 at its full size: FFJET POLSYNCHPL nvals=4 spin=0.998 nn=[512,512,400], formal and DELO
    nn = (512, 512, 400)

    def trace(i1, i2, iname):
        pix, _, s, st = run(nn, iname, i1, i2)
        return pix, s, st
    g = np.load(os.path.join(G, "ffjet_100x100x400.npz"))
    ref = {k: g["ivals_" + k][:, 0].astype(float).sum() / 10000.0 for k in ("formal", "delo")}
    _full_size_properties(trace, nn, ("formal", "delo"), ref, 0.03)


def test_config4_full_size_properties(fdev, tmp_path):
    # BASELINE config 4 at its full size: HARM POLSYNCHTH nvals=4 nn=[512,512,400] on the 256 x 256-zone synthetic dump
    # (dump040 itself does not ship with the reference).  r04: the flux reference is the COMPILED REFERENCE's 64 x 64 x 400 image of that
    # same 256-zone dump (tests/golden/harm_zones256_64x64x400.npz), and the device's own 64 x 64 image is held to it pixel by pixel
    # first -- the full-size image is then compared with the reference, not with the device itself.
    import hashlib
    import sys
    from grtrans_amd import capi
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_harm_dump
    make_harm_dump.write(str(tmp_path), n=256)
    fdev.load_harm(os.path.join(str(tmp_path), "dump"), os.path.join(str(tmp_path), "dump040"), 40, 0.9375)
    em = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1., 1e5, 3.5, muval=0.25, mdot=1.57e15)
    g = np.load(os.path.join(G, "harm_zones256_64x64x400.npz"))
    assert hashlib.sha256(open(os.path.join(str(tmp_path), "dump040"), "rb").read()).hexdigest() == str(g["dump_sha256"])
    small = capi.make_camera(1, (-13., 13., -13., 13.), (64, 64, 400), 0.9375, 0.6428, 0.04, -0.5)
    ref = {}
    for iname in ("formal", "delo"):
        p, _, st, _ = capi.trace_tile(small, 1, 4096, "HARM", em, [FREQ_HARM], iname)
        assert np.array_equal(st, g["npts"])
        want = g["ivals_" + iname]
        assert np.array_equal(np.isfinite(p[0]), np.isfinite(want))
        ok = np.isfinite(want).all(axis=1)
        assert stokes_err(p[0][ok], want[ok]) < STOKES_TOL, iname
        ref[iname] = np.nansum(want[:, 0].astype(float)) / 4096.0
    nn = (512, 512, 400)
    cam = capi.make_camera(1, (-13., 13., -13., 13.), nn, 0.9375, 0.6428, 0.04, -0.5)

    def trace(i1, i2, iname):
        pix, _, s, st = capi.trace_tile(cam, i1, i2, "HARM", em, [FREQ_HARM], iname)
        return pix, s, st
    _full_size_properties(trace, nn, ("formal", "delo"), ref, 0.05)


def test_geokerr_rays_abi_vs_reference_geokerr(fdev):
    # G-10, class_geokerr's f2py surface (class_geokerr.f90:69-144): `pixel` -> per-pixel constants, `call_geokerr` ->
    # GEOKERR's raw arrays u, mu, dt, dphi, lam, tpmi, tpri for caller-given rays.  Goldens: the compiled reference's
    # GEOKERR on the FFJET regression camera (raw arrays of nine rays; tpmi, tpri, ncase of all 10000).
    from grtrans_amd import capi
    nn = (100, 100, 400)
    g = np.load(os.path.join(G, "ffjet_100x100x400.npz"))
    gr = np.load(os.path.join(G, "ffjet_rays_100x100x400.npz"))
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    px = capi.camera_pixels(cam, 1, 10000)
    for k in ("alpha", "beta", "q2", "l", "su", "sm", "uf"):
        assert np.array_equal(px[k], g["cam_" + k]), k
    assert np.array_equal(px["tpr"], g["cam_tpr"]) and px["u0"] == g["cam_scal"][0] and px["offset"] == g["cam_scal"][2]
    out = capi.geokerr_rays(px["u0"], cam.uout, cam.mu0, cam.a, px["offset"], nn[2], px["uf"], px["l"], px["q2"], px["alpha"],
                            px["beta"], px["tpr"], px["su"], px["sm"], mufill=cam.mufill, kext=cam.kext, next=cam.next)
    assert np.array_equal(out["npts"], g["npts"]) and np.array_equal(out["ncase"], g["ncase"])
    m = out["tpmi"].shape[1]
    assert np.array_equal(out["tpmi"], g["tpmi"][:, :m]) and np.array_equal(out["tpri"], g["tpri"][:, :m])      # bit-exact integers
    for i in gr["rays"]:
        n = int(gr["ray%d_npts" % i])
        for k, tol in (("u", 1e-12), ("mu", 1e-10), ("dphi", 1e-9), ("dt", 1e-13), ("lam", 1e-13)):
            ref = gr["ray%d_raw_%s" % (i, k)]
            err = np.abs(out[k][i - 1, :n] - ref).max() / np.abs(ref).max()
            assert err < tol, (i, k, err)
        assert (out["u"][i - 1, n:] == 0).all()
    # a caller's own subset in its own order (not a camera): rays 5051, 1, 3366 -> the same rows
    sel = np.array([5051, 1, 3366]) - 1
    sub = capi.geokerr_rays(px["u0"], cam.uout, cam.mu0, cam.a, px["offset"], nn[2], px["uf"][sel], px["l"][sel], px["q2"][sel],
                            px["alpha"][sel], px["beta"][sel], px["tpr"][sel], px["su"][sel], px["sm"][sel], mufill=cam.mufill,
                            kext=cam.kext, next=cam.next)
    for k in ("u", "mu", "dt", "dphi", "lam", "tpmi", "tpri"):
        assert np.array_equal(sub[k], out[k][sel]), k
    with pytest.raises(capi.GrtError):      # mu as the independent variable is refused, not emulated
        capi.geokerr_rays(px["u0"], cam.uout, cam.mu0, cam.a, 0.5, 10, px["uf"][:4], px["l"][:4], px["q2"][:4], px["alpha"][:4],
                          px["beta"][:4], px["tpr"][:4], px["su"][:4], px["sm"][:4], usegeor=1, mufill=0)


def test_resident_geodesic_is_reused_for_the_next_parameter_set(fdev):
    # grtrans_driver.f90:134: one geodesic serves every m = 1..nparams.  A second trace of the same camera and pixel range
    # (other emissivity parameters, other frequencies) re-launches only the transfer kernels and gives the bits of a cold run.
    import ctypes as C
    from grtrans_amd import capi
    nn = (48, 48, 200)
    cam = capi.make_camera(1, (-40., 20., -20., 40.), nn, 0.998, 0.906, 0.01, -0.5)
    em1 = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    em2 = capi.make_emis("POLSYNCHPL", 3.4e9, 3.0, 40.0, 50., 1e5, 3.5)
    L = capi.lib()
    L.grt_geo_passes.restype = C.c_longlong
    for rec in (False, True):
        capi.record_steps(rec)
        n0 = L.grt_geo_passes()
        capi.trace_tile(cam, 1, 2304, "FFJET", em1, [FREQ], "formal")
        warm, _, sw, stw = capi.trace_tile(cam, 1, 2304, "FFJET", em2, [FREQ, 2 * FREQ], "delo")
        assert L.grt_geo_passes() == n0 + 1                      # the second call ran no set-up / geodesic kernel
        capi.geo_reuse(False)
        cold, _, sc, stc = capi.trace_tile(cam, 1, 2304, "FFJET", em2, [FREQ, 2 * FREQ], "delo")
        capi.geo_reuse(True)
        assert L.grt_geo_passes() == n0 + 2
        assert np.array_equal(warm, cold) and np.array_equal(sw, sc) and stw.ray_steps == stc.ray_steps
        # another pixel range or camera drops the record
        capi.trace_tile(cam, 1, 2304, "FFJET", em1, [FREQ], "formal")
        other, _, _, sto = capi.trace_tile(cam, 1, 1000, "FFJET", em2, [FREQ], "delo")
        assert L.grt_geo_passes() == n0 + 4 and np.array_equal(other[0], cold[0, :1000])
    capi.record_steps(False)


def test_more_frequencies_than_any_batch_limit(fdev):
    # pgrtrans.f90:87-91 has no limit on nfreq; 40 cameras of one geodesic in one call equal their own single runs
    nn = (12, 12, 60)
    freqs = list(1e10 * np.exp(np.arange(40) * np.log(1e15 / 1e10) / 39))
    multi, _, _, _ = run(nn, "formal", freqs=freqs)
    assert multi.shape[0] == 40
    for k in (0, 13, 31, 32, 39):
        single, _, _, _ = run(nn, "formal", freqs=[freqs[k]])
        assert np.array_equal(multi[k], single[0]), k


def test_polarized_thindisk_bbpol_and_one_point_diagnostics(fdev, tmp_path, monkeypatch):
    # The reference's `thindisk` regression problem (run_grtrans_test_problems_public.py:111-129): THINDISK + BBPOL, nvals = 4,
    # standard = 2, nup = 1 -- Chandrasekhar's limb darkening and polarization degree, rotated by the transported basis --
    # and with extra=1 the 7 one-point diagnostics of grtrans_driver.f90:298-311.  Golden = the compiled reference's own
    # grtrans_main.  Through the C ABI, ./grtrans and run_pgrtrans.
    from grtrans_amd import capi, grtrans_batch as gb
    g = np.load(os.path.join(G, "thindisk_bbpol_100x100x1.npz"))
    fdev.set_thindisk(0.1, -1., 1e8)
    fdev.load_chandra()
    cam = capi.make_camera(2, (-21., 21., -21., 21.), (100, 100, 1), 0.9, 0.26, 0.01, -0.5)
    em = capi.make_emis("BBPOL", 10., 1.0, 6.0, 100., 1e5, 3.5)
    freqs = list(2.41e16 * np.exp(np.arange(3) * np.log(6.31e18 / 2.41e16) / 2))
    pix, _, status, st = capi.trace_tile(cam, 1, 10000, "THINDISK", em, freqs, "lsoda", nvals=4)
    assert pix.shape == (3, 10000, 4) and st.ray_steps == int((status > 0).sum())
    assert stokes_err(pix[:, :, :3], g["ivals"][:, :, :3]) < STOKES_TOL
    assert (pix[:, :, 3] == 0).all() and (g["ivals"][:, :, 3] == 0).all()
    pix7, _, _, _ = capi.trace_tile(cam, 1, 10000, "THINDISK", em, freqs, "lsoda", nvals=4, nextra=7)
    assert np.array_equal(pix7[:, :, :4], pix)
    ref = g["ivals_extra"].astype(float)
    for k in range(4, 11):
        assert np.abs(pix7[:, :, k] - ref[:, :, k]).max() / np.abs(ref[:, :, k]).max() < 2e-5, k
    with pytest.raises(capi.GrtError):      # nvals = 1 with extra=1: the reference reads an unloaded table there
        capi.trace_tile(cam, 1, 100, "THINDISK", capi.make_emis("BB", 10., 1.0, 6.0, 100., 1e5, 3.5), freqs, "lsoda", nvals=1, nextra=7)
    kw = dict(fname='THINDISK', nfreq=3, nmu=1, fmin=2.41e16, fmax=6.31e18, ename='BBPOL', nvals=4, spin=0.9, standard=2, nn=[100, 100, 1],
              uout=0.01, mbh=10, mumin=.26, mumax=.26, gridvals=[-21, 21, -21, 21], extra=1, cflag=0)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert x.ivals.shape == (10000, 11, 3) and np.array_equal(x.ivals.transpose(2, 0, 1), pix7)
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert y.ivals.shape == (11, 10000, 3) and np.array_equal(y.ivals.transpose(2, 1, 0).astype(np.float32), pix7)


@pytest.mark.parametrize("key", ["a", "b"])
def test_standard2_with_many_points_vs_reference(fdev, key):
    # standard=2 with nup > 1 (G-6: GEOR as the driver of the geodesic, geokerr_wrapper.f:421-461): mu steps from the camera's mu0
    # to the equatorial plane, points with u_f < 0 trimmed from both ends and lambda referred to LAMBDAI(npts') as
    # geodesics.f90:298-347 writes it.  Checked against the compiled reference: trimmed point counts (and the status of rays
    # whose sum of r is negative) bit-exact on every ray; the per-step record r, theta, phi, lambda of the golden rays -- some
    # of them trimmed -- against the raw GEOKERR arrays put through geokerr_wrapper's own post-processing.
    # (Images are NOT compared: a standard=2 ray starts at r = 1/u0 >= 1e4, where the reference's FFJET sampler reads past its
    # table and the GRMHD samplers return NaN for a third of the pixels -- there is no defined answer to match)
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "standard2_ffjet.npz"))
    spin, mu0, a1, a2, b1, b2, nx, ny, nup = g[key + "_cam"]
    nn = (int(nx), int(ny), int(nup))
    cam = capi.make_camera(2, (a1, a2, b1, b2), nn, spin, mu0, 0.01, -0.5)
    assert cam.usegeor == 1 and cam.mufill == 0
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    capi.record_steps(True)
    pix, _, status, st = capi.trace_tile(cam, 1, nn[0] * nn[1], "FFJET", em, [FREQ], "formal")
    assert np.array_equal(status, g[key + "_npts"]), np.where(status != g[key + "_npts"])[0][:5]
    assert st.ray_steps == int(np.abs(status).sum())
    ntrim = 0
    for i in g[key + "_rays"]:
        raw = {k: g["%s_ray%d_%s" % (key, i, k)] for k in ("u", "mu", "dphi", "lam")}
        n = len(raw["u"])
        i1, i2 = 1, n                                   # geodesics.f90:298-323
        if raw["u"][0] < 0:
            i1 = next((k + 1 for k in range(1, n) if raw["u"][k] >= 0), 1)
        if raw["u"][n - 1] < 0:
            i2 = next((n - k for k in range(1, n) if raw["u"][n - k - 1] >= 0), n)
        npr = i2 - i1 + 1
        assert npr == abs(status[i - 1])
        ntrim += int(npr < n)
        rec = capi.download_steps(int(i) - 1, 1, nn[2])[:, 0, :npr]
        sl = slice(i1 - 1, i2)
        want = [1.0 / raw["u"][sl], np.arccos(raw["mu"][sl]), cam.phi0 - raw["dphi"][sl], raw["lam"][npr - 1] - raw["lam"][sl]]
        for f, tol in ((0, 1e-10), (1, 1e-9), (2, 1e-8)):
            ok = np.isfinite(want[f])
            assert np.abs(rec[f][ok] - want[f][ok]).max() <= tol * (np.abs(want[f][ok]).max() + 1e-300), (i, f)
        # lambda is a difference of LAMBDAI ~ 1/u0 ~ 5e8: a few units of ITS last place
        assert np.abs(rec[3] - want[3]).max() <= 8 * np.spacing(np.abs(raw["lam"]).max()), i
    assert ntrim >= 1 or key == "b"
    capi.record_steps(False)
    # the HARM image of the same kind of camera: status again bit-exact (other root cases, more trimmed rays)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    if key == "a":
        import tempfile
        import make_harm_dump
        td = tempfile.mkdtemp(prefix="grt_std2_")
        make_harm_dump.write(td, n=64)
        fdev.load_harm(os.path.join(td, "dump"), os.path.join(td, "dump040"), 40, 0.9375)
        camh = capi.make_camera(2, (-13., 13., -13., 13.), (24, 24, 64), 0.9375, 0.6428, 0.04, -0.5)
        emh = capi.make_emis("POLSYNCHTH", 4e6, 1.0, 6.0, 1.0, 1e5, 3.5, mdot=1.57e15)
        _, _, sh, _ = capi.trace_tile(camh, 1, 576, "HARM", emh, [FREQ_HARM], "formal")
        assert np.array_equal(sh, g["harm_npts"])


def test_polar_camera_grid_nrotype1(fdev, tmp_path, monkeypatch):
    # nrotype = 1 (geokerr_wrapper.f:138-149, 263-272): pixels on rings ro_i = r1 (rcut/r1)^(i/nro), r1 = gridvals(1), at azimuths
    # 2 pi (j - 1/2)/nphi; u0 from rcut^2.  Golden = the reference's own main.  C ABI, ./grtrans and run_pgrtrans.
    from grtrans_amd import capi, grtrans_batch as gb
    g = np.load(os.path.join(G, "ffjet_polar_24x16x100.npz"))
    cam = capi.make_camera(1, (2., 0., 0., 0.), (24, 16, 100), 0.998, 0.906, 0.01, -0.5, nrotype=1, rcut=30.)
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    for iname in ("formal", "delo"):
        pix, loc, status, _ = capi.trace_tile(cam, 1, 384, "FFJET", em, [FREQ], iname)
        assert stokes_err(pix[0], g["ivals_" + iname]) < STOKES_TOL, iname
        assert np.array_equal(loc, g["ab"])
    kw = dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=1, nmu=1, fmin=FREQ, fmax=FREQ, ename='POLSYNCHPL', nvals=4,
              spin=0.998, standard=1, nn=[24, 16, 100], uout=0.01, mbh=3.4e9, mumin=.906, mumax=.906, gridvals=[2., 0., 0., 0.], rcut=30.,
              nrotype=1, ntscl=2., nrscl=70., iname='formal', cflag=0)
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', oname='grtrans.out', **kw)
    x.run_grtrans()
    x.read_grtrans_output(bin=1)
    assert stokes_err(x.ivals[:, :, 0], g["ivals_formal"]) < STOKES_TOL and np.array_equal(x.ab, g["ab"])
    y = gb.grtrans()
    y.run_pgrtrans(**kw)
    assert np.array_equal(y.ivals[:, :, 0].T.astype(np.float32), x.ivals[:, :, 0])


def test_quadrature_survives_a_step_of_negative_optical_depth(fdev):
    # tools/fuzz_cameras.py 700 5151 allext, camera 290: a spin-0.985 camera on the Schwarzschild SPHACC flow has a ray (pixel 91) whose affine
    # parameter jumps back by 2.3 where it dives under r = 2.6, a step of optical depth -1200 at 24 GHz.  The reference's nvals = 1
    # quadrature sums j exp(-tau) with tau measured from the camera and never forms e^{+1200}; the device recurrence
    # I <- e^{-dtau} I + ... did (inf - inf = NaN in that pixel, every library up to r03_e) and now banks the negative depth in an exponent
    # until the later steps have paid it back (grt_kernels.hip, IFLAG == GRT_INT_QUAD).
    from grtrans_amd import capi
    from oracle import orabind
    fdev.load_sphacc()
    gv = (-29.623082372504165, 17.25895576416562, -21.347376884693908, 25.534661251975876)
    spin, mu0, nn, freqs = 0.9845592285987614, -0.6534576984273296, (12, 12, 40), [7.23e11, 2.41e11, 2.41e10]
    cam = capi.make_camera(1, gv, nn, spin, mu0, 0.02, -0.5)
    em = capi.make_emis("SYNCHTHAV", 10.0, 1.0, 6.0, 100., 1e5, 5.5)
    o = orabind.Oracle(nn=nn, iname="formal", spin=spin, mu0=mu0, gridvals=gv, uout=0.02, fname="SPHACC", ename="SYNCHTHAV", mbh=10., gmin=100., nvals=1)
    ref, rn, _ = o.trace(1, 144, freqs, nthreads=0)
    for rec in (False, True):
        capi.record_steps(rec)
        pix, _, status, _ = capi.trace_tile(cam, 1, 144, "SPHACC", em, freqs, "formal", nvals=1)
        assert np.array_equal(status, rn)
        assert np.isfinite(ref[2][90]).all() and ref[2][90][0] > 0         # the pixel the device used to lose
        for k in range(3):
            assert np.array_equal(np.isfinite(pix[k]), np.isfinite(ref[k])), k
            ok = np.isfinite(ref[k]).all(axis=1)
            assert np.abs(pix[k][ok] - ref[k][ok]).max() <= STOKES_TOL * np.abs(ref[k][ok]).max(), k
    capi.record_steps(False)


def test_delo_step_on_the_device_with_subnormal_absorption():
    # ADVICE r03 (medium) / VERDICT r03 item 3: tests/delo_tiny_check.hip runs grt_dev_phys.hpp:delo_apply ON THE DEVICE for 16,384 steps
    # between a near-vacuum zone (alpha_I 1e-305 ... 1e-312, at and below the smallest normal number) and an emitting one, in both
    # orders and mixed with ordinary steps in the same waves, and compares with the reference-literal affine map (IEEE divisions by
    # alpha_I first, cofactor inverse: radtrans_integrate.f90:746-837) evaluated on the host.  Before r04 the device returned NaN
    # there (the Newton reciprocal of 1e-310 overflows) where the reference has finite Stokes values.
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "delo_tiny_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "grtrans_amd", "csrc"), "../../tests/delo_tiny_check"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"ordinary (\S+) far_vacuum (\S+) near_vacuum (\S+) both (\S+) nonfinite (\d+) ref_nonfinite (\d+)", r.stdout)
    assert m, r.stdout
    worst = [float(m.group(k)) for k in range(1, 5)]
    assert int(m.group(5)) == 0 and int(m.group(6)) == 0, r.stdout
    assert max(worst) < 1e-10, r.stdout            # (1e-13 ... 1e-12 measured: the two forms differ by rounding only)


# The open cases of the r03 sweeps, each against the COMPILED reference's image of the exact camera
# (tests/golden/open_cases.npz <- tests/make_golden.py:open_cases) with a bar stated from the compiled reference's own behaviour:
# `cond` = how far the reference's own pixels move when the frequencies move by one ulp / 1e-13 or the camera by one ulp / 1e-15 rad.
OPEN_CASE_SETUP = {
    # name: (fluid, emissivity arguments of the sweep, camera, frequencies, pixel range, integrator)
    "koral3d_delo_69ghz": ("KORAL3D", dict(mbh=4e6, nfac=1.5, bfac=6.0, gmin=-1., p1=5.5), 0.9375, 0.0, 0.04, (12, 12, 60),
                           (-9.572477643577713, 5.75296751663922, -11.596725842453694, 3.7287193177632396), [2.3e11, 6.9e10], 43, 132, "delo"),
    "sphacc_formal_equatorial": ("SPHACC", dict(mbh=10., nfac=1.0, bfac=6.0, gmin=100., p1=3.0), 0.0, 0.0, 0.04, (12, 12, 60),
                                 (-36.402300212602, 48.39734536577427, -41.53000646055861, 43.26963911781766), [7.23e11, 2.41e11, 2.41e10], 71, 110, "formal"),
    "sphacc_delo_24ghz": ("SPHACC", dict(mbh=10., nfac=1.0, bfac=6.0, gmin=100., p1=3.5), 0.0, 0.0, 0.01, (12, 12, 100),
                          (-6.46579504122852, 4.212597494570896, -9.68891574108579, 0.9894767947136254), [7.23e11, 2.41e11, 2.41e10], 1, 144, "delo"),
}


@pytest.mark.parametrize("name", sorted(OPEN_CASE_SETUP))
def test_open_cases_of_the_r03_sweeps_vs_compiled_reference(fdev, tmp_path, name):
    # koral3d_delo_69ghz (seed 31337 camera 524): rays that start in near-vacuum zones, j and alpha_I ~ 1e-306 ... 1e-310; DELO's first
    #   emitting step works with S = j / alpha_I of those.  The reference is STABLE here (cond ~ 4e-11): the bar is the suite's 2e-6 of
    #   each Stokes peak (r03: 2.0e-4 with the plain closed-form step, 7.8e-6 with the reference-literal map).  What it was: DELO divides
    #   the NEXT point's Faraday terms by its alpha_I (rho_V / alpha_I ~ 1e303 at the vacuum point beside the first emitting one), which
    #   turns the 1e-308-sized intensity gathered in the vacuum into 1e-6 of the pixel; the reference gathers it from j / fac (53
    #   significant bits), the device gathered it from j of a few units of 4.9e-324.  r04 integrates j scaled by 2^128
    #   (grt_kernels.hip:GRT_JSCALE) and rescales a thick step's end with alpha_I below 1e-140 (grt_dev_phys.hpp:delo_apply): 9e-8.
    # sphacc_formal_equatorial (seed 777 camera 240): GEOKERR's lambda carries a constant of 9e8 on the equatorial Schwarzschild
    #   camera, so d(lambda) of the topmost steps is quantised at 7e-6 of its value IN THE REFERENCE: its own 24 GHz image moves by
    #   6.2e-6 of the U peak when the inclination moves by 1e-15 rad (cond).  Bar: 3 x cond per frequency and Stokes parameter, and
    #   never below the suite's 2e-6.
    # sphacc_delo_24ghz (seed 5151 camera 303): DELO at tens of optical depths per step (alternating-sign growth, intensities of
    #   +-860 against 1e-3 at the higher frequencies): the reference moves by 1.2e-6 under the same perturbations and by 6,900 x any
    #   relative change of the frequency.  Same bar.
    from grtrans_amd import capi
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    g = np.load(os.path.join(G, "open_cases.npz"))
    fname, ekw, spin, mu0, uout, nn, gv, freqs, i1, i2, iname = OPEN_CASE_SETUP[name]
    if fname == "KORAL3D":
        import make_koral3d_dump
        dk, hk = make_koral3d_dump.write(str(tmp_path), n1=32, n2=32, n3=16)
        fdev.load_koral3d(dk, hk, 40, 0.9375)
    else:
        fdev.load_sphacc()
    cam = capi.make_camera(1, gv, nn, spin, mu0, uout, -0.5)
    em = capi.make_emis("POLSYNCHTH", ekw["mbh"], ekw["nfac"], ekw["bfac"], ekw["gmin"], 1e5, ekw["p1"], muval=0.25, mdot=0.0)
    ref, pk, cond = g[name + "_ivals"].astype(np.float64), g[name + "_peak"], g[name + "_cond"]
    bar = np.maximum(2e-6, 3.0 * cond)      # (koral3d_delo_69ghz: cond ~ 4e-11, so the suite's 2e-6; measured 9e-8 since the 2^128 scale of j)
    for rec in (False, True):
        capi.record_steps(rec)
        pix, _, status, _ = capi.trace_tile(cam, i1, i2, fname, em, freqs, iname)
        assert np.array_equal(status, g[name + "_npts"])
        for k in range(len(freqs)):
            assert np.array_equal(np.isfinite(pix[k]), np.isfinite(ref[k])), (name, k)
            okm = np.isfinite(ref[k]).all(axis=1)
            err = np.abs(pix[k][okm].astype(np.float64) - ref[k][okm]).max(axis=0) / pk[k]
            assert np.all(err <= bar[k]), (name, rec, k, err, bar[k])
    capi.record_steps(False)
