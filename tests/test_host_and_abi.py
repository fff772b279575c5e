"""CPU tests: host-side logic and the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

CONFIGS = {
    'ffjet': dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=1, nmu=1, fmin=3.45e11, fmax=3.45e11,
                  ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[100, 100, 400], uout=0.01, mbh=3.4e9,
                  mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70.),
    'ffjet_formal25': dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=25, nmu=1, fmin=2.41e10,
                           fmax=6.31e14, ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[1024, 1024, 400],
                           uout=0.01, mbh=3.4e9, mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2.,
                           nrscl=70., iname='formal', cflag=0),
    'sphacc': dict(fname='SPHACC', nfreq=1, nmu=1, fmin=2.41e11, fmax=2.41e11, ename='POLSYNCHTH', nvals=4, spin=0.,
                   mbh=10., standard=1, uout=.04, nn=[256, 256, 200], mumin=0.5, mumax=0.5, iname='formal'),
    'thindisk': dict(),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_inputs_in_text_identical_to_reference_writer(name, tmp_path, monkeypatch):
    # golden text = what the reference's grtrans_batch.py wrote for the same kwargs (tools/make_golden_inputs.py)
    from grtrans_amd import grtrans_batch as gb
    monkeypatch.chdir(tmp_path)
    x = gb.grtrans()
    x.write_grtrans_inputs('inputs.in', **CONFIGS[name])
    assert open('inputs.in').read() == open(os.path.join(G, 'inputs_%s.in' % name)).read()
    assert open('files.in').read() == open(os.path.join(G, 'files_%s.in' % name)).read()


def test_input_defaults_and_clamping():
    from grtrans_amd.grtrans_batch import grtrans_inputs
    i = grtrans_inputs(nn=[10, 10, 50], i1=500, i2=5000)
    assert (i.i1, i.i2) == (99, 100)                     # grtrans_batch.py:159-164
    i = grtrans_inputs(ename='MAXCOMP', epotherargs=[1., 2., 3., 4.])
    assert i.nweights == 2 and i.delta == 1. and i.nep == 4
    i = grtrans_inputs(fname='FFJET', jdfile='x.bin')
    assert i.fdfile == 'x.bin'


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "grtrans_amd.h")).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    return sorted(set(re.findall(r'\b(grt_[a-z_0-9]+)\s*\(', h)))


def test_abi_library_exports_every_declared_symbol():
    from grtrans_amd import capi
    lib = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), "libgrtrans_amd.so does not export %s" % s


def test_abi_struct_layouts_match_header():
    from grtrans_amd import capi
    assert C.sizeof(capi.GrtCam) == 10 * 8 + 10 * 4 + 8          # ... + rcut
    assert C.sizeof(capi.GrtEmis) == 11 * 8 + 8 * 4 + 2 * 4 + 8 * 8
    assert C.sizeof(capi.GrtStats) == 3 * 8 + 5 * 8


def test_no_cpu_fallback_without_device():
    # Without a HIP device the product must fail loudly, never compute on the host.
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    from grtrans_amd import capi
    with pytest.raises(capi.GrtError):
        capi.Device(0)
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (8, 8, 50), 0.998, 0.906, 0.01)   # pure host arithmetic
    em = capi.make_emis("POLSYNCHPL", 3.4e9, 2.0, 70.0, 100., 1e5, 3.5)
    with pytest.raises(capi.GrtError):
        capi.trace_tile(cam, 1, 64, "FFJET", em, [3.45e11], "formal")


def test_make_camera_matches_reference_camera_scalars():
    from grtrans_amd import capi
    g = np.load(os.path.join(G, "ffjet_100x100x400.npz"))
    cam = capi.make_camera(1, (-40., 20., -20., 40.), (100, 100, 400), 0.998, 0.906, 0.01)
    u0, uout, offset, mu0, a, kext, nxt, nup = g["cam_scal"]
    assert (cam.u0, cam.uout, cam.offset, cam.mu0, cam.a) == (u0, uout, offset, mu0, a)
    assert (cam.kext, cam.next, cam.nup) == (int(kext), int(nxt), int(nup))


def test_strips_tile_the_image():
    from grtrans_amd.tiling import strip
    for npix in (1024, 1000, 7):
        for world in (1, 2, 3, 8):
            if world > npix:
                continue
            edges = [strip(r, world, npix) for r in range(world)]
            assert edges[0][0] == 1 and edges[-1][1] == npix
            assert all(edges[r][1] + 1 == edges[r + 1][0] for r in range(world - 1))


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def test_two_rank_gloo_gather_equals_single_image(tmp_path):
    # N>1 path on CPU: two processes (gloo), each traces its strip with the CPU oracle standing in for
    # the device, tiles gathered with the same helper bench.py uses; must equal the one-process image exactly.
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from grtrans_amd.tiling import strip, gather_tiles
from oracle import orabind
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nn = (12, 11, 60)
o = orabind.Oracle(nn=nn, iname="formal")
i1, i2 = strip(rank, world, nn[0] * nn[1])
pix, npts, _ = o.trace(i1, i2, [3.45e11], nthreads=1)
full = gather_tiles(torch.from_numpy(pix), dist, 0)
if rank == 0:
    ref, _, _ = o.trace(1, nn[0] * nn[1], [3.45e11], nthreads=1)
    assert np.array_equal(full.numpy(), ref), "gathered image differs"
    open(%r, "w").write("ok")
dist.destroy_process_group()
''' % (ROOT, str(tmp_path / "ok")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env,
                          timeout=600)
    assert (tmp_path / "ok").read_text() == "ok"


def test_fits_grammar_checker_accepts_and_rejects():
    # the independent FITS reader the GPU test holds host/fits_min.f90's output against (tests/fits_check.py)
    import fits_check

    def card(k, v, c=""):
        return ("%-8s= %20s%s" % (k, v, (" / " + c) if c else "")).ljust(80)

    def hdr(cards):
        t = "".join(cards) + "END".ljust(80)
        return (t + " " * ((2880 - len(t) % 2880) % 2880)).encode()

    data = np.arange(6, dtype=">f4").tobytes()
    ext = np.arange(3, dtype=">f4").tobytes()
    good = hdr([card("SIMPLE", "T"), card("BITPIX", -32), card("NAXIS", 1), card("NAXIS1", 6), card("EXTEND", "T")]) + data + b"\0" * (2880 - 24) + \
        hdr(["XTENSION= 'IMAGE   '".ljust(80), card("BITPIX", -32), card("NAXIS", 1), card("NAXIS1", 3), card("PCOUNT", 0), card("GCOUNT", 1),
             card("NX", "3.000000E+00", "# x pixels"), "HIERARCH gridvals(3) = -2.000000E+01 / gridvals(3)".ljust(80),
             "FNAME   = 'FFJET   '           / fname".ljust(80)]) + ext + b"\0" * (2880 - 12)
    h = fits_check.check(good)
    assert len(h) == 2 and np.array_equal(h[0][1], np.arange(6)) and dict((k, v) for k, v, _ in h[1][0])["gridvals(3)"] == -20.0
    assert dict((k, v) for k, v, _ in h[1][0])["FNAME"] == "FFJET"
    for bad in (good[:-1], good.replace(b"BITPIX  =", b"bitpix  =", 1), good.replace(b"END" + b" " * 77, b"END x" + b" " * 75, 1),
                good.replace(b"XTENSION= 'IMAGE   '", b"XTENSION= 'IMAGE'   ", 1), good[:100] + b"\1" + good[101:],
                good.replace(b"PCOUNT  =                    0", b"PCOUNT  =                    1", 1)):
        with pytest.raises(AssertionError):
            fits_check.check(bad)


def test_pixel_share_out_fortran_equals_python():
    # module pgrtrans's pixel_ranges (the share-out of an N-GPU run) against grtrans_amd/tiling.py:ranges, which the 2-rank
    # gloo test and bench.py use: strips and round-robin column tiles, every pixel of i1..i2 exactly once
    from grtrans_amd import capi, tiling
    capi.lib()
    L = C.CDLL(os.path.join(ROOT, "grtrans_amd", "host", "libpgrtrans.so"))

    def f(r, w, i1, i2, nphi, tc):
        n = C.c_int(0); lo = (C.c_int * 4096)(); hi = (C.c_int * 4096)()
        L.pgrtrans_pixel_ranges(r, w, i1, i2, nphi, tc, 4096, C.byref(n), lo, hi)
        return [(lo[k], hi[k]) for k in range(n.value)]
    for (i1, i2, nphi) in ((1, 1024 * 1024, 1024), (1, 100, 10), (37, 911, 13), (5, 5, 7)):
        for w in (1, 2, 3, 8):
            for tc in (0, 1, 4, 32):
                seen = []
                for r in range(w):
                    a, b = f(r, w, i1, i2, nphi, tc), tiling.ranges(r, w, i1, i2, nphi, tc)
                    assert a == b, (i1, i2, nphi, w, tc, r)
                    seen += a
                seen.sort()
                assert seen[0][0] == i1 and seen[-1][1] == i2 and all(seen[k][1] + 1 == seen[k + 1][0] for k in range(len(seen) - 1))


def test_pgrtrans_module_surface_without_a_device():
    # grtrans_amd/pgrtrans.py = the f2py-shaped view of the Fortran module in host/libpgrtrans.so: 85 positional arguments,
    # module arrays, del_pgrtrans_data.  Without a GPU the call must fail loudly (grt_init) -- as a GrtError, not a dead
    # interpreter, and never through a CPU path.
    from grtrans_amd import capi, grtrans_batch as gb
    from grtrans_amd.pgrtrans import pgrtrans, _ARGS
    ref_order = re.findall(r"[a-z_0-9]+", "standard,mumin,mumax,nmu,phi0,spin,uout,uin,rcut,nrotype,gridvals,nn,i1,i2,fname,dt,nt,nload,"
                           "nmdot,mdotmin,mdotmax,ename,mbh,nfreq,fmin,fmax,muval,gmin,gmax,p1,p2,jetalpha,stype,use_geokerr,nvals,iname,"
                           "cflag,extra,debug,outfile,fdfile,fhfile,fgfile,fsim,fnt,findf,fnfiles,fjonfix,fnw,fnfreq_tab,fnr,foffset,fdindf,"
                           "fmagcrit,frspot,fr0spot,fn0spot,ftscl,frscl,fwmin,fwmax,ffmin,ffmax,frmax,fsigt,ffcol,fmdot,fnscl,fnnthscl,fnnthp,"
                           "fbeta,fbl06,fnp,ftp,frin,frout,fthin,fthout,fphiin,fphiout,fscalefac,sigcut,epcoefindx,epotherargs,nepotherargs")
    assert [n for n, _ in _ARGS] == ref_order                       # pgrtrans.f90:12-19
    assert np.shape(pgrtrans.ivals) == ()
    with pytest.raises(TypeError):
        pgrtrans.grtrans_main(1, 2, 3)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: the GPU tests drive this surface end to end")
    x = gb.grtrans()
    with pytest.raises(capi.GrtError):
        x.run_pgrtrans(fname='FFJET', nfreq=1, nmu=1, fmin=3.45e11, fmax=3.45e11, ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1,
                       nn=[8, 8, 40], uout=0.01, mbh=3.4e9, mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], iname='formal')
    assert np.shape(pgrtrans.ivals) == ()


def test_two_rank_gloo_interleaved_column_tiles_equal_single_image(tmp_path):
    # The strong-scaling share-out (bench.py --scaling strong, module pgrtrans with GRT_TILE_COLS): column tiles dealt round-robin
    # (tiling.ranges), every rank traces its ranges, rank-ordered gather, rank 0 un-permutes -- with two gloo ranks on CPU and the
    # oracle standing in for the device (no GPU here); must equal the one-process image exactly, three frequencies.
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from grtrans_amd.tiling import ranges, gather_tiles
from oracle import orabind
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nn = (12, 11, 60)
freqs = [2.3e11, 3.45e11, 6.9e11]
o = orabind.Oracle(nn=nn, iname="delo")
npix = nn[0] * nn[1]
mine = ranges(rank, world, 5, npix - 3, nn[1], 2)          # i1 = 5, i2 = npix - 3: ragged first and last tiles
tiles = [o.trace(a, b, freqs, nthreads=1)[0] for (a, b) in mine]
tile = torch.from_numpy(np.concatenate(tiles, axis=1))
full = gather_tiles(tile, dist, 0)
if rank == 0:
    img = np.zeros((3, npix, 4), dtype=np.float32)
    pre = 0
    for r in range(world):
        for (a, b) in ranges(r, world, 5, npix - 3, nn[1], 2):
            img[:, a - 1:b] = full.numpy()[:, pre:pre + b - a + 1]
            pre += b - a + 1
    ref, _, _ = o.trace(5, npix - 3, freqs, nthreads=1)
    assert pre == npix - 7 and np.array_equal(img[:, 4:npix - 3], ref) and (img[:, :4] == 0).all(), "gathered image differs"
    open(%r, "w").write("ok")
dist.destroy_process_group()
''' % (ROOT, str(tmp_path / "ok")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env,
                          timeout=600)
    assert (tmp_path / "ok").read_text() == "ok"


def test_eight_rank_gloo_share_out_and_unpermutation_at_1024_columns(tmp_path):
    # The first 8-GPU run of bench.py / module pgrtrans in small: 8 gloo ranks on CPU, the 1024-column image shared out in
    # tiles of 32 columns round-robin (tiling.ranges, the default of bench.py --scaling strong), every rank fills its rows with
    # the pixel index standing in for the traced values, ONE rank-ordered gather, rank 0 un-permutes: every pixel exactly once
    # and in its place.  Also the strips (tile_cols = 0) and a ragged i1..i2.
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from grtrans_amd.tiling import ranges, gather_tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nx, ny = 1024, 8
npix = nx * ny
for (i1, i2, tc) in ((1, npix, 32), (1, npix, 0), (19, npix - 77, 32), (1, npix, 128)):
    mine = ranges(rank, world, i1, i2, ny, tc)
    rows = np.concatenate([np.arange(a, b + 1, dtype=np.float32) for (a, b) in mine]) if mine else np.zeros(0, np.float32)
    tile = torch.from_numpy(np.stack([rows, -rows], axis=-1)[None])            # [ncams=1][npix_rank][nvals=2]
    full = gather_tiles(tile, dist, 0)
    if rank == 0:
        img = np.zeros((1, npix, 2), dtype=np.float32)
        pre = 0
        for r in range(world):
            for (a, b) in ranges(r, world, i1, i2, ny, tc):
                img[:, a - 1:b] = full.numpy()[:, pre:pre + b - a + 1]
                pre += b - a + 1
        want = np.zeros(npix, dtype=np.float32)
        want[i1 - 1:i2] = np.arange(i1, i2 + 1, dtype=np.float32)
        assert pre == i2 - i1 + 1 and np.array_equal(img[0, :, 0], want) and np.array_equal(img[0, :, 1], -want), (i1, i2, tc)
        if tc == 32 and i1 == 1:
            sizes = [sum(b - a + 1 for a, b in ranges(r, world, i1, i2, ny, tc)) for r in range(world)]
            assert len(set(sizes)) == 1 and len(ranges(0, world, i1, i2, ny, tc)) == 4      # 32 tiles over 8 ranks
if rank == 0:
    open(%r, "w").write("ok")
dist.destroy_process_group()
''' % (ROOT, str(tmp_path / "ok")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env,
                          timeout=900)
    assert (tmp_path / "ok").read_text() == "ok"


def test_bench_never_reports_n_gpus_from_one_process():
    # VERDICT r02 weak #4a: `python bench.py --gpus N` without a launcher must start N ranks itself or fail -- never print
    # n_gpus = N for one GPU's work.  (1) WORLD_SIZE that contradicts --gpus: refused, exit code 2, no JSON line.
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # (2) no launcher at all: the parent starts two ranks through torch.distributed.run and hands their exit code on.  Without a
    # GPU the ranks fail (there is no CPU path): non-zero exit and still no line; on a GPU box the driver's own N > 1 run covers it.
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the self-launch would run a real 2-rank bench here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GRT_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nn", "16", "16", "40",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr or "nproc" in out.stderr.lower()


def test_comm_rendezvous_file_of_another_run_is_not_joined(tmp_path):
    # a rendezvous file left by a crashed run must not be joined.  Rank 1 of 2 polls a path that holds (a) a
    # 128-byte id in the old format, (b) a well-formed record with ANOTHER run's nonce: both are ignored and the call returns
    # GRT_E_IO after its timeout instead of calling ncclCommInitRank with a dead id (no GPU needed: the read comes first).
    from grtrans_amd import capi
    L = capi.lib()
    L.grt_comm_init_file.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_double]
    path = tmp_path / "comm"
    os.environ["GRT_COMM_NONCE"] = "this-run"
    try:
        path.write_bytes(b"\x01" * 128)
        assert L.grt_comm_init_file(1, 2, str(path).encode(), 0.3) == -6
        path.write_bytes(b"GRTCOMM2" + b"other-run".ljust(64, b"\0") + b"\x01" * 128)
        assert L.grt_comm_init_file(1, 2, str(path).encode(), 0.3) == -6
        assert L.grt_comm_init_file(1, 2, b"", 0.3) == -3 and L.grt_comm_init_file(2, 2, str(path).encode(), 0.3) == -3
    finally:
        del os.environ["GRT_COMM_NONCE"]


def test_usable_cpus_respects_affinity_and_quota(monkeypatch):
    # bench.py's cpu_baseline thread count: the affinity mask capped by the cgroup quota, never os.cpu_count()
    sys.path.insert(0, ROOT)
    import bench
    n, quota = bench.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    if quota is not None:
        assert n <= max(1, int(quota))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1, 2})
    assert bench.usable_cpus()[0] <= 3
    assert len(bench.build_id()) == 16 and bench.build_id() == bench.build_id()


def test_code_object_holds_no_device_side_calls(tmp_path):
    # r03: a real function call inside grt_setup_kernel (s_swappc to setup_ray_core, 390 KB the inliner had left outlined) faulted
    # on the MI355X as soon as one more value was live across it; with -enable-ipra=false the same code raised an aperture
    # violation.  The large routines are now inlined by attribute (GRT_ALWAYS_INLINE) and the gfx950 code object must stay free
    # of calls: a new outlined routine shows up here before it shows up as a device fault.
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "grtrans_amd", "csrc", "grt_kernels.o")
    if not (os.path.exists(obj) and os.path.exists(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("no object file / llvm tools here")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
    subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-s", co], capture_output=True, text=True, check=True).stdout
    funcs = [l.split()[-1] for l in syms.splitlines() if " FUNC " in l and not l.rstrip().endswith(".kd")]
    outlined = [f for f in funcs if "grt_" not in f or "kernel" not in f]
    assert not outlined, outlined
    dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
    assert dis.count("s_swappc_b64") == 0      # (s_setpc_b64 alone is the long-branch idiom of these 100-KB kernels, not a call)



def test_no_kernel_parks_values_in_agprs():
    # r03: two transfer-kernel instantiations compiled for one wave per SIMD (256 VGPRs + values parked in AGPRs) returned garbage on
    # the MI355X -- KORAL3D DELO batches, and the IHARM hybrid kernel with 300-350 zero pixels that moved from run to run -- while the
    # same sources at two waves (256 registers, scratch instead of AGPRs) were correct.  Every kernel is now compiled for two waves;
    # the compiler's own resource remarks of the shipped object must show no AGPR anywhere.
    usage = os.path.join(ROOT, "grtrans_amd", "csrc", "kernel_resource_usage.txt")
    if not os.path.exists(usage):
        pytest.skip("no resource-usage remarks here (the library was not built in this tree)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_usage
    k = kernel_usage.parse(open(usage).read())
    assert len(k) > 100
    assert all("AGPRs" in v and "Occupancy" in v and "VGPRs" in v for v in k.values()), "resource remarks not understood"
    bad = {n: v for n, v in k.items() if int(v["AGPRs"]) != 0 or int(v["Occupancy"]) < 2}
    assert not bad, sorted(bad)[:5]


def test_no_live_range_copy_under_a_partial_exec_mask():
    # r03: in grt_transfer_kernel<FORMAL, IHARM, HYBRIDTHPL, 1, false> hipcc's register allocator saved two loop-carried accumulators
    # (v_mov v[124:125], v[220:221]) in the FLOW block of a divergent if-else BEFORE the instruction that flips the exec mask, i.e.
    # for the lanes of the `then` side only; the restore ran for every lane, so the lanes of the `else` side got another variable's
    # value.  Pixels that must be zeroed by the fac gate came out NaN (or, one build earlier, zero pixels moved from run to run).
    # The divergent branch is gone from the source; this scans the assembly of the shipped object for the signature (a register-to-
    # register copy before the exec flip whose source lives through the if-else, whose destination the else side never writes and
    # which is undone later) so that the next occurrence is a failing CPU test, not a wrong image.
    asm = os.path.join(ROOT, "grtrans_amd", "csrc", "grt_kernels.gfx950.s")
    if not os.path.exists(asm):
        pytest.skip("no device assembly here (the library was not built in this tree)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_flow_copies
    sites = check_flow_copies.scan(asm)
    assert not sites, sites[:3]


def test_the_flow_copy_scanner_sees_the_defect_and_not_the_ordinary_merge():
    # positive and negative control of tools/check_flow_copies.py on two hand-reduced shapes of compiler output (tests/golden/flow_copy_*.s):
    # the save / restore pair around a divergent if-else that cost r03 an image, and the then side's ordinary copy into a merge register
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_flow_copies
    bad = check_flow_copies.scan(os.path.join(ROOT, "tests", "golden", "flow_copy_bad.s"))
    assert len(bad) == 1 and bad[0][0] == "kernel_with_the_defect" and len(bad[0][2]) == 2
    assert check_flow_copies.scan(os.path.join(ROOT, "tests", "golden", "flow_copy_ok.s")) == []
    # r04: the same placement of a SPILL (scratch_store of a value that lives through the if-else, reloaded from the slot under the full
    # mask) is seen; the then side's own spill into a merge slot and an SGPR spill (v_writelane ignores the exec mask) are not flagged
    bad = check_flow_copies.scan(os.path.join(ROOT, "tests", "golden", "flow_spill_bad.s"))
    assert len(bad) == 1 and bad[0][0] == "kernel_with_the_spill_defect" and bad[0][2][0].startswith("scratch_store_dwordx2")
    assert check_flow_copies.scan(os.path.join(ROOT, "tests", "golden", "flow_spill_ok.s")) == []


def test_harness_unit_conversions_and_polarization_fractions():
    # calc_freqs / convert_to_lum / convert_to_Jy and the lp, cp, lpf, cpf of calc_spec (grtrans_batch.py:490-564): the formulas of the
    # reference's harness on a small synthetic image in the file reader's layout
    from grtrans_amd import grtrans_batch as gb
    x = gb.grtrans()
    x.inputs = gb.grtrans_inputs(fmin=1e10, fmax=1e12, nfreq=5, mbh=4e6)
    x.calc_freqs(5)
    assert np.allclose(x.freqs, 1e10 * 100.0 ** (np.arange(5) / 4.0), rtol=1e-14) and x.freqs[-1] == pytest.approx(1e12, rel=1e-14)
    x.calc_freqs(1)
    assert np.array_equal(x.freqs, [1e10])
    # a RECTANGULAR image: pixels run beta-fastest (camera.f90: ny values of beta per alpha column), so the alpha step is
    # ab[ny,0] - ab[0,0].  The reference's harness takes ab[self.nx,0] - ab[0,0] (grtrans_batch.py:503,548,559) -- the same number for the
    # square images its test problems use, a latent slip otherwise; this harness uses the stride the pixel order defines
    nx, ny = 7, 5
    rng = np.random.default_rng(3)
    a, b = np.meshgrid(np.linspace(-5, 5, nx), np.linspace(-4, 4, ny), indexing="ij")
    x.ab = np.stack([a.ravel(), b.ravel()], axis=1)
    x.ivals = rng.uniform(0.1, 1.0, (nx * ny, 4, 2)) * np.array([1.0, 0.3, -0.2, 0.05])[None, :, None]
    x.nx, x.ny, x.nvals = nx, ny, 4
    x.calc_spec(2)
    da, db = 10.0 / (nx - 1), 8.0 / (ny - 1)
    assert x.da == pytest.approx(da) and x.db == pytest.approx(db)
    spec = x.ivals.sum(axis=0) * da * db
    assert np.allclose(x.spec, spec)
    assert np.allclose(x.lp, np.hypot(spec[1], spec[2]) / spec[0]) and np.allclose(x.cp, spec[3] / spec[0])
    assert np.allclose(x.lpf, np.hypot(x.ivals[:, 1], x.ivals[:, 2]).sum(axis=0) * da * db / spec[0])
    assert np.allclose(x.cpf, np.abs(x.ivals[:, 3]).sum(axis=0) * da * db / spec[0])
    lbh = 6.67259e-8 * 1.99e33 * 4e6 / 8.98755179e20
    iv0, sp0 = x.ivals.copy(), x.spec.copy()
    x.convert_to_lum()
    assert np.allclose(x.spec, sp0 * 4 * np.pi * lbh ** 2) and np.allclose(x.ivals, iv0 * 4 * np.pi * lbh ** 2 * da * db)
    x.ivals, x.spec = iv0.copy(), sp0.copy()
    D = 8.1 * 3.086e21
    x.convert_to_Jy(D)
    assert np.allclose(x.spec, sp0 * lbh ** 2 / D ** 2 * 1e23) and np.allclose(x.ivals, iv0 * lbh ** 2 / D ** 2 * 1e23 * da * db)
    # the pgrtrans layout (ivals[stokes, pixel, camera], ab[2, pixel]) gives the same fractions
    y = gb.grtrans()
    y.inputs = x.inputs
    y.ab, y.ivals, y.nx, y.ny, y.nvals = x.ab.T.copy(), np.transpose(iv0, (1, 0, 2)).copy(), nx, ny, 4
    y.calc_spec_pgrtrans(2)
    assert np.allclose(y.spec, sp0) and np.allclose(y.lp, np.hypot(sp0[1], sp0[2]) / sp0[0]) and np.allclose(y.cpf, np.abs(iv0[:, 3]).sum(axis=0) * da * db / sp0[0])


def test_bench_self_launch_builds_the_exact_launcher_command(monkeypatch):
    # `python bench.py --gpus 8` without a launcher starts its ranks through exactly the command line the driver
    # uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <its own
    # arguments>), with dmabuf IPC selected, and ends with the launcher's exit code
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)

    class A:
        gpus = 8
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.self_launch(A, argv) == 7
    cmd = seen["cmd"]
    port = cmd[cmd.index("--master-port") + 1]
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", port, os.path.join(ROOT, "bench.py")] + argv
    assert 1024 <= int(port) <= 65535 and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_rank_that_exits_98_or_3_ends_the_launched_job_nonzero_in_bounded_time(tmp_path):
    # the library's communicator watchdog ends a rank with exit code 98, bench.py's gather watchdog with 3: the launcher bench.py relies
    # on must then end the WHOLE job non-zero promptly -- not leave the other ranks waiting in a collective.  Two ranks on CPU: rank 1
    # exits with the code after a second, rank 0 would sleep for two minutes.
    import subprocess
    import time
    helper = tmp_path / "rank.py"
    helper.write_text("import os, sys, time\n"
                      "code = int(sys.argv[1])\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    time.sleep(1.0); os._exit(code)\n"
                      "time.sleep(120)\n")
    for code in (98, 3):
        port = _free_port()
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(helper), str(code)], capture_output=True, text=True, timeout=100)
        assert r.returncode != 0, r.stdout + r.stderr
        assert time.time() - t0 < 60.0


def test_differential_comparer_sees_a_seeded_fault_and_not_a_last_bit(tmp_path):
    # negative / positive control of tools/diff_compare.py (the comparer of the differential-compilation GPU test): one stored-point
    # count off by one, one turning-point count off by one, one record value 1e-9 of its peak off, one image value 1e-6 off and a
    # NaN that moved are each reported; one fp32 image value moved by one unit in the last place is not.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import diff_compare
    rng = np.random.default_rng(5)
    base = {"log": np.array(["cam 0 test"]), "c0_status": rng.integers(40, 400, 144).astype(np.int32),
            "c0_pix": rng.normal(size=(2, 144, 4)).astype(np.float32), "c0_rec": rng.normal(size=(8, 144, 60)),
            "c0_gk_tpmi": rng.integers(0, 3, (144, 60)).astype(np.int32), "c0_gk_lam": rng.normal(size=(144, 60))}
    base["c0_pix"][0, 7] = np.nan

    def run(mut):
        other = {k: v.copy() for k, v in base.items()}
        mut(other)
        np.savez(tmp_path / "a.npz", **base); np.savez(tmp_path / "b.npz", **other)
        return diff_compare.compare(np.load(tmp_path / "a.npz"), np.load(tmp_path / "b.npz"))[0]

    assert run(lambda o: None) == []

    def ulp(o):
        v = o["c0_pix"].view(np.uint32); v[1, 3, 2] += 1
    assert run(ulp) == []

    def st(o): o["c0_status"][5] += 1
    def tp(o): o["c0_gk_tpmi"][3, 9] += 1
    def rec(o): o["c0_rec"][4, 10, 10] += 1e-9 * np.abs(o["c0_rec"][4]).max()
    def img(o): o["c0_pix"][0, 20, 1] += np.float32(1e-6) * np.abs(o["c0_pix"][0, :, 1][np.isfinite(o["c0_pix"][0, :, 1])]).max()
    def nan(o): o["c0_pix"][0, 7] = 0.0; o["c0_pix"][0, 8] = np.nan
    for m in (st, tp, rec, img, nan):
        assert len(run(m)) == 1, m.__name__


def test_makefile_keeps_the_required_flags_when_hipflags_is_overridden():
    # correctness of the shipped object rests on -mllvm -disable-machine-licm (and on -ffp-contract=fast-honor-pragmas for the
    # REAL(4) islands); a user overriding HIPFLAGS must not lose them silently.  `make -n` with HIPFLAGS replaced: every device compile
    # line still carries both.
    out = subprocess.run(["make", "-n", "-B", "-C", os.path.join(ROOT, "grtrans_amd", "csrc"), "HIPFLAGS=--offload-arch=gfx950 -O1 -I.", "grt_kernels.o", "grt_api.o", "grt_gather.o"],
                         capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if "hipcc" in l and " -c " in l]
    assert len(lines) >= 3
    for l in lines:
        assert "-disable-machine-licm" in l and "-ffp-contract=fast-honor-pragmas" in l, l
