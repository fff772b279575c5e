#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-pixel path (BASELINE.json metric).

A "step" = one pass of the hot path (set-up kernel + trace kernel) over one camera tile of
synthetic-free, fully deterministic input: FFJET (BL09 M87 jet table shipped in grtrans_amd/data)
POLSYNCHPL nvals=4 spin=0.998 nn=[1024,1024,400] nfreq=1, formal integrator -- the configuration
the metric "ray-steps/sec + HBM GB/s" is quoted on.  Inputs (fluid table, coefficient tables) are
resident in HBM before the timed region; outputs stay in HBM (16 B/ray D2H is outside).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).
  --scaling strong (default): BASELINE's metric -- the SAME nn=[1024,1024,400] camera shared out over the N ranks in
      tiles of --tile-cols image columns dealt round-robin (grtrans_amd/tiling.py:ranges, the share-out
      module pgrtrans uses); `value` = ray-steps of the whole image / max-over-ranks time.
  --scaling weak: the camera is nn=[1024*N,1024,400] over the same field of view and rank r traces the
      contiguous pixel strip i1..i2 of 1024*1024 rays (the reference's own i1,i2 partition key).
Either way the only exchange is one gather of the ranks' fp32 cameras to rank 0 over xGMI after the timed steps,
through the library's own RCCL communicator (grt_stage_* / grt_gather_stage, include/grtrans_amd.h; the ncclUniqueId
travels over torch.distributed) with torch.distributed.gather as the fall-back; `gather_ms` is reported.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FREQ = 3.45e11
# headline workload first; the others are BASELINE.json's smaller configs, selectable for profiling only
WORKLOADS = {
    "ffjet": dict(fname="FFJET", ename="POLSYNCHPL", spin=0.998, mu0=0.906, uout=0.01, gridvals=(-40., 20., -20., 40.),
                  mbh=3.4e9, nfac=2.0, bfac=70.0, freq=3.45e11),
    "sphacc": dict(fname="SPHACC", ename="POLSYNCHTH", spin=0.0, mu0=0.5, uout=0.04, gridvals=(-25., 25., -25., 25.),
                   mbh=10.0, nfac=1.0, bfac=6.0, freq=2.41e11),
    # synthetic 256x256-zone HARM-format dump written by tools/make_harm_dump.py (seed 20261002) at start-up
    "harm": dict(fname="HARM", ename="POLSYNCHTH", spin=0.9375, mu0=0.6428, uout=0.04, gridvals=(-13., 13., -13., 13.),
                 mbh=4e6, nfac=1.0, bfac=6.0, freq=2.3e11, gmin=1.0, mdot=1.57e15),
}
BYTES_PER_STEP = 64.0          # one fp64 SoA record (r,theta,phi,lambda,I,Q,U,V) per ray-step
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# fp64 vector (non-matrix) peak.  Source: AMD's MI355X product brief lists 78.6 TFLOP/s "peak double precision vector (FP64)"; the same figure
# follows from the guide's numbers (/opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD16 at 2.4 GHz, fp64 FMA at full rate):
# 256 x 64 lanes x 2 flop x 2.4e9 = 78.6e12 = half the 157.3 TFLOP/s fp32 vector figure the guide states (packed fp32).  The guide itself
# does not print the fp64 number.
FP64_VALU_PEAK_TFLOPS = 78.6
TILE_COLS_DEFAULT = 1          # strong scaling share-out: single image columns round-robin (the generic ray class sits in a few columns)


def build_id():
    """Identity of the device code the PMC summaries under profiles/ were collected with: sha256 over the sources of
    grtrans_amd/csrc (the library is built from exactly these by its Makefile)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "grtrans_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.h")) +
                    glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def usable_cpus():
    """Host cores this process may actually run on: the scheduler affinity mask, capped by the cgroup CPU quota
    (v2 cpu.max, v1 cfs_quota_us / cfs_period_us).  os.cpu_count() ignores both; r02's driver run started 256
    OpenMP threads on a box that gave it far fewer and measured thread thrash instead of the reference."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n), quota


def cpu_baseline(nx, ny, nup, budget_s=18.0):
    """Reference Fortran (oracle/_ref) -- or the C++ port if that library is absent -- on a bounded
    contiguous pixel sample around the image centre of the SAME camera.  Runs in a child
    process: the reference writes banners to Fortran unit 6, which must not reach this process's stdout
    (one JSON line only); idle OpenMP threads sleep (OMP_WAIT_POLICY=passive) so that a host shared with other
    jobs is not measured through spinning threads."""
    import subprocess
    import tempfile
    env = dict(os.environ)
    env.setdefault("OMP_WAIT_POLICY", "passive")
    env.setdefault("OMP_PROC_BIND", "false")
    env.pop("OMP_NUM_THREADS", None)
    with tempfile.NamedTemporaryFile(suffix=".json") as tf:
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "json.dump(bench._cpu_baseline(%d, %d, %d, %f), open(%r, 'w'))" % (ROOT, nx, ny, nup, budget_s, tf.name))
        subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, env=env)
        return json.load(open(tf.name))


def _cpu_baseline(nx, ny, nup, budget_s):
    threads, quota = usable_cpus()
    kw = dict(nn=(nx, ny, nup), iname="formal")
    try:
        from oracle import refbind
        if not refbind.available():
            raise RuntimeError("no _ref")
        eng, kind = refbind.Ref(**kw), "reference"
        kind_note = "reference modules (unmodified Fortran, oracle/_ref), pixel loop of oracle/ref_driver.f90:ref_trace"
    except Exception:
        from oracle import orabind
        eng, kind = orabind.Oracle(**kw), "port"
        kind_note = "C++ restatement oracle/grto_*.hpp (oracle/_ref absent)"
    centre = (nx // 2) * ny + 1
    room = nx * ny - centre
    # untimed warm-up: OpenMP thread start, first touch of the tables and of every thread's work arrays
    t0 = time.perf_counter()
    eng.trace(centre, centre + min(max(threads * 4, 64), room) - 1, [FREQ], nthreads=threads)
    first_call_s = time.perf_counter() - t0
    # sample: at least 32 rays per thread, grown until ONE timed call costs about a fifth of the budget
    nrays = min(max(threads * 32, 256), room)
    target = budget_s / 5.0
    while True:
        _, npts, sec = eng.trace(centre, centre + nrays - 1, [FREQ], nthreads=threads)
        if sec >= 0.6 * target or nrays >= min(400000, room):
            break
        nrays = int(min(max(nrays * min(target / max(sec, 1e-3), 16.0), nrays * 2), 400000, room))
    steps = int(np.abs(npts).sum())
    rates = []
    for _ in range(3):
        _, npts, sec = eng.trace(centre, centre + nrays - 1, [FREQ], nthreads=threads)
        rates.append(int(np.abs(npts).sum()) / sec)
    med = sorted(rates)[1]
    # one thread on a small sample: what one core does alone (parallel efficiency of the figure above = value / threads / this)
    n1 = min(96, room)
    _, npts1, sec1 = eng.trace(centre, centre + n1 - 1, [FREQ], nthreads=1)
    single = int(np.abs(npts1).sum()) / sec1
    return {"value": med, "unit": "ray-steps/s", "cores": threads, "kind": kind, "kind_note": kind_note,
            "threads": threads, "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
            "per_thread": med / threads, "single_thread": single, "first_call_s": first_call_s, "timed_calls": rates,
            "sample": "%d contiguous pixels from the image centre of the same %dx%dx%d FFJET camera (%d ray-steps per call); one "
                      "untimed warm-up call, then median of 3 timed calls; OpenMP schedule(static,1), %d threads = affinity mask "
                      "capped by the cgroup quota" % (nrays, nx, ny, nup, steps, threads)}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process never touches a GPU; it starts the N ranks
    (torch.distributed.run, one per GPU, RCCL) as a child and ends with the child's exit code.  One process must never print
    n_gpus = N for one GPU's work."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nn", type=int, nargs=3, default=[1024, 1024, 400])
    ap.add_argument("--iname", default="formal")
    ap.add_argument("--workload", default="ffjet", choices=sorted(WORKLOADS))
    ap.add_argument("--nfreq", type=int, default=1, help="BASELINE config 5 uses 25 log-spaced frequencies 2.41e10..6.31e14")
    ap.add_argument("--no-steps-record", action="store_true", help="do not write the per-step SoA record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # BASELINE.json's metric is the SAME nn=[1024,1024,400] camera at 1/2/4/8 GPUs: strong scaling is the default
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--tile-cols", type=int, default=TILE_COLS_DEFAULT,
                    help="strong scaling: image columns per tile (round-robin over the ranks); 0 = contiguous strips")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("GRT_BENCH_BACKEND") is None:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report one for the other\n" % (args.gpus, world))
        sys.exit(2)
    n_gpus = world
    dist = None
    import torch
    # test hooks (single-GPU boxes): GRT_BENCH_BACKEND=gloo runs the N>1 code path with CPU collectives,
    # GRT_BENCH_DEVICE pins every rank to one HIP ordinal.  The driver's runs use neither (nccl = RCCL, one GPU per rank).
    backend = os.environ.get("GRT_BENCH_BACKEND", "nccl")
    if "GRT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["GRT_BENCH_DEVICE"])
    cdev = "cuda" if backend == "nccl" else "cpu"
    # GRT_BENCH_FORCE_DIST=1 (test hook): initialise the process group even for one rank, so that a 1-GPU box runs the
    # barrier / gather / all_reduce of the N>1 path through RCCL
    if world > 1 or os.environ.get("GRT_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    from grtrans_amd import capi

    nx, ny, nup = args.nn
    strong = (args.scaling == "strong")
    nx_total = nx if strong else nx * world    # weak scaling: camera widens with the rank count
    dev = capi.Device(local_rank)
    W = WORKLOADS[args.workload]
    dev.load_ffjet()
    dev.load_sphacc()
    dev.load_tables()
    if args.workload == "harm":
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_harm_dump
        hd = tempfile.mkdtemp(prefix="grt_harm_")
        make_harm_dump.write(hd, n=256)
        dev.load_harm(os.path.join(hd, "dump"), os.path.join(hd, "dump040"), 40, W["spin"])
    cam = capi.make_camera(1, W["gridvals"], (nx_total, ny, nup), W["spin"], W["mu0"], W["uout"], -0.5)
    em = capi.make_emis(W["ename"], W["mbh"], W["nfac"], W["bfac"], W.get("gmin", 100.), 1e5, 3.5, mdot=W.get("mdot", 0.0))
    from grtrans_amd.tiling import ranges as pixel_ranges
    if strong:
        tcols = args.tile_cols if world > 1 else 0
        all_ranges = [pixel_ranges(r, world, 1, nx * ny, ny, tcols) for r in range(world)]
    else:
        all_ranges = [[(r * nx * ny + 1, (r + 1) * nx * ny)] for r in range(world)]
    my_ranges = all_ranges[rank]
    npix_of_rank = [sum(b - a + 1 for a, b in rr) for rr in all_ranges]
    capi.record_steps(not args.no_steps_record)
    capi.geo_reuse(False)      # every timed step is the whole hot path: set-up + geodesics + transfer (no resident-record re-use)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    freqs = [W["freq"]] if args.nfreq == 1 else list(2.41e10 * np.exp(np.arange(args.nfreq) * np.log(6.31e14 / 2.41e10) / (args.nfreq - 1)))

    import ctypes as C

    def one_step(stage=False):
        # one step = every pixel range of this rank in ONE grt_trace_ranges call (one set of launches: a rank's column tiles
        # together fill the GPU, tile by tile they would not)
        _, _, _, st = capi.trace_ranges(cam, my_ranges, W["fname"], em, freqs, args.iname, resident=True)
        if stage:    # device-to-device into the rank-level image the gather sends
            capi.check(capi.lib().grt_stage_resident(C.c_int(0), C.c_int(args.nfreq), C.c_int64(npix_of_rank[rank]), C.c_int64(0)),
                       "grt_stage_resident")
        return st

    # secondary figure: image-only mode (no per-step record, azimuth integrals skipped), untimed region
    image_only = None
    if not args.no_steps_record:
        capi.record_steps(False)
        one_step()
        st0 = one_step()
        image_only = {"ray_steps_per_s": st0.ray_steps / (st0.ms_total / 1e3), "ms_geo": st0.ms_geo,
                      "ms_transfer": st0.ms_transfer}
        capi.record_steps(True)
    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    ray_steps = 0
    ms_trace = 0.0
    ms_setup = 0.0
    ms_geo = 0.0
    ms_xfer = 0.0
    for _ in range(args.steps):
        st = one_step()
        ray_steps += st.ray_steps
        ms_trace += st.ms_trace
        ms_setup += st.ms_setup
        ms_geo += st.ms_geo
        ms_xfer += st.ms_transfer
    barrier()
    dt = time.perf_counter() - t0
    gather_ms, gather_via = None, None
    ray_steps_all = ray_steps
    per_rank_ms = None
    if dist is not None:       # the timing of the job: max over ranks of the bracketed wall time, sum of the ranks' ray-steps
        tmax = torch.tensor([dt, float(ray_steps), ms_trace], dtype=torch.float64, device=cdev)
        tsum = tmax.clone()
        mine = torch.tensor([dt / args.steps * 1e3, float(ray_steps // args.steps)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [[float(e[0]), int(e[1].item())] for e in every]     # load balance as measured: [ms per step, ray-steps] per rank
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        ray_steps_all = int(tsum[1].item())
        ms_trace = float(tmax[2])
    # The image assembly below (one gather, after the timed steps) runs with more than one rank on the driver's node only: if it
    # does not finish, rank 0 prints the line without the gather timing and EVERY rank exits non-zero (a hung gather is a failure).
    import threading
    emitted = threading.Lock()
    state = {"emit": None}
    gather_timeout = float(os.environ.get("GRT_BENCH_GATHER_TIMEOUT", "240"))

    def on_timeout():
        if emitted.acquire(blocking=False):
            if rank == 0 and state["emit"] is not None:
                state["emit"](None, "FAILED: gather did not finish within %g s (line printed without it, exit code 3)" % gather_timeout)
                sys.stdout.flush()
            os._exit(3)
    watchdog = threading.Timer(gather_timeout, on_timeout) if dist is not None else None
    # the library's ncclCommInitRank watchdog (exit code 98) fires first: a rank whose peer never joins then fails inside
    # grt_comm_init and takes the agreed fall-back instead of sitting in it until the gather watchdog above ends the job
    os.environ.setdefault("GRT_COMM_INIT_TIMEOUT", "%g" % max(10.0, 0.5 * gather_timeout))

    def emit(gather_ms, gather_via):
        value = ray_steps_all / dt
        # dominant kernel = grt_geo_kernel (fast class); its launch + the transfer launch process the same
        # ray-steps, so the algorithmic 64 B/step is priced over the sum of the two (HIP events on the library stream)
        t_kernel = ms_trace / args.steps / 1e3
        achieved = BYTES_PER_STEP * (ray_steps / args.steps) / t_kernel / 1e9
        traffic = None
        valu = None
        stale = None
        bid = build_id()
        # the PMC passes were collected on the default workload only (tools/profile_round.sh -> profiles/traffic.json, valu.json)
        same_as_profiled = (args.workload == "ffjet" and [nx, ny, nup] == [1024, 1024, 400] and args.nfreq == 1
                            and args.iname == "formal" and not args.no_steps_record and world == 1)
        if same_as_profiled:
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                vj = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
            except Exception:
                tj = vj = None
            # the counters are only as good as the code they were counted on: a library built from other sources than the ones
            # the PMC passes ran is priced with NOTHING rather than with stale counts
            if tj is not None and (tj.get("build_id") != bid or vj.get("build_id") != bid):
                stale = {"stale": True, "profiled_build_id": vj.get("build_id"), "this_build_id": bid,
                         "note": "profiles/valu.json / traffic.json were collected with other device sources: re-run tools/profile_round.sh"}
                tj = vj = None
            if tj is not None:
                traffic = tj.get("hbm_bytes_per_launch")
                v = vj
                # fp64 VALU roofline (the bound that matters): instruction counts per launch from the committed
                # SQ-counter summary (same workload, same device sources), priced over THIS run's HIP-event kernel time
                flop = 2.0 * v["fma_f64_wave_insts"] * 64 + (v["mul_f64_wave_insts"] + v["add_f64_wave_insts"]) * 64.0
                valu = {"bound": "fp64 vector ALU", "fma_f64": v["fma_f64_wave_insts"], "mul_f64": v["mul_f64_wave_insts"],
                        "add_f64": v["add_f64_wave_insts"], "trans_f64": v["trans_f64_wave_insts"], "valu_wave_insts": v["valu_wave_insts"],
                        "counts": "wave64 instructions per trace phase (geo + transfer launches of both classes), %s, build %s" % (v["source"], bid),
                        "flop_per_launch": flop, "achieved": flop / t_kernel / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flop / t_kernel / 1e12 / FP64_VALU_PEAK_TFLOPS,
                        # issue slots: 4 cycles per wave64 fp64 instruction (16 for the quarter-rate seeds) on each of 4 SIMDs x 256 CUs
                        "valu_issue_utilisation": (4.0 * (v["valu_wave_insts"] - v["trans_f64_wave_insts"]) + 16.0 * v["trans_f64_wave_insts"]) /
                                                  (t_kernel * v["sclk_hz"] * 4 * 256)}
        mode = ("strong: the whole image shared out over the GPUs" if world > 1 else "one GPU traces the whole image") if strong else "weak: this camera per GPU"
        out = {
            "metric": "ray-steps/sec", "value": value, "unit": "ray-steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "deterministic: BL09 FFJET table shipped with the reference (no RNG in the path)",
            "config": {"workload": "%s %s nvals=4 spin=%g mu0=%g nn=[%d,%d,%d] (%s), nfreq=%d (%g Hz), iname=%s, "
                                   "gridvals=%s, uout=%g" % (W["fname"], W["ename"], W["spin"], W["mu0"], nx, ny, nup, mode, args.nfreq,
                                                             freqs[0], args.iname, list(W["gridvals"]), W["uout"]),
                       "pixels_per_gpu": npix_of_rank[0], "ray_steps_per_step_per_gpu": ray_steps // args.steps,
                       "parallelism": (("tiles of %d image columns round-robin over the GPUs" % args.tile_cols if args.tile_cols > 0 else "contiguous pixel strips")
                                       if strong and world > 1 else "pixel strips i1..i2 per GPU") +
                                      ", 1 process/GPU, one grt_trace_ranges call per step and rank, one RCCL gather of the fp32 cameras",
                       "per_step_record_written": not args.no_steps_record, "nfreq": args.nfreq,
                       "ray_step_frequencies_per_s": value * args.nfreq, "per_rank_ms_and_ray_steps": per_rank_ms,
                       "gather_ms": gather_ms, "gather_via": gather_via, "setup_kernel_ms": ms_setup / args.steps,
                       "geo_kernel_ms": ms_geo / args.steps, "transfer_kernel_ms": ms_xfer / args.steps,
                       "kernel_ms_note": "geo_kernel_ms: HIP events around the fast ray class's geodesic launches (image-only kernel, and with the "
                                         "record its in-line azimuth, grt_phi_kernel on the plunging rays and the left-overs behind it); transfer_kernel_ms: the rest of the trace phase",
                       "image_only_mode": image_only, "build_id": bid},
            "roofline": {"bound": "hbm", "kernel": "grt_geo_kernel+grt_transfer_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_ratio": (traffic / (BYTES_PER_STEP * (ray_steps / args.steps))) if traffic else None,
                         "valu": valu,
                         "note": "algorithmic 64 B per ray-step (SURVEY 8d) over the trace kernels' HIP-event time (rank 0's launches; "
                                 "N > 1: max over ranks); the path is fp64-VALU issue bound, not HBM bound: see roofline.valu and DESIGN.md 5"},
        }
        if valu is not None:
            # The bound that matters is named first: the dominant kernels are fp64-VALU bound, so the roofline's
            # own achieved / peak / frac are the fp64 vector figures; the HBM reading (64 B per ray-step against 8 TB/s) stays beside it.
            hbm = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "traffic_ratio")}
            out["roofline"].update({"bound": "fp64_valu", "achieved": valu["achieved"], "peak": valu["peak"], "unit": "TFLOP/s",
                                    "frac": valu["frac"], "hbm": hbm})
        if stale is not None:
            out["roofline"].update(stale)
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == "ffjet":
            out["cpu_baseline"] = cpu_baseline(nx, ny, nup)
        print(json.dumps(out))
    state["emit"] = emit
    if watchdog is not None:
        watchdog.daemon = True
        watchdog.start()
    gather_failed = False
    try:
        if dist is not None:
            # assemble ivals on rank 0 (pgrtrans.f90:221-242): ONE gather of the ranks' fp32 cameras over xGMI, after the timed steps
            L = capi.lib()
            nv = 4

            def agree(ok):
                """every rank takes the same path: min over the ranks of this rank's success flag"""
                t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item() > 0.5)

            capi.check(L.grt_stage_reset(C.c_int(args.nfreq), C.c_int(nv), C.c_int64(npix_of_rank[rank])), "grt_stage_reset")
            one_step(stage=True)                                    # untimed: fills the rank-level image (same bits as the timed steps)
            torch.cuda.synchronize()
            full = None
            use_lib = (cdev == "cuda" and os.environ.get("GRT_BENCH_GATHER") != "torch")
            why = "torch gather requested"
            if use_lib:
                # the library's own RCCL communicator; its ncclUniqueId travels over the process group that exists already
                ok = True
                # Rank 0 ALWAYS broadcasts -- the id, or None when it could not make one -- so that every rank issues the same
                # collectives in the same order (a rank 0 that skipped the broadcast left the others waiting in it until the watchdog)
                box = [None]
                if rank == 0:
                    try:
                        idbuf = C.create_string_buffer(128)
                        capi.check(L.grt_comm_unique_id(idbuf), "grt_comm_unique_id")
                        box = [idbuf.raw]
                    except Exception as exc:
                        ok, why = False, str(exc)[:80]
                dist.broadcast_object_list(box, src=0)
                if box[0] is None:
                    ok, why = False, (why if rank == 0 else "rank 0 has no communicator id")
                else:
                    try:
                        capi.check(L.grt_comm_init(C.c_int(rank), C.c_int(world), C.c_char_p(box[0])), "grt_comm_init")
                    except Exception as exc:
                        ok, why = False, str(exc)[:80]
                use_lib = agree(ok)           # one rank without a communicator: ALL ranks fall back together
            if use_lib:
                cnt = (C.c_int64 * world)(*npix_of_rank)
                host = np.zeros((args.nfreq, sum(npix_of_rank), nv), dtype=np.float32) if rank == 0 else None
                dist.barrier(); torch.cuda.synchronize()
                tg = time.perf_counter()
                capi.check(L.grt_gather_stage(cnt, C.c_int(0), host.ctypes.data_as(C.c_void_p) if rank == 0 else None), "grt_gather_stage")
                gather_ms = (time.perf_counter() - tg) * 1e3
                gather_via = "grt_gather_stage (library RCCL communicator, grouped send/recv per camera, incl. D2H on rank 0)"
                full = host
                L.grt_comm_free()
            else:      # the process group's own gather (rank-ordered tiles, same layout), decided by all ranks together
                from grtrans_amd.tiling import gather_tiles
                L.grt_comm_free()
                tile = torch.empty((args.nfreq, npix_of_rank[rank], nv), dtype=torch.float32, device="cuda")
                capi.trace_ranges(cam, my_ranges, W["fname"], em, freqs, args.iname, resident=True)
                capi.check(L.grt_copy_resident_pixvals(C.c_void_p(tile.data_ptr()), C.c_int64(tile.numel())), "grt_copy_resident_pixvals")
                torch.cuda.synchronize()
                if cdev == "cpu":
                    tile = tile.cpu()
                dist.barrier()
                tg = time.perf_counter()
                fullt = gather_tiles(tile, dist, 0)
                torch.cuda.synchronize()
                gather_ms = (time.perf_counter() - tg) * 1e3
                gather_via = "torch.distributed.gather (%s)" % why
                full = fullt.cpu().numpy() if rank == 0 else None
            if rank == 0:
                assert full.shape[1] == sum(npix_of_rank)
                if os.environ.get("GRT_BENCH_CHECK_IMAGE") == "1":
                    # test hook: the gathered, un-permuted image equals one whole-image trace on this rank, bit for bit
                    img = np.zeros((args.nfreq, nx_total * ny, nv), dtype=np.float32)
                    pre = 0
                    for rr in all_ranges:
                        for (a, b) in rr:
                            img[:, a - 1:b] = full[:, pre:pre + b - a + 1]
                            pre += b - a + 1
                    ref, _, _, _ = capi.trace_tile(cam, 1, nx_total * ny, W["fname"], em, freqs, args.iname)
                    assert np.array_equal(img, ref), "gathered image differs from the single-GPU image"

    except Exception as exc:      # (the gather failed: the job's line is still printed, the exit code says so)
        gather_ms, gather_via = None, "FAILED: %s" % str(exc)[:120]
        gather_failed = True
    if watchdog is not None:
        watchdog.cancel()
    if emitted.acquire(blocking=False) and rank == 0:
        emit(gather_ms, gather_via)
    sys.stdout.flush()
    if gather_failed:
        os._exit(3)          # (not through destroy_process_group: the other ranks may be stuck in the failed collective)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
