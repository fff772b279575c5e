#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-pixel path (BASELINE.json metric).

A "step" = one pass of the hot path (set-up kernel + trace kernel) over one camera tile of
synthetic-free, fully deterministic input: FFJET (BL09 M87 jet table shipped in grtrans_amd/data)
POLSYNCHPL nvals=4 spin=0.998 nn=[1024,1024,400] nfreq=1, formal integrator -- the configuration
the metric "ray-steps/sec + HBM GB/s" is quoted on.  Inputs (fluid table, coefficient tables) are
resident in HBM before the timed region; outputs stay in HBM (16 B/ray D2H is outside).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  Weak scaling: the camera is
nn=[1024*N,1024,400] over the same field of view and rank r traces the contiguous pixel strip
i1..i2 of 1024*1024 rays (the reference's own i1,i2 partition key); the only exchange is one
gather of the fp32 tiles to rank 0 over xGMI after the timed kernels of each step.
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


def cpu_baseline(nx, ny, nup, budget_s=15.0):
    """Reference Fortran (oracle/_ref) -- or the C++ port if that library is absent -- on a bounded
    contiguous pixel sample around the image centre of the SAME camera, all host cores.  Runs in a child
    process: the reference writes banners to Fortran unit 6, which must not reach this process's stdout
    (one JSON line only)."""
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".json") as tf:
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "json.dump(bench._cpu_baseline(%d, %d, %d, %f), open(%r, 'w'))" % (ROOT, nx, ny, nup, budget_s, tf.name))
        subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        return json.load(open(tf.name))


def _cpu_baseline(nx, ny, nup, budget_s):
    cores = os.cpu_count() or 1
    kw = dict(nn=(nx, ny, nup), iname="formal")
    try:
        from oracle import refbind
        if not refbind.available():
            raise RuntimeError("no _ref")
        eng, kind = refbind.Ref(**kw), "reference"
    except Exception:
        from oracle import orabind
        eng, kind = orabind.Oracle(**kw), "port"
    centre = (nx // 2) * ny + 1
    nrays = max(cores * 8, 64)
    # grow the sample until it costs about budget_s of wall time on this host (bounded by 200k rays)
    while True:
        _, npts, sec = eng.trace(centre, centre + nrays - 1, [FREQ], nthreads=cores)
        if sec >= 0.5 * budget_s or nrays >= min(200000, nx * ny - centre):
            break
        nrays = int(min(max(nrays * min(budget_s / max(sec, 1e-3), 16.0), nrays * 2), 200000, nx * ny - centre))
    steps = int(np.abs(npts).sum())
    return {"value": steps / sec, "unit": "ray-steps/s", "cores": cores, "kind": kind,
            "sample": "%d contiguous pixels from the image centre of the same %dx%dx%d FFJET camera, "
                      "%d ray-steps in %.1f s, OpenMP schedule(static,1)" % (nrays, nx, ny, nup, steps, sec)}


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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
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
    nx_total = nx * world                      # weak scaling: camera widens with the rank count
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
    i1 = rank * nx * ny + 1
    i2 = (rank + 1) * nx * ny
    capi.record_steps(not args.no_steps_record)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    freqs = [W["freq"]] if args.nfreq == 1 else list(2.41e10 * np.exp(np.arange(args.nfreq) * np.log(6.31e14 / 2.41e10) / (args.nfreq - 1)))

    def one_step():
        _, _, _, st = capi.trace_tile(cam, i1, i2, W["fname"], em, freqs, args.iname, resident=True)
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
    gather_ms = None
    if dist is not None:
        # assemble ivals on rank 0 (pgrtrans.f90:221-242): one RCCL gather of the fp32 tiles over xGMI
        import ctypes as C
        from grtrans_amd.tiling import gather_tiles
        tile = torch.empty((args.nfreq, nx * ny, 4), dtype=torch.float32, device="cuda")
        capi.check(capi.lib().grt_copy_resident_pixvals(C.c_void_p(tile.data_ptr()), C.c_int64(tile.numel())),
                   "grt_copy_resident_pixvals")
        torch.cuda.synchronize()
        if cdev == "cpu":
            tile = tile.cpu()
        tg = time.perf_counter()
        full = gather_tiles(tile, dist, 0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert full.shape[1] == nx * ny * world
        tmax = torch.tensor([dt, float(ray_steps), ms_trace], dtype=torch.float64, device=cdev)
        tsum = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        ray_steps_all = int(tsum[1].item())
        ms_trace = float(tmax[2])
    else:
        ray_steps_all = ray_steps

    if rank == 0:
        value = ray_steps_all / dt
        # dominant kernel = grt_geo_kernel (fast class); its launch + the transfer launch process the same
        # ray-steps, so the algorithmic 64 B/step is priced over the sum of the two (HIP events on the library stream)
        t_kernel = ms_trace / args.steps / 1e3
        achieved = BYTES_PER_STEP * (ray_steps / args.steps) / t_kernel / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        # the PMC passes were collected on the default workload only (tools/profile_round.sh)
        same_as_profiled = (args.workload == "ffjet" and [nx, ny, nup] == [1024, 1024, 400] and args.nfreq == 1
                            and args.iname == "formal" and not args.no_steps_record)
        if os.path.exists(tfile) and same_as_profiled:
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "ray-steps/sec", "value": value, "unit": "ray-steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "deterministic: BL09 FFJET table shipped with the reference (no RNG in the path)",
            "config": {"workload": "%s %s nvals=4 spin=%g mu0=%g nn=[%d,%d,%d] per GPU, nfreq=%d (%g Hz), iname=%s, "
                                   "gridvals=%s, uout=%g" % (W["fname"], W["ename"], W["spin"], W["mu0"], nx, ny, nup, args.nfreq,
                                                             freqs[0], args.iname, list(W["gridvals"]), W["uout"]),
                       "pixels_per_gpu": nx * ny, "ray_steps_per_step_per_gpu": ray_steps // args.steps,
                       "parallelism": "pixel strips i1..i2 per GPU, 1 process/GPU, RCCL gather of fp32 tiles",
                       "per_step_record_written": not args.no_steps_record, "nfreq": args.nfreq,
                       "ray_step_frequencies_per_s": value * args.nfreq,
                       "gather_ms": gather_ms, "setup_kernel_ms": ms_setup / args.steps,
                       "geo_kernel_ms": ms_geo / args.steps, "transfer_kernel_ms": ms_xfer / args.steps,
                       "image_only_mode": image_only},
            "roofline": {"bound": "hbm", "kernel": "grt_geo_kernel+grt_transfer_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic 64 B per ray-step (SURVEY 8d) over the trace kernel's HIP-event time; "
                                 "the kernel is fp64-VALU bound (~1e4 flop-equivalents per step), see DESIGN.md"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == "ffjet":
            out["cpu_baseline"] = cpu_baseline(nx, ny, nup)
        print(json.dumps(out))
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
