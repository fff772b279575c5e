#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-pixel path (BASELINE.json metric).

A "step" = one pass of the hot path (set-up kernel + trace kernel) over one camera tile of
synthetic-free, fully deterministic input: FFJET (BL09 M87 jet table shipped in grtrans_amd/data)
POLSYNCHPL nvals=4 spin=0.998 nn=[1024,1024,400] nfreq=1, formal integrator -- the configuration
the metric "ray-steps/sec + HBM GB/s" is quoted on.  Inputs (fluid table, coefficient tables) are
resident in HBM before the timed region; outputs stay in HBM (16 B/ray D2H is outside).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).
  --scaling weak (default): the camera is nn=[1024*N,1024,400] over the same field of view and rank r traces the
      contiguous pixel strip i1..i2 of 1024*1024 rays (the reference's own i1,i2 partition key).
  --scaling strong: BASELINE config 5's shape -- the SAME nn=[1024,1024,400] camera shared out over the N ranks in
      tiles of --tile-cols image columns dealt round-robin (grtrans_amd/tiling.py:ranges, the share-out
      module pgrtrans uses); `value` = ray-steps of the whole image / max-over-ranks time.
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
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: fp64 vector (non-matrix) peak


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--tile-cols", type=int, default=32, help="strong scaling: image columns per tile (round-robin over the ranks)")
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
        my_ranges = pixel_ranges(rank, world, 1, nx * ny, ny, args.tile_cols if world > 1 else 0)
        all_ranges = [pixel_ranges(r, world, 1, nx * ny, ny, args.tile_cols if world > 1 else 0) for r in range(world)]
    else:
        my_ranges = [(rank * nx * ny + 1, (rank + 1) * nx * ny)]
        all_ranges = [[(r * nx * ny + 1, (r + 1) * nx * ny)] for r in range(world)]
    npix_of_rank = [sum(b - a + 1 for a, b in rr) for rr in all_ranges]
    capi.record_steps(not args.no_steps_record)
    capi.geo_reuse(False)      # every timed step is the whole hot path: set-up + geodesics + transfer (no resident-record re-use)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    freqs = [W["freq"]] if args.nfreq == 1 else list(2.41e10 * np.exp(np.arange(args.nfreq) * np.log(6.31e14 / 2.41e10) / (args.nfreq - 1)))

    import ctypes as C

    class Acc:           # one step = every range of this rank traced once (one range unless --scaling strong with N > 1)
        ray_steps = 0
        ms_trace = ms_setup = ms_geo = ms_xfer = ms_total = 0.0

    def one_step(stage=False):
        a = Acc()
        off = 0
        for (i1, i2) in my_ranges:
            _, _, _, st = capi.trace_tile(cam, i1, i2, W["fname"], em, freqs, args.iname, resident=True)
            a.ray_steps += st.ray_steps; a.ms_trace += st.ms_trace; a.ms_setup += st.ms_setup
            a.ms_geo += st.ms_geo; a.ms_xfer += st.ms_transfer; a.ms_total += st.ms_total
            if stage:    # device-to-device into the rank-level image the gather sends
                capi.check(capi.lib().grt_stage_resident(C.c_int(0), C.c_int(args.nfreq), C.c_int64(i2 - i1 + 1), C.c_int64(off)),
                           "grt_stage_resident")
            off += i2 - i1 + 1
        return a

    # secondary figure: image-only mode (no per-step record, azimuth integrals skipped), untimed region
    image_only = None
    if not args.no_steps_record:
        capi.record_steps(False)
        one_step()
        st0 = one_step()
        image_only = {"ray_steps_per_s": st0.ray_steps / (st0.ms_total / 1e3), "ms_geo": st0.ms_geo,
                      "ms_transfer": st0.ms_xfer}
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
        ms_xfer += st.ms_xfer
    barrier()
    dt = time.perf_counter() - t0
    gather_ms, gather_via = None, None
    ray_steps_all = ray_steps
    if dist is not None:       # the timing of the job: max over ranks of the bracketed wall time, sum of the ranks' ray-steps
        tmax = torch.tensor([dt, float(ray_steps), ms_trace], dtype=torch.float64, device=cdev)
        tsum = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        ray_steps_all = int(tsum[1].item())
        ms_trace = float(tmax[2])
    # The image assembly below (one gather, after the timed steps) has only ever run with more than one rank on the driver's
    # node: if it does not finish, the line is printed without its timing and the job ends instead of hanging.
    import threading
    emitted = threading.Lock()
    state = {"emit": None}

    def on_timeout():
        if emitted.acquire(blocking=False):
            if rank == 0 and state["emit"] is not None:
                state["emit"](None, "gather did not finish within %s s (line printed without it)" % os.environ.get("GRT_BENCH_GATHER_TIMEOUT", "240"))
                sys.stdout.flush()
            os._exit(0)
    watchdog = threading.Timer(float(os.environ.get("GRT_BENCH_GATHER_TIMEOUT", "240")), on_timeout) if dist is not None else None
    def emit(gather_ms, gather_via):
        value = ray_steps_all / dt
        # dominant kernel = grt_geo_kernel (fast class); its launch + the transfer launch process the same
        # ray-steps, so the algorithmic 64 B/step is priced over the sum of the two (HIP events on the library stream)
        t_kernel = ms_trace / args.steps / 1e3
        achieved = BYTES_PER_STEP * (ray_steps / args.steps) / t_kernel / 1e9
        traffic = None
        valu = None
        # the PMC passes were collected on the default workload only (tools/profile_round.sh -> profiles/traffic.json, valu.json)
        same_as_profiled = (args.workload == "ffjet" and [nx, ny, nup] == [1024, 1024, 400] and args.nfreq == 1
                            and args.iname == "formal" and not args.no_steps_record and len(my_ranges) == 1)
        if same_as_profiled:
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
            try:
                # fp64 VALU roofline (the bound that matters): instruction counts per launch from the committed
                # SQ-counter summary (same workload, same library), priced over THIS run's HIP-event kernel time
                v = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
                flop = 2.0 * v["fma_f64_wave_insts"] * 64 + (v["mul_f64_wave_insts"] + v["add_f64_wave_insts"]) * 64.0
                valu = {"bound": "fp64 vector ALU", "fma_f64": v["fma_f64_wave_insts"], "mul_f64": v["mul_f64_wave_insts"],
                        "add_f64": v["add_f64_wave_insts"], "trans_f64": v["trans_f64_wave_insts"], "valu_wave_insts": v["valu_wave_insts"],
                        "counts": "wave64 instructions per trace phase (geo + transfer launches of both classes), %s" % v["source"],
                        "flop_per_launch": flop, "achieved": flop / t_kernel / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flop / t_kernel / 1e12 / FP64_VALU_PEAK_TFLOPS,
                        # issue slots: 4 cycles per wave64 fp64 instruction (16 for the quarter-rate seeds) on each of 4 SIMDs x 256 CUs
                        "valu_issue_utilisation": (4.0 * (v["valu_wave_insts"] - v["trans_f64_wave_insts"]) + 16.0 * v["trans_f64_wave_insts"]) /
                                                  (t_kernel * v["sclk_hz"] * 4 * 256)}
            except Exception:
                valu = None
        out = {
            "metric": "ray-steps/sec", "value": value, "unit": "ray-steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "deterministic: BL09 FFJET table shipped with the reference (no RNG in the path)",
            "config": {"workload": "%s %s nvals=4 spin=%g mu0=%g nn=[%d,%d,%d] %s, nfreq=%d (%g Hz), iname=%s, "
                                   "gridvals=%s, uout=%g" % (W["fname"], W["ename"], W["spin"], W["mu0"], nx, ny, nup,
                                                             "whole image over all GPUs" if strong else "per GPU", args.nfreq,
                                                             freqs[0], args.iname, list(W["gridvals"]), W["uout"]),
                       "pixels_per_gpu": npix_of_rank[0], "ray_steps_per_step_per_gpu": ray_steps // args.steps,
                       "parallelism": ("tiles of %d image columns round-robin over the GPUs" % args.tile_cols if strong and world > 1 else
                                       "pixel strips i1..i2 per GPU") + ", 1 process/GPU, one RCCL gather of the fp32 cameras",
                       "per_step_record_written": not args.no_steps_record, "nfreq": args.nfreq,
                       "ray_step_frequencies_per_s": value * args.nfreq,
                       "gather_ms": gather_ms, "gather_via": gather_via, "setup_kernel_ms": ms_setup / args.steps,
                       "geo_kernel_ms": ms_geo / args.steps, "transfer_kernel_ms": ms_xfer / args.steps,
                       "image_only_mode": image_only},
            "roofline": {"bound": "hbm", "kernel": "grt_geo_kernel+grt_transfer_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_ratio": (traffic / (BYTES_PER_STEP * (ray_steps / args.steps))) if traffic else None,
                         "valu": valu,
                         "note": "algorithmic 64 B per ray-step (SURVEY 8d) over the trace kernels' HIP-event time; the path is "
                                 "fp64-VALU issue bound, not HBM bound: see roofline.valu and DESIGN.md 5"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == "ffjet":
            out["cpu_baseline"] = cpu_baseline(nx, ny, nup)
        print(json.dumps(out))
    state["emit"] = emit
    if watchdog is not None:
        watchdog.daemon = True
        watchdog.start()
    try:
        if dist is not None:
            # assemble ivals on rank 0 (pgrtrans.f90:221-242): ONE gather of the ranks' fp32 cameras over xGMI, after the timed steps
            L = capi.lib()
            nv = 4
            capi.check(L.grt_stage_reset(C.c_int(args.nfreq), C.c_int(nv), C.c_int64(npix_of_rank[rank])), "grt_stage_reset")
            one_step(stage=True)                                    # untimed: fills the rank-level image (same bits as the timed steps)
            torch.cuda.synchronize()
            full = None
            try:
                if cdev != "cuda" or os.environ.get("GRT_BENCH_GATHER") == "torch":
                    raise RuntimeError("torch gather requested")
                # the library's own RCCL communicator; its ncclUniqueId travels over the process group that exists already
                idbuf = C.create_string_buffer(128)
                if rank == 0:
                    capi.check(L.grt_comm_unique_id(idbuf), "grt_comm_unique_id")
                box = [idbuf.raw]
                dist.broadcast_object_list(box, src=0)
                capi.check(L.grt_comm_init(C.c_int(rank), C.c_int(world), C.c_char_p(box[0])), "grt_comm_init")
                cnt = (C.c_int64 * world)(*npix_of_rank)
                host = np.zeros((args.nfreq, sum(npix_of_rank), nv), dtype=np.float32) if rank == 0 else None
                dist.barrier(); torch.cuda.synchronize()
                tg = time.perf_counter()
                capi.check(L.grt_gather_stage(cnt, C.c_int(0), host.ctypes.data_as(C.c_void_p) if rank == 0 else None), "grt_gather_stage")
                gather_ms = (time.perf_counter() - tg) * 1e3
                gather_via = "grt_gather_stage (library RCCL communicator, grouped send/recv per camera, incl. D2H on rank 0)"
                full = host
                L.grt_comm_free()
            except Exception as exc:      # fall back to the process group's own gather (rank-ordered tiles, same layout)
                from grtrans_amd.tiling import gather_tiles
                tile = torch.empty((args.nfreq, npix_of_rank[rank], nv), dtype=torch.float32, device="cuda")
                off = 0
                L.grt_comm_free()
                for (i1, i2) in my_ranges:
                    capi.trace_tile(cam, i1, i2, W["fname"], em, freqs, args.iname, resident=True)
                    t1 = torch.empty((args.nfreq, i2 - i1 + 1, nv), dtype=torch.float32, device="cuda")
                    capi.check(L.grt_copy_resident_pixvals(C.c_void_p(t1.data_ptr()), C.c_int64(t1.numel())), "grt_copy_resident_pixvals")
                    tile[:, off:off + i2 - i1 + 1] = t1
                    off += i2 - i1 + 1
                torch.cuda.synchronize()
                if cdev == "cpu":
                    tile = tile.cpu()
                dist.barrier()
                tg = time.perf_counter()
                fullt = gather_tiles(tile, dist, 0)
                torch.cuda.synchronize()
                gather_ms = (time.perf_counter() - tg) * 1e3
                gather_via = "torch.distributed.gather (%s)" % (str(exc)[:80])
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

    except Exception as exc:      # (both gather paths failed: the job's line is still printed)
        gather_ms, gather_via = None, "gather failed: %s" % str(exc)[:120]
    if watchdog is not None:
        watchdog.cancel()
    if emitted.acquire(blocking=False) and rank == 0:
        emit(gather_ms, gather_via)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
