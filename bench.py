#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the sphere-tracing path on the Mandelbulb
scene at 3840x2160 (BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one sample of every pixel of the frame (one run of the reference's
main() per pixel: RenderJobExecutor.tsx:299) + the frame assembly on rank 0.
With N GPUs the frame's rows are sharded in 8-row stripes dealt round-robin
(no exchange between samples; each pixel depends only on itself, SURVEY.md
8(e)) and the colour plane is gathered to rank 0 over RCCL once per step and
put back in image order, as the reference presents once per sample in its live
loop (index.tsx:158-169).  Total work is fixed => strong scaling.

Rank 0 prints ONE JSON line.  `roofline` is the fp32-VALU roofline of the
pixel kernel (the path has no contraction, so no MFMA; HBM traffic is ~100 B
per ~2e4 flops): achieved = algorithmic flops per launch / HIP-event kernel
time.  `cpu_baseline` is the oracle (the CPU restatement, kind "port") on the
host cores over a bounded sample of the same frame.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The HIP runtime maps a process's streams onto 4 hardware queues by default; this path uses the context's stream, up to
# three side streams for the samples in flight, torch's default stream and RCCL's -- with 4 queues two of the side streams
# share one and their kernels serialise (8-GPU share of the headline frame: 0.59 instead of 0.42 ms per sample).
# Read by the runtime when it starts, so it has to be in the environment before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # C3b is the headline; the others are selectable for profiling
    "c3b": dict(name="C3b mandelbulb(power 8, 8 iter, bailout 2) 3840x2160 full [256] 1 point light", scene="mandelbulb",
                width=3840, height=2160, counts=(256,), mode="full", position=(0.0, 0.0, -2.5), light=True),
    "c3a": dict(name="C3a mandelbulb 3840x2160 preview [256]", scene="mandelbulb", width=3840, height=2160, counts=(256,),
                mode="preview", position=(0.0, 0.0, -2.5), light=False),
    "c2": dict(name="C2 single sphere 1920x1080 preview [128]", scene="sphere", width=1920, height=1080, counts=(128,),
               mode="preview", position=(0.0, 0.0, -3.0), light=False),
    "c4": dict(name="C4 csg64 smooth-union 4096x4096 full [128] 1 light", scene="csg64", width=4096, height=4096, counts=(128,),
               mode="full", position=(0.0, 0.0, -5.0), light=True),
    "c5": dict(name="C5 csg64 8192x8192 full [128,64,64] soft light", scene="csg64", width=8192, height=8192, counts=(128, 64, 64),
               mode="full", position=(0.0, 0.0, -5.0), light="soft"),
}


def make_workload(key):
    from raymarching_engine_amd import job as J, scene as S

    w = WORKLOADS[key]
    sc = {"mandelbulb": S.Mandelbulb, "sphere": S.single_sphere, "csg64": S.csg64}[w["scene"]]()
    lights = []
    if w["light"]:
        lights = [J.point_light((2.0, 3.0, -4.0), size=0.3 if w["light"] == "soft" else 0.0)]
    schema = J.make_schema(sc, w["width"], w["height"], counts=w["counts"], render_mode=w["mode"], position=w["position"], lights=lights)
    return w, sc, schema


def cpu_baseline(sc, schema, target_seconds=12.0):
    """The oracle on the host cores over evenly spaced rows of the same frame."""
    from oracle import oracle as O
    from raymarching_engine_amd import job as J

    O.build()
    O.set_tan_mode(O.TAN_PORTABLE)
    W, H = schema["render"]["width"], schema["render"]["height"]
    u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
    cores = O.host_cores()

    # one row per thread-chunk is too fine for OpenMP-over-rows: render bands of `cores` rows
    def run_bands(starts, band, count=False):
        t0 = time.perf_counter()
        flops = 0
        for y in starts:
            fr = O.Frame(W, H, y, band)
            flops += O.render(sc, u, fr, threads=cores, count_flops=count)
        return time.perf_counter() - t0, flops

    band = max(1, cores)
    probe_starts = [int((H - band) * f) for f in (0.1, 0.5, 0.9)]
    t_probe, _ = run_bands(probe_starts, band)
    per_band = max(t_probe / len(probe_starts), 1e-4)
    n = int(max(4, min((H // band), target_seconds / per_band)))
    starts = [int((H - band) * (i + 0.5) / n) for i in range(n)]
    t, _ = run_bands(starts, band)
    px = n * band * W
    # instrumented algorithmic flops per pixel-sample on a thinner sample of the same rows
    _, flops = run_bands(starts[:: max(1, n // 12)], band, count=True)
    flops_px = flops / (len(starts[:: max(1, n // 12)]) * band * W)
    return {"value": px / t / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{n} bands of {band} rows evenly spaced over the {W}x{H} frame ({px} pixel-samples, {t:.1f} s), oracle/rm_oracle.c with OpenMP"}, flops_px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3b", choices=list(WORKLOADS))
    ap.add_argument("--strict", action="store_true", help="parity build instead of the fast build")
    ap.add_argument("--megakernel", action="store_true", help="force the one-thread-one-pixel kernel (default: the library picks per job)")
    ap.add_argument("--wavefront", action="store_true", help="force the wavefront pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-leg", action="store_true",
                    help="also time the same K steps with 3 samples in flight on this one GPU (reported as `overlap`, never as `value`)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="consecutive samples that may overlap on one GPU (rm_ctx_set_samples_in_flight); default: 1 on one GPU, "
                         "so that a kernel's duration in a rocprofv3 trace is the time of a step, 3 when the frame is sharded "
                         "(a shard's launch is too small to fill the chip: a ray is a ~1 ms serial chain)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from raymarching_engine_amd import abi, dist as rmdist, job as J, native, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("RM_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank (testing aid)
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:
            for key, val in (("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(key, val)
        dist.init_process_group("nccl", device_id=dev)

    wl, sc, schema = make_workload(args.workload)
    W, H = wl["width"], wl["height"]
    flags = abi.RM_RENDER_STRICT if args.strict else abi.RM_RENDER_FAST
    if args.megakernel:
        flags |= abi.RM_RENDER_MEGAKERNEL
    if args.wavefront:
        flags |= abi.RM_RENDER_WAVEFRONT

    ctx = native.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # launches ordered with torch / RCCL work
    in_flight = args.in_flight if args.in_flight > 0 else (1 if world == 1 else 3)
    ctx.set_samples_in_flight(in_flight)
    gatherer = rmdist.FrameGatherer(H, W, world, rank, dev, force=force_dist, ctx=ctx)
    row_count = gatherer.rows
    # planes live in torch memory (padded to the largest shard so that the gather is regular)
    planes = [torch.zeros((gatherer.max_rows, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
    fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, world, rank, *(p.data_ptr() for p in planes))
    assert fb.row_count == row_count
    scene = ctx.create_scene(sc)

    h2, h3 = J.halton(2), J.halton(3)

    pending = [None]
    u_step = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))  # only randNoise changes from sample to sample

    def step():
        # render sample n, then start the gather of its (snapshotted) colour plane; the gather runs over
        # RCCL while sample n+1 renders, and frame n is assembled on rank 0 at the start of step n+1
        u_step.randNoise[0], u_step.randNoise[1] = next(h2), next(h3)
        ctx.render_sample(scene, fb, u_step, None, flags)
        if world > 1 or force_dist:
            if pending[0] is not None:
                gatherer.finish(pending[0])
            pending[0] = gatherer.start(planes[0], dist)

    def drain():
        if (world > 1 or force_dist) and pending[0] is not None:
            gatherer.finish(pending[0])
            pending[0] = None

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # the last frame is assembled inside the timed region: K samples rendered, K frames assembled
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = W * H * args.steps / elapsed / 1e6

    overlap = None
    if world == 1 and not force_dist and in_flight == 1 and args.overlap_leg:
        # for information: the same K steps with consecutive samples overlapping on the device (what a sharded run uses)
        ctx.set_samples_in_flight(3)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        ctx.set_samples_in_flight(in_flight)
        overlap = {"samples_in_flight": 3, "value": W * H * args.steps / e1 / 1e6, "ms_per_step": e1 / args.steps * 1e3}

    out = None
    if rank == 0:
        # kernel time of this rank's launch, HIP events on the launch stream
        u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
        kernel_ms = ctx.render_timed(scene, fb, u, max(3, min(args.steps, 10)), None, flags | abi.RM_RENDER_NO_OVERLAP)
        cpu, flops_px = (None, None)
        if world == 1 and not args.no_cpu_baseline:
            cpu, flops_px = cpu_baseline(sc, schema)
        nominal_px = {"c3b": 145.5e3, "c3a": 71.7e3, "c2": 2.2e3, "c4": 403e3, "c5": 817e3}[args.workload]
        px_launch = row_count * W
        roof = None
        traffic = None
        try:  # measured in a separate rocprofv3 --pmc run (profiles/r01_traffic.json says how)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if args.workload == "c3b" and not args.strict:
                traffic = (tj["hbm_bytes_per_frame"] if args.wavefront else tj["megakernel_hbm_bytes_per_frame"]) * px_launch / (3840 * 2160)
        except Exception:
            pass
        if flops_px is not None:
            achieved = flops_px * px_launch / (kernel_ms * 1e-3) / 1e12
            roof = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                    "flops_per_pixel_sample_instrumented": flops_px, "flops_per_pixel_sample_nominal": nominal_px,
                    "frac_nominal": nominal_px * px_launch / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                    "kernel_ms": kernel_ms, "pixels_per_launch": px_launch,
                    "hbm_algorithmic_GBs": 96.0 * px_launch / (kernel_ms * 1e-3) / 1e9, "hbm_peak_GBs": PEAK_HBM_GBS}
        out = {
            "metric": "Mpixels/sec at 3840x2160 Mandelbulb" if args.workload in ("c3b", "c3a") else "Mpixels/sec",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "build": "strict" if args.strict else "fast",
                       "rows_per_gpu": row_count, "pipeline": "megakernel" if args.megakernel else "wavefront" if args.wavefront else "auto (wavefront)" if (args.workload == "c4" and world == 1) else "auto (megakernel)",
                       "sharding": f"{shard.STRIPE_ROWS}-row stripes round-robin over ranks, colour plane gathered to rank 0 over RCCL every step (overlapped with the next sample's render) and put back in image order" if world > 1 else "none",
                       "planes": "color+normal_dof+albedo_depth fp32, accumulated in place", "samples_in_flight": in_flight},
            "roofline": roof, "cpu_baseline": cpu, "overlap": overlap,
        }
    fb.destroy()
    scene.destroy()
    ctx.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
