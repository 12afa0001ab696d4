#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the sphere-tracing path on the Mandelbulb
scene at 3840x2160 (BASELINE.json), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts
N fresh ranks itself (python -m torch.distributed.run, 127.0.0.1) BEFORE it
touches a GPU, relays rank 0's JSON line and exits with the children's status;
it fails loudly when fewer than N GPUs are visible.  Under a launcher
(WORLD_SIZE set, as the driver does) it is one of the ranks; WORLD_SIZE must
equal --gpus.

A step = one sample of every pixel of the frame (one run of the reference's
main() per pixel: RenderJobExecutor.tsx:299).  With N GPUs the frame's rows are
sharded in 8-row stripes dealt round-robin (no exchange between samples; each
pixel depends only on itself) and the job presents as the
reference's does, every render.sampleYieldInterval samples
(RenderJobExecutor.tsx:163; --yield-interval, 8 when sharded): every rank
tone-maps its stripes (display.frag with depth of field off, rm_present_rows)
and the RGBA8 rows are gathered to rank 0 over RCCL and put back in image order
-- one gather per presented frame, overlapped with the render
of the next samples; the samples between two yields go to the library in one
rm_render_samples call.  --yield-interval 1 presents and gathers after every
sample, like the live loop (index.tsx:158-169).  Total work is fixed => strong
scaling.

Rank 0 prints ONE JSON line.  `roofline` is the fp32-VALU roofline of the
pixel kernel (the path has no contraction, so no MFMA; HBM traffic is ~100 B
per ~2e4 flops): achieved = algorithmic flops per launch / HIP-event kernel
time, the flops instrumented over EVERY row of the frame by the counting oracle
(profiles/flops_per_pixel.json, tools/count_flops.py).  `cpu_baseline` is the
oracle (the CPU restatement, kind "port") on the host cores over an unbiased row
sample of the same frame.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

# The HIP runtime maps a process's streams onto 4 hardware queues by default; this path uses the context's stream, up to
# three side streams for the samples in flight, torch's default stream and RCCL's -- with 4 queues two of the side streams
# share one and their kernels serialise (8-GPU share of the headline frame: 0.59 instead of 0.42 ms per sample).
# Read by the runtime when it starts, so it has to be in the environment before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
# one node by contract (N GPUs over xGMI): no InfiniBand to probe, rendezvous over loopback
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0
VALU_ISSUE_PER_S = 1.0e12  # wave-level VALU instructions/s the chip sustains at the clock it holds under a dense fp32 stream (profiles/r02_valu_issue_rate.txt)
# Round 6's issue-slot model (tools/ubench/fold_rate.hip, profiles/r06_fold_rate.txt): the 64-row fold alone issues 1.03e12 slots/s with no
# transcendental and no LDS read in it; a quarter-rate instruction among ordinary ones takes ~7.1 slots where it stands alone and ~4.1 in a group
# of four (the kernels group them: 3.1 beyond the one it is counted as), a ds_read_b128 2.1.
ISSUE_SLOTS_PER_S, TRANS_EXTRA_SLOTS, LDS_SLOTS = 1.03e12, 3.1, 2.1


def issue_accounted(ent, kernel_ms):
    """Share of the kernel's time that its instruction stream accounts for under the model above (None without the counters)."""
    if not ent or not ent.get("sq_insts_valu_per_frame") or ent.get("sq_insts_lds_per_frame") is None:
        return None
    slots = ent["sq_insts_valu_per_frame"] + TRANS_EXTRA_SLOTS * ent["trans_f32_per_frame"] + LDS_SLOTS * ent["sq_insts_lds_per_frame"]
    return slots / (kernel_ms * 1e-3) / ISSUE_SLOTS_PER_S

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
    # not a BASELINE configuration: the job the reference's page starts with (index.tsx:121-182,308-333: the fractal1 example scene, 1280x720,
    # full mode, [128,128,64,32,32], no lights) -- what a user who switches over runs first
    "live": dict(name="live default: fractal1 (sphere-grid fractal) 1280x720 full [128,128,64,32,32]", scene="fractal1", width=1280, height=720,
                 counts=(128, 128, 64, 32, 32), mode="full", position=(0.0, 0.0, 0.0), light=False),
}
NOMINAL_FLOPS_PX = {"c3b": 145.5e3, "c3a": 71.7e3, "c2": 2.2e3, "c4": 403e3, "c5": 817e3, "live": 409 * 150.0}  # SURVEY.md 8(d), fixed-E estimate


def make_workload(key):
    from raymarching_engine_amd import job as J, scene as S

    w = WORKLOADS[key]
    sc = {"mandelbulb": S.Mandelbulb, "sphere": S.single_sphere, "csg64": S.csg64, "fractal1": S.SphereGridFractal}[w["scene"]]()
    lights = []
    if w["light"]:
        lights = [J.point_light((2.0, 3.0, -4.0), size=0.3 if w["light"] == "soft" else 0.0)]
    schema = J.make_schema(sc, w["width"], w["height"], counts=w["counts"], render_mode=w["mode"], position=w["position"], lights=lights)
    return w, sc, schema


def instrumented_flops(key, rows_held):
    """Algorithmic flops of ONE launch over the image rows `rows_held` (an array of row indices), and per pixel of
    the whole frame, from the committed count of the reference algorithm (profiles/flops_per_pixel.json)."""
    path = os.path.join(ROOT, "profiles", "flops_per_pixel.json")
    try:
        e = json.load(open(path))[key]
    except Exception:
        return None, None, None
    W = e["width"]
    if e["row_stride"] == 1:  # every row counted: price exactly the rows this launch renders
        per_row = e["flops_per_row"]
        return float(sum(per_row[int(r)] for r in rows_held)), e["flops_per_pixel_sample"], e
    return e["flops_per_pixel_sample"] * len(rows_held) * W, e["flops_per_pixel_sample"], e


def cpu_baseline(sc, schema, target_seconds=16.0):
    """The oracle on the host cores over rows spaced evenly over the WHOLE frame (every k-th row: an unbiased sample).
    The box's host is shared and an OpenMP team of every hardware thread oversubscribes it, so the same 1/8-of-the-budget
    row sample is timed with teams of 32, 64, 128 and all hardware threads first; the timed sample then runs with the team
    that was fastest, and `cores` is THAT count (`threads_tried` keeps every rate)."""
    from oracle import oracle as O
    from raymarching_engine_amd import job as J

    O.build()
    O.set_tan_mode(O.TAN_PORTABLE)
    # the C library's float transcendentals: what a CPU port would call, and about twice as fast as the portable
    # double-precision ones the oracle uses as the bit-exact checker (oracle/pm_math.h) -- the faster one is the baseline
    O.set_math_mode(O.MATH_LIBM)
    W, H = schema["render"]["width"], schema["render"]["height"]
    u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
    cores = O.host_cores()

    def rows_for(n):
        k = max(1, H // n)
        return list(range(k // 2, H, k))

    def rate(rows, threads):
        t0 = time.perf_counter()
        O.render_rows(sc, u, W, H, rows, threads=threads)
        t = time.perf_counter() - t0
        return len(rows) * W / t / 1e6, t

    # probe: about one row per core, spread over the frame, to size the samples
    probe = rows_for(max(8, min(H, cores)))
    r0, t = rate(probe, cores)
    per_row = max(t / len(probe), 1e-6)
    teams = sorted({min(cores, c) for c in (32, 64, 128, cores)})
    sweep_rows = rows_for(int(max(len(probe), min(H, target_seconds / 8.0 / per_row))))
    tried = {c: rate(sweep_rows, c)[0] for c in teams}
    best = max(tried, key=tried.get)
    n = int(max(len(probe), min(H, 0.5 * target_seconds * tried[best] * 1e6 / W)))
    rows = rows_for(n)
    value, t = rate(rows, best)
    px = len(rows) * W
    return {"value": value, "unit": "Mpixels/s", "cores": best, "hardware_threads": cores, "kind": "port",
            "threads_tried": {str(c): round(v, 4) for c, v in tried.items()},
            "threads_sample": f"every {max(1, H // len(sweep_rows))}th row ({len(sweep_rows)} rows) per team size",
            "sample": f"every {max(1, H // n)}th row of the {W}x{H} frame ({len(rows)} rows, {px} pixel-samples, {t:.1f} s) on the fastest team ({best} threads), "
                      "oracle/rm_oracle.c with OpenMP over the rows, libm transcendentals"}


def reference_gl(key):
    """The reference's own GLSL under software GL, as measured in the BUILD container (oracle/gl/time_reference.py): it cannot
    run on the GPU box (only this repository travels), so it is quoted with its hardware and stack, never re-timed here."""
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "r01_b1_swiftshader_reference.json")))
    except Exception:
        return None
    tag = {"c3b": "C3b", "c3a": "C3a", "c2": "C2", "c4": "C4", "c5": "C5"}.get(key, "-")
    hits = [r for r in rows if r["case"].startswith(tag + " ")]
    if not hits:
        return None
    r = hits[-1]  # the largest size timed for that configuration (cost per pixel does not depend on the size)
    return {"value": r["Mpix_per_s"], "unit": "Mpixels/s", "cores": r["cores"], "kind": "reference",
            "stack": "the reference's raymarcher.frag under SwiftShader (HeadlessChrome 88, WebGL2) standing in for llvmpipe, 8 cores of the build container",
            "sample": r["case"], "source": "profiles/r01_b1_swiftshader_reference.json"}


def counters_entry(key, strict, pipeline, windowed, gl_stack=0):
    """This workload's entry of profiles/<round>_counters.json (separate rocprofv3 --pmc passes of this command, tools/profile_gpu.sh
    + tools/update_counters.py): HBM bytes and executed lane-flops per frame.  The counters belong to the kernel sources they
    were measured on (their hash is recorded): after a source change they are withheld until re-measured."""
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("update_counters", os.path.join(ROOT, "tools", "update_counters.py"))
        uc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(uc)
        name = uc.COUNTERS_ROUND + "_counters.json"
        cj = json.load(open(os.path.join(ROOT, "profiles", name)))
        base = key + ("_shard" if windowed else "") + ("_glstack" if gl_stack else "_strict" if strict else "_fast")
        ent = cj.get(base + "_" + pipeline) or cj.get(base)  # (workloads profiled with both implementations carry the implementation in the key)
        if not ent or ent.get("kernel_source_sha256") != uc.kernel_source_hash() or (ent.get("pipeline") != pipeline and pipeline is not None):
            return None
        return dict(ent, counters_file="profiles/" + name)
    except Exception:
        return None



# ---- hardware counters measured by THIS run (round 6) --------------------------------------------------------------------------
# Until round 5 `traffic`, `frac_executed`, `valu_issue_busy` and `fma_share` were replayed from the builder's own rocprofv3 passes
# (profiles/<round>_counters.json).  The default one-GPU invocation now measures them itself, before it touches the GPU: three
# child processes -- rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --counter-child -- each rendering a few frames of
# the headline and of every `workloads` leg; a --pmc pass carries --kernel-trace only, FETCH_SIZE and WRITE_SIZE have a pass each
# (MI355X_MICROARCH.md: the TCC block has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2), and the gfx950 correction of FETCH_SIZE
# (128-B requests tallied at 64 B: x 2) is applied.  Anything that goes wrong there -- no rocprofv3, a timeout, an unexpected file --
# falls back to the replayed file, and `counters_from` says which it was.
COUNTER_LEGS = (("c3b", "fast", 0), ("c2", "fast", 0), ("c3a", "fast", 0), ("c4", "fast", 0), ("c5", "fast", 0), ("c3b", "strict", 0), ("c3b", "glstack", 2))
COUNTER_WARM, COUNTER_FRAMES = 2, 2  # launches per leg: the first two settle the tile order and earn a long table its culling grid
PMC_PASSES = (("sq", "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAIT_ANY"),
              ("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("waves", "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"))


def leg_flags(abi, build, gl_stack):
    return abi.RM_RENDER_FAST if build == "fast" else abi.RM_RENDER_STRICT | (abi.RM_RENDER_MEGAKERNEL if gl_stack else 0)


def counter_child():
    """`bench.py --counter-child` (run under rocprofv3 --pmc by live_counters): COUNTER_WARM + COUNTER_FRAMES samples of every leg,
    one pixel-kernel launch each, in COUNTER_LEGS order; prints how many launches each leg made."""
    from raymarching_engine_amd import abi, job as J, native

    ctx = native.Context(0)
    ctx.set_samples_in_flight(1)
    legs = []
    for key, build, gl_stack in COUNTER_LEGS:
        wl, sc, schema = make_workload(key)
        ctx.set_gl_stack(gl_stack)
        h = ctx.create_scene(sc)
        fb = ctx.create_framebuffer(wl["width"], wl["height"])
        J.reset_halton()
        for _ in range(COUNTER_WARM + COUNTER_FRAMES):
            ctx.render_sample(h, fb, J.uniforms_from_schema(schema, J.next_rand_noise()), None, leg_flags(abi, build, gl_stack) | abi.RM_RENDER_NO_OVERLAP)
            ctx.sync()
        legs.append({"key": key + "_" + build, "launches": COUNTER_WARM + COUNTER_FRAMES, "frames": COUNTER_FRAMES, "pixels": wl["width"] * wl["height"],
                     "pipeline": ctx.last_pipeline()})
        fb.destroy()
        h.destroy()
    ctx.set_gl_stack(0)
    ctx.close()
    print("COUNTER_CHILD " + json.dumps(legs), flush=True)


def under_a_profiler():
    e = os.environ
    return bool(e.get("ROCP_TOOL_LIBRARIES") or e.get("ROCPROFILER_REGISTER_FORCE_LOAD") or "rocprof" in e.get("LD_PRELOAD", "") or e.get("RM_BENCH_NO_LIVE_COUNTERS"))


def live_counters(timeout_s=150.0):
    """{leg key: per-frame counters} measured now (see above), or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if os.environ.get("RM_BENCH_FAIL_LIVE_COUNTERS") == "1":  # (testing aid: the fall-back to the replay file)
        return None, "RM_BENCH_FAIL_LIVE_COUNTERS=1 (testing aid)"
    tmp = tempfile.mkdtemp(prefix="rm_bench_pmc_", dir=os.environ.get("TMPDIR") or "/tmp")
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR") or "/tmp")
    t_start = time.perf_counter()
    sums = {}  # leg key -> counter -> sum over the leg's measured frames
    legs = None
    try:
        for tag, group in PMC_PASSES:
            out = os.path.join(tmp, tag)
            cmd = [exe, "--pmc", *group.split(), "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--counter-child"]
            left = timeout_s - (time.perf_counter() - t_start)
            if left < 10.0:
                return None, f"out of time before the {tag} pass"
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, env=env, cwd=tmp)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("COUNTER_CHILD ")]
            if r.returncode != 0 or not line:
                return None, f"the {tag} pass failed (status {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            legs = json.loads(line[-1][len("COUNTER_CHILD "):])
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows += [row for row in csv.DictReader(open(f)) if "rm_pixel_kernel" in row.get("Kernel_Name", "")]
            per_dispatch = {}
            for row in rows:
                d = per_dispatch.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"]})
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            order = [per_dispatch[k] for k in sorted(per_dispatch)]
            if len(order) != sum(l["launches"] for l in legs):
                return None, f"the {tag} pass saw {len(order)} pixel-kernel dispatches, {sum(l['launches'] for l in legs)} were launched"
            at = 0
            for l in legs:
                mine = order[at:at + l["launches"]]
                at += l["launches"]
                if len({d["name"] for d in mine}) != 1:
                    return None, f"the {tag} pass: leg {l['key']} is not one kernel ({sorted({d['name'] for d in mine})})"
                acc = sums.setdefault(l["key"], {"kernel": mine[0]["name"], "frames": l["frames"], "pixels": l["pixels"], "pipeline": l["pipeline"]})
                for d in mine[-l["frames"]:]:
                    for c, v in d.items():
                        if c != "name":
                            acc[c] = acc.get(c, 0.0) + v
    except subprocess.TimeoutExpired:
        return None, f"timed out after {timeout_s:.0f} s"
    except Exception as e:  # the measurement is an extra: any surprise means "replay"
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for key, a in sums.items():
        n = a["frames"]
        g = lambda c: a.get(c, 0.0) / n
        valu = g("SQ_INSTS_VALU")
        lanes = g("SQ_THREAD_CYCLES_VALU") / (valu * 64) if valu else 0.0
        out[key] = {"profile": "this run", "counters_file": "measured in this run", "pipeline": a["pipeline"], "frames_profiled": n, "pixels_per_frame": a["pixels"], "kernel": a["kernel"],
                    "hbm_bytes_per_frame": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1000.0,  # (KB; FETCH_SIZE x 2: gfx950 tallies 128-B requests at 64 B)
                    "hbm_bytes_per_pixel": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1000.0 / a["pixels"],
                    "executed_lane_flops_per_frame": (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_TRANS_F32")) * 64 * lanes,
                    "sq_insts_valu_per_frame": valu, "trans_f32_per_frame": g("SQ_INSTS_VALU_TRANS_F32"), "fma_f32_per_frame": g("SQ_INSTS_VALU_FMA_F32"),
                    "mul_f32_per_frame": g("SQ_INSTS_VALU_MUL_F32"), "add_f32_per_frame": g("SQ_INSTS_VALU_ADD_F32"), "sq_insts_salu_per_frame": g("SQ_INSTS_SALU"),
                    "sq_insts_lds_per_frame": g("SQ_INSTS_LDS"), "lanes_active": lanes,
                    "wave_cycles_waiting": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
                    "seconds_spent": None}
    took = time.perf_counter() - t_start
    for v in out.values():
        v["seconds_spent"] = took
    return out, f"measured in this run: {len(PMC_PASSES)} rocprofv3 --pmc passes (--kernel-trace only) of `bench.py --counter-child`, {COUNTER_FRAMES} frames per leg, {took:.0f} s"


def self_launch(args):
    """--gpus N > 1 with no launcher: start N fresh ranks before this process touches a GPU."""
    import socket

    import torch  # importing torch and counting devices does not initialise the GPU

    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("RM_BENCH_SHARE_GPU") != "1":  # (testing aid: every rank on GPU 0, see main)
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        sys.exit(f"bench.py --gpus {args.gpus}: the ranks exited with status {proc.returncode}")
    if line is None or json.loads(line).get("n_gpus") != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: rank 0 did not report n_gpus = {args.gpus}")
    print(line, flush=True)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3b", choices=list(WORKLOADS))
    ap.add_argument("--rows", default="", help="render only image rows A:B of the frame (e.g. one shard of C4/C5 on one GPU)")
    ap.add_argument("--stripe-of", type=int, default=0, help="one GPU standing in for rank 0 of an N-way split: render its 8-row stripes of the frame (no collective)")
    ap.add_argument("--strict", action="store_true", help="parity build instead of the fast build")
    ap.add_argument("--gl-stack", type=int, default=0, choices=(0, 1, 2), help="the parity build in the GL stack's arithmetic (rm_ctx_set_gl_stack: 1 = with the portable tangent, 2 = its own tan: the "
                                                                              "reference's bits as SwiftShader renders the unmodified shader); implies --strict --megakernel")
    ap.add_argument("--megakernel", action="store_true", help="force the one-thread-one-pixel kernel (default: the library picks per job)")
    ap.add_argument("--wavefront", action="store_true", help="the wavefront pipeline (the tests' second implementation: loads tests/_xcheck/libhip_raymarch_xcheck.so instead of the product library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-leg", action="store_true",
                    help="also time the same K steps with 3 samples in flight on this one GPU (reported as `overlap`, never as `value`)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="consecutive samples that may overlap on one GPU (rm_ctx_set_samples_in_flight); default: 1 on one GPU, "
                         "so that a kernel's duration in a rocprofv3 trace is the time of a step, 4 when the frame is sharded "
                         "(a shard's launch is too small to fill the chip: a ray is a ~1 ms serial chain)")
    ap.add_argument("--yield-interval", type=int, default=0,
                    help="render.sampleYieldInterval of the job (RenderJobExecutor.tsx:163): a frame is presented -- and, when "
                         "sharded, gathered -- every this many samples; the samples in between go to the library in one "
                         "rm_render_samples call.  Default: 1 on one GPU (the live loop's value, index.tsx:166), 8 when the "
                         "frame is sharded")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of exactly --steps steps each; `value` is their median, `spread` their range")
    ap.add_argument("--dof", action="store_true", help="give the job the live default depth of field (dof.amount 0.01 at 1.5): a sharded present then "
                                                       "gathers the packed (colour, DoF radius) rows and rank 0 runs the blur")
    ap.add_argument("--no-far-jump", action="store_true", help="RM_RENDER_NO_FAR_JUMP: march escaping rays step by step (measurement switch, same bits)")
    ap.add_argument("--no-cull", action="store_true", help="RM_RENDER_NO_CULL: fold every row of a primitive table at every point (measurement switch, same bits)")
    ap.add_argument("--counter-child", action="store_true", help="(internal) render a few frames of every leg and exit: what live_counters profiles")
    ap.add_argument("--dump-counters", default="", help="write this run's measured counters to the given file in the format of profiles/<round>_counters.json (what a later run replays when it cannot measure)")
    ap.add_argument("--no-live-counters", action="store_true", help="do not measure the hardware counters in this run (replay profiles/<round>_counters.json)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` legs (the other BASELINE configurations and the two parity builds of the headline, a few steps each)")
    ap.add_argument("--no-check-frame", action="store_true", help="sharded runs check the assembled frame by default (see --check-frame); this skips it")
    ap.add_argument("--check-frame", action="store_true",
                    help="sharded runs: after the timed legs rank 0 renders the same samples on ONE framebuffer, presents it and compares "
                         "the bytes with the frame it assembled from the gathered rows (reported as `frame_check`)")
    args = ap.parse_args()
    if args.wavefront and not os.environ.get("RM_LIB"):
        # the wavefront pipeline is not in the product library (round 5): it is timed from the tests' cross-check build of the same sources
        os.environ["RM_LIB"] = os.path.join(ROOT, "tests", "_xcheck", "libhip_raymarch_xcheck.so")

    if args.counter_child:
        counter_child()
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); they must agree")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    live, live_note = None, None
    default_run = (world == 1 and os.environ.get("RM_BENCH_FORCE_DIST") != "1" and args.workload == "c3b" and not args.strict and not args.rows and args.stripe_of <= 1
                   and not args.wavefront and not args.dof and not args.no_far_jump and not args.no_cull and not args.gl_stack)
    if default_run and not args.no_live_counters and not under_a_profiler():
        live, live_note = live_counters()  # child processes, before this one touches the GPU

    if live and args.dump_counters:
        import importlib.util
        spec = importlib.util.spec_from_file_location("update_counters", os.path.join(ROOT, "tools", "update_counters.py"))
        uc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(uc)
        sha = uc.kernel_source_hash()
        dump = {"_about": "Per-frame hardware counters of bench.py's headline and of every `workloads` leg, as measured by bench.py itself (live_counters: rocprofv3 --pmc child "
                          "passes of `bench.py --counter-child`, --kernel-trace only, FETCH_SIZE and WRITE_SIZE in passes of their own; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1000: "
                          "gfx950 tallies 128-B requests at 64 B; executed_lane_flops = (ADD + MUL + 2 FMA + TRANS fp32 wave instructions) x 64 x lanes active).  A later run that "
                          "cannot measure (no rocprofv3, or under a profiler itself) replays this file while the kernel sources still hash to kernel_source_sha256.",
                "_measured": live_note}
        for key, ent in live.items():
            k2 = {"c4_fast": "c4_fast_megakernel", "c5_fast": "c5_fast_megakernel"}.get(key, key)  # (the keys counters_entry looks up)
            dump[k2] = dict(ent, profile=os.path.relpath(args.dump_counters, ROOT), kernel_source_sha256=sha)
            if k2 != key:
                dump[key] = dump[k2]
        json.dump(dump, open(args.dump_counters, "w"), indent=1)

    import numpy as np
    import torch

    from raymarching_engine_amd import abi, dist as rmdist, job as J, native, shard

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    # RM_BENCH_SHARE_GPU=1 (testing aid, with RM_BENCH_BACKEND=gloo): every rank uses GPU 0, so that the whole N > 1 control
    # flow -- rendezvous, sharding, yields, gathers, assembly, the max over ranks -- runs on a box with one GPU
    share_gpu = os.environ.get("RM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
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
        backend = os.environ.get("RM_BENCH_BACKEND", "nccl")  # "gloo": testing aid, the gathered rows travel through host memory
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    sharded = world > 1 or force_dist
    if not sharded:
        backend = None

    wl, sc, schema = make_workload(args.workload)
    W, H = wl["width"], wl["height"]
    if args.dof:  # the live default (index.tsx:309-310): the job has depth of field, so a sharded present gathers the packed rows
        schema["dof"]["amount"], schema["dof"]["distance"] = 0.01, 1.5
    if args.gl_stack:
        args.strict = args.megakernel = True
    flags = abi.RM_RENDER_STRICT if args.strict else abi.RM_RENDER_FAST
    if args.megakernel:
        flags |= abi.RM_RENDER_MEGAKERNEL
    if args.wavefront:
        flags |= abi.RM_RENDER_WAVEFRONT
    if args.no_far_jump:
        flags |= abi.RM_RENDER_NO_FAR_JUMP
    if args.no_cull:
        flags |= abi.RM_RENDER_NO_CULL

    # The job runs through the render-job API (job.do_render_job on a job.RenderJobContext): sharded, the context owns this
    # rank's striped framebuffer and the gatherers, and makes a torch stream of its own current for renders, snapshots, the
    # collective and the assembly (not torch's default stream: its NULL handle is not ordered with the context's stream).
    rows_window = None
    if args.rows:  # one shard of a frame on one GPU: the context holds that window of rows (global pixel coordinates)
        if sharded:
            sys.exit("bench.py: --rows is a one-GPU option")
        a, b = (int(v) for v in args.rows.split(":"))
        rows_window = (a, b)
    stripes = None
    if args.stripe_of > 1:  # one GPU standing in for rank 0 of an N-way split: its 8-row stripes of the frame, no collective
        if sharded or rows_window:
            sys.exit("bench.py: --stripe-of is a one-GPU option (and not with --rows)")
        stripes = (args.stripe_of, 0)
    group = rmdist.ShardGroup(dist, dev, force=force_dist) if sharded else None
    jctx = J.RenderJobContext(local_rank, flags=flags, group=group, rows=(rows_window[0], rows_window[1] - rows_window[0]) if rows_window else None,
                              stripes=stripes)
    ctx = jctx.native
    if args.gl_stack:
        ctx.set_gl_stack(args.gl_stack)
    if not sharded:
        render_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(render_stream)
        ctx.set_stream(render_stream.cuda_stream)
    # measured on one GPU standing in for a rank (a depth sweep of round 2, ms per sample of a rank's stripes, depth 1/2/3/4/6):
    # whole frame 2.48/2.57/2.55/2.51/2.51, 1/2 of it 1.45/1.32/1.31/1.29/1.28, 1/4 0.88/0.70/0.70/0.68/0.66, 1/8 0.58/0.41/0.46/0.37/0.36
    # sharded, the job yields every 8 samples: a rank's 8 samples go out as ONE launch of a full frame's worth of workgroups
    # (rm_render_samples, rm_ctx_set_sample_batch), three such launches in flight, one present + gather per yield
    yield_interval = args.yield_interval if args.yield_interval > 0 else (1 if world == 1 else 8)
    in_flight = args.in_flight if args.in_flight > 0 else (1 if world == 1 else 4 if yield_interval == 1 else 3)
    ctx.set_samples_in_flight(in_flight)
    payload = "f32dof" if schema["dof"]["amount"] != 0.0 else "rgba8"
    schema["render"]["frameid"] = 7
    fb = jctx.fbo_create(W, H, 7)  # (the job finds it in the context's cache: same size, same frameid -> it keeps accumulating)
    row_count = fb.row_count
    scene = jctx.get_scene(sc)
    samples = [0]
    base_exposure = schema["render"]["exposure"] / schema["render"]["samplesPerPixel"]

    def run(n, interval):
        """n samples of the job through do_render_job: `interval` of them between two yields (one rm_render_samples call);
        sharded, every yield starts the gather of what is shown -- it travels while the next samples render -- and the
        previous frame is assembled on rank 0 first (at most one gather is outstanding)."""
        r = schema["render"]
        r["samplesPerPixel"], r["sampleYieldInterval"] = n, interval
        r["exposure"] = base_exposure * n  # the shader's exposure is render.exposure / samplesPerPixel (RenderJobExecutor.tsx:254-256): the same per sample in every leg
        base = samples[0]

        def present(schema_, context_, fb_, k):
            if k == 0 or not sharded:  # (nothing to show yet; unsharded, `value` is the render alone: planes resident, nothing presented)
                return
            if fb_._pending is not None:
                fb_.finish_present()
            fb_.start_present(base + k)

        res = J.drain(J.do_render_job(schema, jctx)(present))
        if not res.get("success"):
            sys.exit(f"bench.py: the render job failed: {res}")
        samples[0] += n

    def drain():
        if sharded and fb._pending is not None:
            fb.finish_present()

    def timed(n, interval):
        """n steps between barriers + device synchronisation on both sides; the MAX over ranks."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, interval)
        drain()  # the last frame is assembled inside the timed region: n samples rendered, ceil(n / interval) frames assembled
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

    run(args.warmup, yield_interval)
    drain()
    if sharded and args.warmup <= 0:
        # no warm-up step, so no frame has been presented yet: one untimed gather of the (empty) frame, so that RCCL sets up its
        # point-to-point connections outside the timed region (a job's last sample always presents: RenderJobExecutor.tsx:338)
        fb.start_present(1)
        fb.finish_present()
    # EXACTLY K steps per timed region; three regions, `value` from the median (the spread says what one region is worth)
    regions = sorted(timed(args.steps, yield_interval) for _ in range(max(1, args.repeats)))
    elapsed = regions[len(regions) // 2]

    px_frame = W * row_count if (rows_window is not None or stripes is not None) else W * H
    ms_per_step = elapsed / args.steps * 1e3
    value = px_frame * args.steps / elapsed / 1e6
    spread = {"regions": len(regions), "steps_each": args.steps, "ms_per_step_min": regions[0] / args.steps * 1e3,
              "ms_per_step_max": regions[-1] / args.steps * 1e3, "value_min": px_frame * args.steps / regions[-1] / 1e6,
              "value_max": px_frame * args.steps / regions[0] / 1e6}

    # for information, never `value`: the same K steps with a present + gather after EVERY sample (the live loop's
    # sampleYieldInterval = 1), four single-sample launches in flight
    every_sample = None
    if sharded and yield_interval > 1:
        ctx.set_samples_in_flight(4)
        run(min(args.warmup, 8) or 1, 1)
        drain()
        e1 = timed(args.steps, 1)
        ctx.set_samples_in_flight(in_flight)
        every_sample = {"sample_yield_interval": 1, "samples_in_flight": 4, "value": px_frame * args.steps / e1 / 1e6,
                        "ms_per_step": e1 / args.steps * 1e3}

    # --check-frame: the canvas rank 0 was last presented against the same samples rendered and presented on ONE framebuffer
    # here (the accumulation per pixel is in sample order either way: the bytes must be identical)
    frame_check = None
    if sharded and not args.no_check_frame:  # (on by default since round 6: the first run on a real node proves its own frame)
        fb.start_present(samples[0])
        canvas = fb.finish_present()
        torch.cuda.synchronize()
        if rank == 0:
            whole = ctx.create_framebuffer(W, H)
            J.reset_halton()
            pairs = [J.next_rand_noise() for _ in range(samples[0])]
            ctx.set_sample_batch(1)
            ctx.set_samples_in_flight(1)
            ctx.render_samples(scene, whole, J.uniforms_from_schema(schema, pairs[0]), pairs, None, flags)
            expect = whole.present(samples[0])
            whole.destroy()
            ctx.set_sample_batch(0)
            ctx.set_samples_in_flight(in_flight)
            got = canvas.cpu().numpy()
            frame_check = bool(np.array_equal(got, expect))
            if not frame_check:
                sys.exit(f"bench.py: frame check failed: the assembled frame differs from the single-framebuffer render in {int((got != expect).sum())} bytes")

    overlap = None
    if world == 1 and not force_dist and in_flight == 1 and args.overlap_leg:
        # for information: the same K steps with consecutive samples overlapping on the device (what a sharded run uses)
        ctx.set_samples_in_flight(3)
        run(3, 1)
        e1 = timed(args.steps, 1)
        ctx.set_samples_in_flight(in_flight)
        overlap = {"samples_in_flight": 3, "value": px_frame * args.steps / e1 / 1e6, "ms_per_step": e1 / args.steps * 1e3}

    # kernel time of every rank's launch: HIP events on the launch stream
    tile = None  # the whole window this rank's framebuffer holds
    u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
    n_timed = max(3, min(args.steps, 20))
    ctx.render_timed(scene, fb, u, 2, tile, flags | abi.RM_RENDER_NO_OVERLAP)  # let the cost order of this job settle
    kernel_ms = ctx.render_timed(scene, fb, u, n_timed, tile, flags | abi.RM_RENDER_NO_OVERLAP)
    pipeline = ctx.last_pipeline()  # what the library dispatched for this job (rm_ctx_last_pipeline), not a re-derivation of its rule
    kernel_ms_per_rank = [kernel_ms]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = kernel_ms
        dist.all_reduce(t)
        kernel_ms_per_rank = [float(v) for v in t.tolist()]

    # The other claims of DESIGN.md section 6 on the same line (default invocation only; a few steps each): the other BASELINE
    # configurations in the fast build, and the headline in the two parity builds -- strict (the oracle's bits) and the GL
    # stack's arithmetic with its own tan (rm_ctx_set_gl_stack(ctx, 2): the reference's bits as SwiftShader renders them).
    workloads = None
    if (world == 1 and not force_dist and rows_window is None and stripes is None and args.workload == "c3b" and not args.strict
            and not args.no_workloads and not args.wavefront and not args.dof):
        workloads = {}
        legs = (("c2", "fast", 0, 8), ("c3a", "fast", 0, 5), ("c4", "fast", 0, 5), ("c5", "fast", 0, 3), ("c3b", "strict", 0, 5), ("c3b", "glstack", 2, 3))
        for n_leg, (key, build, gl_stack, k_steps) in enumerate(legs):
            wl2, sc2, schema2 = make_workload(key)
            W2, H2 = wl2["width"], wl2["height"]
            leg_flags = (abi.RM_RENDER_FAST if build == "fast" else abi.RM_RENDER_STRICT | (abi.RM_RENDER_MEGAKERNEL if gl_stack else 0))
            for sw, bit in ((args.no_far_jump, abi.RM_RENDER_NO_FAR_JUMP), (args.no_cull, abi.RM_RENDER_NO_CULL)):
                if sw:
                    leg_flags |= bit
            jctx.flags = leg_flags
            ctx.set_gl_stack(gl_stack)
            schema2["render"]["frameid"] = 100 + n_leg
            fb2 = jctx.fbo_create(W2, H2, 100 + n_leg)
            scene2 = jctx.get_scene(sc2)
            base2 = schema2["render"]["exposure"] / schema2["render"]["samplesPerPixel"]

            def run2(n):
                r = schema2["render"]
                r["samplesPerPixel"], r["sampleYieldInterval"], r["exposure"] = n, 1, base2 * n
                res = J.drain(J.do_render_job(schema2, jctx)(lambda *a: None))
                if not res.get("success"):
                    sys.exit(f"bench.py: workload {key} ({build}) failed: {res}")

            run2(2)  # (a long table earns its culling grid here; the tile cost order settles)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run2(k_steps)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t0
            u2 = J.uniforms_from_schema(schema2, (0.5, 1.0 / 3.0))
            k_ms = ctx.render_timed(scene2, fb2, u2, 3, None, leg_flags | abi.RM_RENDER_NO_OVERLAP)
            pipe2 = ctx.last_pipeline()
            ent = (live or {}).get(key + "_" + build) or counters_entry(key, build == "strict", pipe2, False, gl_stack)
            leg = {"workload": wl2["name"], "build": build + (" (rm_ctx_set_gl_stack 2: the GL stack's arithmetic, its own tan)" if gl_stack else ""),
                   "steps": k_steps, "ms_per_step": e2 / k_steps * 1e3, "kernel_ms": k_ms, "mpix_s": W2 * H2 * k_steps / e2 / 1e6,
                   "frac_executed": ent["executed_lane_flops_per_frame"] / (k_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS if ent else None,
                   "hbm_bytes_per_px": ent["hbm_bytes_per_pixel"] if ent else None, "lanes_active": ent["lanes_active"] if ent else None,
                   "valu_issue_busy": (ent["sq_insts_valu_per_frame"] + 2.2 * ent["trans_f32_per_frame"]) / (k_ms * 1e-3) / VALU_ISSUE_PER_S if ent else None,
                   "salu_per_valu": ent["sq_insts_salu_per_frame"] / ent["sq_insts_valu_per_frame"] if ent and ent.get("sq_insts_salu_per_frame") else None,
                   "issue_accounted": issue_accounted(ent, k_ms),
                   "counters_from": (ent["counters_file"] + ("" if ent["profile"] == "this run" else " -> " + ent["profile"])) if ent else None}
            workloads[key + "_" + build] = leg
            jctx.fbo_delete(W2, H2, 100 + n_leg)
            while jctx._purgatory:  # (C5's planes are 3.2 GB: give them back before the next leg)
                jctx._purgatory.pop(0)[1].destroy()
        ctx.set_gl_stack(0)
        jctx.flags = flags

    # what the collective library actually saw (sharded runs): proof on the line that N ranks took part
    collective = None
    if sharded:
        collective = {"backend": backend, "world_size_seen": dist.get_world_size(), "ranks_sharing_one_gpu": bool(share_gpu),
                      "payload": payload, "gathered_bytes_per_present": int(fb.gathered_bytes_per_present()) if hasattr(fb, "gathered_bytes_per_present") else None,
                      "frame_check": frame_check, "frame_check_samples": samples[0] if frame_check is not None else None}

    out = None
    if rank == 0:
        rows_held = (np.arange(rows_window[0], rows_window[1]) if rows_window is not None else shard.owned_rows(H, stripes[0], stripes[1]) if stripes is not None
                     else shard.owned_rows(H, world, rank))
        px_launch = len(rows_held) * W
        flops_launch, flops_px, fixture = instrumented_flops(args.workload, rows_held)
        useful_launch, useful_px, useful_fixture = instrumented_flops(args.workload + "_pruned", rows_held)  # ... of the PRUNED algorithm (frac_useful)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sc, schema)
            cpu["reference_gl"] = reference_gl(args.workload)
        nominal_px = NOMINAL_FLOPS_PX[args.workload]
        roof = None
        # NOT measured in this run: replayed from the builder's separate rocprofv3 --pmc runs of this command, kept in
        # profiles/<round>_counters.json with the hash of the kernel sources they were measured on (withheld when the sources changed)
        traffic = executed = counters_file = counters_from = issue_busy = fma_share = None
        ent = (live or {}).get("c3b_fast") if default_run else None
        if ent and ent["pipeline"] != pipeline:
            ent = None
        ent = ent or counters_entry(args.workload, args.strict, pipeline, rows_window is not None or stripes is not None, args.gl_stack)
        if ent:
            share = px_launch / ent["pixels_per_frame"]
            traffic = ent["hbm_bytes_per_frame"] * share
            executed = ent["executed_lane_flops_per_frame"] * share
            counters_file = ent["profile"]
            counters_from = (live_note if ent["profile"] == "this run" else
                             f"{ent['counters_file']} (builder-measured with rocprofv3 --pmc, REPLAYED; kernel sources sha256 {ent['kernel_source_sha256'][:16]})"
                             + (f"; not measured in this run: {live_note}" if live_note else ""))
            if ent.get("fma_f32_per_frame") is not None and ent["sq_insts_valu_per_frame"]:
                fma_share = ent["fma_f32_per_frame"] / ent["sq_insts_valu_per_frame"]
                # issue slots: a transcendental holds the SIMD's issue for 3.2 ordinary slots; a gfx950 sustains ~1.0e12 wave-level
                # VALU instructions/s at the clock it holds under this load (profiles/r02_valu_issue_rate.txt)
                issue_busy = (ent["sq_insts_valu_per_frame"] + 2.2 * ent["trans_f32_per_frame"]) * share / (kernel_ms * 1e-3) / VALU_ISSUE_PER_S
        if flops_launch is not None:
            sec = kernel_ms * 1e-3
            achieved = flops_launch / sec / 1e12
            roof = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                    "frac_nominal": nominal_px * px_launch / sec / 1e12 / PEAK_FP32_VALU_TFLOPS,
                    "frac_executed": executed / sec / 1e12 / PEAK_FP32_VALU_TFLOPS if executed else None,
                    # the arithmetic the marches NEED, instrumented by the same oracle: a ray is counted up to its bitwise fixed point
                    # (settled, or overflowed to its end state) and not beyond -- a numerator the kernel cannot beat by skipping steps,
                    # so this fraction cannot exceed 1 (frac can: 4.5 on C4)
                    "frac_useful": useful_launch / sec / 1e12 / PEAK_FP32_VALU_TFLOPS if useful_launch else None,
                    "flops_per_pixel_sample_useful": useful_px,
                    "counters_from": counters_from, "valu_issue_busy": issue_busy, "fma_share": fma_share, "issue_accounted": issue_accounted(ent, kernel_ms),
                    "flops_per_pixel_sample_instrumented": flops_px, "flops_per_pixel_sample_nominal": nominal_px,
                    "flops_per_launch_instrumented": flops_launch,
                    "flops_source": f"profiles/flops_per_pixel.json[{args.workload}] (every {fixture['row_stride']} row(s) of the whole frame; tools/count_flops.py)",
                    "executed_source": counters_file,
                    "kernel_ms": kernel_ms, "kernel_ms_per_rank": kernel_ms_per_rank, "kernel_launches_timed": n_timed, "pixels_per_launch": px_launch,
                    "hbm_algorithmic_GBs": 96.0 * px_launch / sec / 1e9, "hbm_peak_GBs": PEAK_HBM_GBS,
                    "note": "frac prices the REFERENCE algorithm (every step of its fixed-count marches, SURVEY 8(d)) at the kernel's time; the kernel skips, "
                            "exactly, the steps whose outcome is known -- a ray that stopped moving, an escaping ray's way to overflow -- so frac can "
                            "exceed 1 on frames that are mostly sky; frac_useful prices the same algorithm with every march counted up to its bitwise "
                            "fixed point only (profiles/flops_per_pixel.json[<workload>_pruned]; tools/count_flops.py --pruned): at most 1 by construction; "
                            "frac_executed is the hardware's own count of the arithmetic done.  traffic, frac_executed, "
                            "valu_issue_busy and fma_share come from the hardware counters named in counters_from (measured by this run's own rocprofv3 passes "
                            "on the default one-GPU invocation, else replayed from the builder's): frac_executed = (ADD + MUL + 2 FMA + TRANS) "
                            "x 64 x lanes active / kernel time / peak; with fma_share of the VALU instructions being FMAs (2 flops) and the rest 1 or 0, and "
                            "valu_issue_busy of the issue slots taken, that is what bounds it.  issue_accounted (round 6) = (VALU + 3.1 TRANS + 2.1 LDS instructions) / kernel time / 1.03e12: the share of the "
                            "kernel's time its instruction stream accounts for at what instructions cost on this chip in such a mix (tools/ubench/fold_rate.hip: a transcendental in a group of four "
                            "4.1 slots, a ds_read_b128 2.1, a dense fp32 stream 1.03e12 slots/s); the rest is waves that are not there to issue"}
        out = {
            "metric": "Mpixels/sec at 3840x2160 Mandelbulb" if args.workload in ("c3b", "c3a") else "Mpixels/sec",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "spread": spread,
            "config": {"workload": wl["name"] + (" + depth of field 0.01 @ 1.5" if args.dof else ""), "build": (f"glstack (rm_ctx_set_gl_stack {args.gl_stack})" if args.gl_stack else "strict" if args.strict else "fast"),
                       "rows_per_gpu": row_count, "stripe_of": args.stripe_of if stripes else None,
                       "pipeline": pipeline + ("" if (args.megakernel or args.wavefront) else " (the library's choice)"),
                       "host": "job.do_render_job on a job.RenderJobContext" + (" (sharded: dist.ShardGroup)" if sharded else ""),
                       "sharding": (f"{shard.STRIPE_ROWS}-row stripes round-robin over ranks; a step = one sample of every pixel, plus -- every {yield_interval} sample(s) "
                                    f"(render.sampleYieldInterval) -- one present: each rank {'packs (colour, DoF radius) of' if payload == 'f32dof' else 'tone-maps'} its rows, the {payload} "
                                    f"rows are {'all-gathered to every rank' if payload == 'f32dof' else 'gathered to rank 0'} over {'RCCL' if backend == 'nccl' else backend + ' (testing aid: through host memory)'} (overlapped with the next samples' render) and put back in image order"
                                    + (", every rank runs the present pass with its blur for the stripes it holds (1 / N of the pass) and the RGBA8 rows are gathered to rank 0" if payload == "f32dof" else ""))
                       if sharded else "none",
                       "planes": "color+normal_dof+albedo_depth fp32, accumulated in place", "samples_in_flight": in_flight,
                       "sample_yield_interval": yield_interval},
            "roofline": roof, "cpu_baseline": cpu, "overlap": overlap, "present_every_sample": every_sample, "frame_check": frame_check,
            "collective": collective, "workloads": workloads,
        }
    jctx.fbo_delete(W, H, 7)
    fb.destroy()
    scene.destroy()
    ctx.close()
    if sharded:
        # RCCL writes a version banner to C stdout when it starts; on a pipe or a file that buffer is flushed when the process exits,
        # i.e. AFTER the line below.  Every rank empties it before the last barrier, so that rank 0's JSON is the last line of the
        # job's output whichever way a reader picks it.
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
