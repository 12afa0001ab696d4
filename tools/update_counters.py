#!/usr/bin/env python3
"""profiles/<round>_counters.json (round = $RM_COUNTERS_ROUND, default COUNTERS_ROUND below) from a tools/profile_gpu.sh summary:
    tools/update_counters.py <key> <gpurun_out/prof_<tag>/summary.txt> <profiles/kept copy.txt> <frames> <pixels per frame> <pipeline>
<key> = <workload>[_shard]_<build> (what bench.py looks up), <frames> = the frames each PMC pass of that run rendered (warmup + steps x
repeats + the 2 + max(3, min(steps, 20)) launches of the kernel timing), <pipeline> = what the library dispatched (megakernel | wavefront).
Per frame: the SUM over every rm:: kernel's dispatches / frames -- one kernel per frame for the pixel kernel, a dozen for the wavefront
pipeline.  Records the hash of the kernel sources the counters were measured on; bench.py reports `traffic` / `frac_executed` only
while the sources still hash to it."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS_ROUND = "r06"  # the file bench.py reads: profiles/<COUNTERS_ROUND>_counters.json
SKIP = ("rm_order_",)  # the tile-cost sort (two small launches on a side stream): not part of the frame's work


def kernel_source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raymarching-engine_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "hip_raymarch.h"), "rb").read())
    return h.hexdigest()


def parse(summary):
    text = open(summary).read()
    text = text[text.index("== PMC"):]
    kernels, cur = {}, None
    for line in text.splitlines()[1:]:
        if not line.startswith("   "):
            name = line.split(" {")[0].strip()
            cur = kernels.setdefault(name, {"meta": line[line.index("{"):] if "{" in line else ""})
        else:
            m = re.match(r"\s+(\w+)\s+n=\s*(\d+) mean ([0-9.e+-]+) sum ([0-9.e+-]+)", line)
            if m and cur is not None:
                cur[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return kernels


if __name__ == "__main__":
    key, summary, kept, frames, pixels, pipeline = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    kernels = {k: v for k, v in parse(summary).items() if not any(s in k for s in SKIP)}
    tot = {"hbm": 0.0, "flops": 0.0, "valu": 0.0, "threads": 0.0, "trans": 0.0, "fma": 0.0, "mul": 0.0, "add": 0.0}
    per_kernel = {}
    for name, v in kernels.items():
        if "SQ_INSTS_VALU" not in v:
            continue
        s = lambda c: v[c][2] if c in v else 0.0
        la = s("SQ_THREAD_CYCLES_VALU") / (s("SQ_INSTS_VALU") * 64) if s("SQ_INSTS_VALU") else 0.0
        flops = (s("SQ_INSTS_VALU_ADD_F32") + s("SQ_INSTS_VALU_MUL_F32") + 2 * s("SQ_INSTS_VALU_FMA_F32") + s("SQ_INSTS_VALU_TRANS_F32")) * 64 * la
        hbm = (2 * s("FETCH_SIZE") + s("WRITE_SIZE")) * 1000.0
        tot["hbm"] += hbm; tot["flops"] += flops; tot["valu"] += s("SQ_INSTS_VALU"); tot["threads"] += s("SQ_THREAD_CYCLES_VALU"); tot["trans"] += s("SQ_INSTS_VALU_TRANS_F32")
        tot["fma"] += s("SQ_INSTS_VALU_FMA_F32"); tot["mul"] += s("SQ_INSTS_VALU_MUL_F32"); tot["add"] += s("SQ_INSTS_VALU_ADD_F32")
        per_kernel[name] = {"dispatches_per_frame": v["SQ_INSTS_VALU"][0] / frames, "sq_insts_valu_per_frame": s("SQ_INSTS_VALU") / frames, "lanes_active": la,
                            "hbm_bytes_per_frame": hbm / frames, "executed_lane_flops_per_frame": flops / frames, "resources": v["meta"]}
    path = os.path.join(ROOT, "profiles", os.environ.get("RM_COUNTERS_ROUND", COUNTERS_ROUND) + "_counters.json")
    d = json.load(open(path)) if os.path.exists(path) else {
        "_about": "Per-frame hardware counters of bench.py's workloads from separate rocprofv3 --pmc passes of the same command (tools/profile_gpu.sh, "
                  "tools/profile_all.sh), summed over every rm:: kernel of a frame; the summaries they come from are the files named in `profile`.  "
                  "hbm_bytes_per_frame = (2 x FETCH_SIZE + WRITE_SIZE) x 1000 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md, HBM); "
                  "executed_lane_flops_per_frame = (ADD + MUL + 2 x FMA + TRANS fp32 wave-level instructions) x 64 x the fraction of lanes active "
                  "(SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64)), per kernel -- compare / select / move / integer / fp64 instructions are not counted, and a "
                  "doubling done by a VOP3 output modifier is not an instruction.  bench.py reads both (roofline.traffic, roofline.frac_executed) and withholds "
                  "them once the kernel sources no longer hash to kernel_source_sha256."}
    d[key] = {"profile": kept, "pipeline": pipeline, "frames_profiled": frames, "pixels_per_frame": pixels,
              "hbm_bytes_per_frame": tot["hbm"] / frames, "executed_lane_flops_per_frame": tot["flops"] / frames,
              "sq_insts_valu_per_frame": tot["valu"] / frames, "trans_f32_per_frame": tot["trans"] / frames,
              "fma_f32_per_frame": tot["fma"] / frames, "mul_f32_per_frame": tot["mul"] / frames, "add_f32_per_frame": tot["add"] / frames,
              "lanes_active": tot["threads"] / (tot["valu"] * 64) if tot["valu"] else 0.0,
              "hbm_bytes_per_pixel": tot["hbm"] / frames / pixels, "kernels": per_kernel, "kernel_source_sha256": kernel_source_hash()}
    json.dump(d, open(path, "w"), indent=1)
    print("updated", path, key, "hbm/frame %.3g B (%.1f B/px), executed %.3g lane-flops/frame" % (tot["hbm"] / frames, tot["hbm"] / frames / pixels, tot["flops"] / frames))
