#!/usr/bin/env python3
"""profiles/r02_counters.json from a tools/profile_gpu.sh summary: tools/update_counters.py gpurun_out/prof_<tag>/summary.txt profiles/<kept copy>.txt
Records the hash of the kernel sources the counters were measured on; bench.py reports `traffic` / `frac_executed` only while
the sources still hash to it."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raymarching-engine_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "hip_raymarch.h"), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    summary, kept = sys.argv[1], sys.argv[2]
    text = open(summary).read()
    blk = text[text.index("void rm::rm_pixel_kernel<7, false, true, false>(KParams) {"):]
    blk = blk[:blk.index("rm::rm_order_kernel")]
    v = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+n=\s*\d+ mean ([0-9.e+]+)", blk)}
    path = os.path.join(ROOT, "profiles", "r02_counters.json")
    d = json.load(open(path))
    e = d["c3b_fast"]
    e["profile"] = kept
    e["kernel_source_sha256"] = kernel_source_hash()
    e["fetch_size_kb_raw"], e["write_size_kb"] = v["FETCH_SIZE"], v["WRITE_SIZE"]
    e["hbm_bytes_per_frame"] = (v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1000.0
    e["sq_insts_valu"], e["sq_thread_cycles_valu"] = v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"]
    la = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64)
    e["lanes_active"] = la
    w = {k: v["SQ_INSTS_VALU_" + k] for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
    e["wave_level_flop_instructions"] = w
    e["executed_lane_flops_per_frame"] = (w["ADD_F32"] + w["MUL_F32"] + 2 * w["FMA_F32"] + w["TRANS_F32"]) * 64 * la
    json.dump(d, open(path, "w"), indent=1)
    print("updated", path, "hash", e["kernel_source_sha256"][:16])
