#!/usr/bin/env python3
"""After `round_evidence.sh <round> A`, `collect_profiles.sh <round>` and `round_evidence.sh <round> B`: move what those wrote under
gpurun_out/ into profiles/<round>_*, keeping each file's hand-written header (the text above its first data line) and rebuilding the
per-phase table of the headline kernel from the cumulative lines.   python tools/install_evidence.py r04"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rd = lambda *p: open(os.path.join(*p)).read()
def header(name, marker):
    s = rd(P, name)
    return s[:s.index(marker)]
def body(path, marker):
    s = "\n".join(l for l in rd(G, path).splitlines() if "amdgpu.ids" not in l)
    return s[s.index(marker):] + "\n"

# ---- the headline kernel's phases (only when part A ran the diagnostic builds: rounds that did not keep the previous round's table)
HAVE_PHASES = os.path.exists(os.path.join(G, f"{R}_phase.txt")) and os.path.exists(os.path.join(P, f"{R}_phase_cost.txt"))
raw = [] if not HAVE_PHASES else [l.rstrip() for l in open(os.path.join(G, f"{R}_phase.txt")) if l.startswith("stop ")]
def phases():
    rows = {}
    for l in raw:
        t = l.split(); d = {"ms": float(t[6])}
        for i in range(7, len(t) - 1, 2):
            d[t[i]] = float(t[i + 1])
        rows[int(t[1])] = d
    names = ["camera block (texcoord, RNG init, jitter, ray)", "+ camera march", "+ shading of the bounce (emission, normal, material, next ray, G-buffer)",
             "+ light draw and its term", "+ shadow march", "+ shadow test, blend, store = the product kernel"]
    tot = rows[0]
    mix = lambda d, inst: [100 * d[k] / inst for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")]
    table = [f"{'phase':<78} {'ms':>6} {'share':>6} {'VALU inst':>9} {'share':>6} {'lanes':>6} {'FMA':>5} {'MUL':>5} {'ADD':>5} {'TRANS':>5} {'other':>5}"]
    prev = {k: 0.0 for k in tot}
    for n, name in zip((1, 2, 3, 4, 5, 0), names):
        r = dict(rows[n])
        if n == 5: r["ms"] = min(r["ms"], tot["ms"])  # the diagnostic exit of stop 5 stores more than the product kernel does
        d = {k: r[k] - prev[k] for k in r}
        inst = d["SQ_INSTS_VALU"]; f, m, a, t = mix(d, inst)
        table.append(f"{name:<78} {max(d['ms'], 0):6.3f} {100 * max(d['ms'], 0) / tot['ms']:5.1f}% {inst:9.3g} {100 * inst / tot['SQ_INSTS_VALU']:5.1f}% "
                     f"{d['SQ_THREAD_CYCLES_VALU'] / (inst * 64):6.3f} {f:4.0f}% {m:4.0f}% {a:4.0f}% {t:4.1f}% {100 - f - m - a - t:4.0f}%")
        prev = r
    inst = tot["SQ_INSTS_VALU"]; f, m, a, t = mix(tot, inst)
    table.append(f"{'whole kernel':<78} {tot['ms']:6.3f} {'':>6} {inst:9.3g} {'':>6} {tot['SQ_THREAD_CYCLES_VALU'] / (inst * 64):6.3f} {f:4.0f}% {m:4.0f}% {a:4.0f}% {t:4.1f}% {100 - f - m - a - t:4.0f}%")
    old = rd(P, f"{R}_phase_cost.txt")
    open(os.path.join(P, f"{R}_phase_cost.txt"), "w").write(old[:old.index("stop 1 kernel ms")] + "\n".join(raw) + "\n\n" + "\n".join(table) + "\n\n" + old[old.index("Reading."):])
    print("\n".join(table))

    # ---- C4 / a C5 stripe by phase
    old = rd(P, f"{R}_phase_tables_c4_c5.txt")
    mid = "\nC5, rank 0's stripes of 8"
    mid_line = old[old.index(mid):].split("\n")[1]
    open(os.path.join(P, f"{R}_phase_tables_c4_c5.txt"), "w").write(old[:old.index("stop 1:")] + rd(G, f"{R}_phase_c4.txt") + "\n" + mid_line + "\n" + rd(G, f"{R}_phase_c5s.txt"))
if HAVE_PHASES:
    phases()

# ---- bench lines, tables
open(os.path.join(P, f"{R}_bench_default.json"), "w").write(rd(G, f"{R}b", "bench_default.json"))
open(os.path.join(P, f"{R}_bench_workloads.txt"), "w").write(rd(G, f"{R}b", "bench_workloads.txt"))
for name, src, marker in ((f"{R}_time_all.txt", "time_all.txt", "workload "), (f"{R}_shard_emulation.txt", "shard_emulation.txt", "N=1:"), (f"{R}_present_by_parts.txt", "present_by_parts.txt", "present 3840x2160")):
    text = header(name, marker) + body(os.path.join(f"{R}b", src), marker)
    open(os.path.join(P, name), "w").write(text)
last = lambda path: [l for l in open(os.path.join(G, f"{R}b", path)) if l.startswith('{"metric"')][-1]
l1, l2 = last("bench_ranks_sharing_dof.txt"), last("bench_ranks_sharing.txt")
name = f"{R}_bench_ranks_sharing_one_gpu.txt"
text = header(name, '{"metric"') + l1 + l2
open(os.path.join(P, name), "w").write(text)
print("frame_check with / without depth of field:", json.loads(l1)["frame_check"], json.loads(l2)["frame_check"])

# ---- the fuzz log
F = os.path.join(G, f"{R}_fuzz")
name = f"{R}_fuzz_log.txt"
txt = (header(name, rd(P, name).split("\n\n", 1)[1][:20]) + rd(F, "log.txt") + "\n-- every randomised test of tests/test_gpu_parity.py under seeds 1, 2, 3 (RM_RANDOM_JOBS=2000 RM_RANDOM_SCENES=1500 RM_RANDOM_JOBS2=1000)\n"
       + rd(F, "log_seeds.txt") + "\n-- GL-stack jobs (RM_RANDOM_GL_JOBS=8000, seed 9)\n" + rd(F, "log_gl.txt") + "\n-- tools/dbg/abuse_fuzz.py 4000, seeds 41-44\n" + rd(F, "log_abuse.txt"))
open(os.path.join(P, name), "w").write(txt)
bad = [l for l in txt.splitlines() if "failed" in l or "error" in l.lower()]
print("fuzz log:", "ALL GREEN" if not bad else bad, "(addenda appended by hand to the previous log are not carried over)")
