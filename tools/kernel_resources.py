"""Per-kernel register / scratch / spill table of one kernel TU, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

  python tools/kernel_resources.py rm_fast [--grep rm_pixel_kernel] [extra hipcc flags ...]

Compiles raymarching-engine_amd/csrc/<unit>.hip with the product's flags (build.py COMMON + UNITS) and prints one line per
kernel: SGPRs, VGPRs, scratch bytes per lane, SGPR spills, VGPR spills, waves per SIMD, LDS bytes.  Runs without a GPU."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raymarching-engine_amd"))
import build as B  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    unit = sys.argv[1]
    args = sys.argv[2:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    obj = "/tmp/_kr_%s.o" % unit
    cmd = [B.hipcc(), *B.COMMON, *B.UNITS[unit], *args, "-Rpass-analysis=kernel-resource-usage", "-c", str(B.CSRC / f"{unit}.hip"), "-o", obj]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"(?=remark: [^\n]*Function Name)", txt):
        m = re.search(r"Function Name: (\S+)", b)
        if not m:
            continue

        def g(k):
            mm = re.search(k + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1

        rows.append((m.group(1), g("SGPRs"), g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    names = demangle([r[0] for r in rows])
    print("%-78s %5s %5s %7s %6s %6s %5s %6s" % ("kernel", "SGPR", "VGPR", "scratch", "Sspill", "Vspill", "waves", "LDS"))
    for r in rows:
        n = re.sub(r"^void rm::", "", names[r[0]])
        n = re.sub(r"\(rm::KParams\)|\(KParams\)", "", n)
        if pat and pat not in n:
            continue
        print("%-78s %5d %5d %7d %6d %6d %5d %6d" % ((n[:78],) + r[1:]))
    print("object:", obj)


if __name__ == "__main__":
    main()
