"""Builds libhip_raymarch.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
LIB = HERE / "libhip_raymarch.so"
ARCH = "gfx950"

# -fno-slp-vectorize: hipcc otherwise packs the scalar f32 chains of the distance estimators into
# v_pk_*_f32 plus register shuffles, which is 7 % slower on the headline kernel (measured on MI355X)
# -fvisibility=hidden: the library exports the RM_API entry points of include/hip_raymarch.h and nothing else
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-fvisibility=hidden"]
# Both kernel units are contract-off: an FMA exists only where the code asks for one (rm_device.hpp FM::fma).
# Neither gets -fno-hip-fp32-correctly-rounded-divide-sqrt: plain '/' and
# sqrtf stay IEEE in both (the random stream relies on it); the fast build
# asks for v_rcp_f32 / v_sqrt_f32 explicitly where it wants them.
UNITS = {
    "rm_strict": ["-ffp-contract=off"],
    "rm_fast": ["-ffp-contract=off"],
    "rm_glstack": ["-ffp-contract=off"],
    "rm_api": [],
}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sources_mtime() -> float:
    files = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + [CSRC / "exports.map", HERE.parent / "include" / "hip_raymarch.h"]
    return max(f.stat().st_mtime for f in files)


def build_native(force: bool = False, verbose: bool = False, extra=(), out: Path = None, tag: str = "") -> Path:
    """`extra`/`out`/`tag` build an experiment variant next to the product library."""
    global OBJ
    lib = Path(out) if out else LIB
    if not force and lib.exists() and lib.stat().st_mtime >= _sources_mtime():
        return lib
    obj = CSRC / ("_obj" + tag)
    obj.mkdir(exist_ok=True)
    cc = hipcc()

    def compile_unit(name):
        cmd = [cc, *COMMON, *UNITS[name], *extra, "-c", str(CSRC / f"{name}.hip"), "-o", str(obj / f"{name}.o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_unit, UNITS))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + str(CSRC / "exports.map"), "-o", str(lib)] + [str(obj / f"{n}.o") for n in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


# The tests' second implementation of the per-pixel program -- the wavefront pipeline (csrc/rm_wavefront.inc: the same program cut
# at its marches into queue-driven stages) -- is not in the product library: the same sources with -DRM_WITH_WAVEFRONT=1 make
# tests/_xcheck/libhip_raymarch_xcheck.so, which only tests/ (and the tools that time both implementations) load.
XCHECK_LIB = HERE.parent / "tests" / "_xcheck" / "libhip_raymarch_xcheck.so"


def build_crosscheck(force: bool = False, verbose: bool = False) -> Path:
    XCHECK_LIB.parent.mkdir(parents=True, exist_ok=True)
    return build_native(force=force, verbose=verbose, extra=("-DRM_WITH_WAVEFRONT=1",), out=XCHECK_LIB, tag="_xcheck")


def build_node_addon(force: bool = False) -> Path:
    """The N-API addon of the JS host (js/rm_napi.cc), against the system Node headers."""
    src, out = HERE / "js" / "rm_napi.cc", HERE / "js" / "rm_napi.node"
    inc = Path("/usr/include/node")
    if not (inc / "node_api.h").exists():
        raise RuntimeError("Node N-API headers not found (/usr/include/node)")
    if not force and out.exists() and out.stat().st_mtime >= max(src.stat().st_mtime, LIB.stat().st_mtime):
        return out
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-I", str(inc), str(src), "-o", str(out),
                    "-L", str(HERE), "-lhip_raymarch", "-Wl,-rpath,$ORIGIN/.."], check=True)
    return out


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_crosscheck(force="--force" in sys.argv, verbose=True))
    print(build_node_addon(force="--force" in sys.argv))
