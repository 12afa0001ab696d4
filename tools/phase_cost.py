#!/usr/bin/env python3
"""Builds the diagnostic variants of the library that leave the pixel kernel after phase n (RM_DIAG_STOP), for
tools/phase_cost.sh: cumulative time and VALU instructions per phase of the headline frame."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rm_build", os.path.join(ROOT, "raymarching-engine_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for n in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else range(1, 6):
    print(b.build_native(force=True, extra=(f"-DRM_DIAG_STOP={n}",), out=os.path.join(ROOT, "tools", f"_exp_stop{n}.so"), tag=f"_stop{n}"))
