#!/usr/bin/env python
"""A/B of tuning switches on single launches: runs tools/bench_conv.py cases once per environment setting.

    python tools/tuning/ab_env.py "WSR_CT_W4=0,1,2,4" hr0 lr grow up
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_conv  # noqa: E402

if __name__ == "__main__":
    var, vals = sys.argv[1].split("=")
    cases = sys.argv[2:]
    for v in vals.split(","):
        if v in ("", "-"):
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        bench_conv.o._lib.lib().wsr_reload_env()  # (the switches are cached per call site)
        print(f"---- {var}={v or '(unset)'}", flush=True)
        for c in cases:
            bench_conv.CASES[c]()
