#!/usr/bin/env python
"""Round 6 A/B: two co-resident workgroups per CU for the 32-wide growth launches (WSR_CT_DIET, WSR_CT_NARROW_M).

Single launches through the C-ABI (tools/bench_conv.py cases), batch 1 and batch 4, each variant on the same device in
one process:   python tools/tuning/r06_ab_diet.py > gpurun_out/r06_b_ab_diet.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_conv as bc  # noqa: E402

LR = (32, 32, 128)


def cases(B):
    for i in (1, 2, 3):
        bc.conv_case(f"B={B} grow {32 * i}->32", 32 * i, 32, (3, 3, 3), LR, 256, 256, 128 + 32 * i, B=B)
    for i in (1, 2, 3):
        bc.conv_case(f"B={B} window dgrad {32 * i}->32", 32 * i, 32, (3, 3, 3), LR, 256, 256, 128, what="dgrad", B=B)


if __name__ == "__main__":
    variants = [("baseline", {}), ("diet", {"WSR_CT_DIET": "1"}), ("M256", {"WSR_CT_NARROW_M": "256"}),
                ("diet+M256", {"WSR_CT_DIET": "1", "WSR_CT_NARROW_M": "256"})]
    for rep in range(2):
        for name, env in variants:
            for k in ("WSR_CT_DIET", "WSR_CT_NARROW_M"):
                os.environ.pop(k, None)
            os.environ.update(env)
            bc.o._lib.lib().wsr_reload_env()
            print(f"---- {name} (pass {rep})", flush=True)
            for B in (1, 4):
                cases(B)
