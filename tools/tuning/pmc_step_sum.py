#!/usr/bin/env python
"""Per-kernel HBM traffic and MFMA occupancy of one train step from the three counter passes of pmc_step.sh.

    python tools/tuning/pmc_step_sum.py NAME STATS_CSV > profiles/NAME_hbm_kernels.txt   (+ profiles/<round>_hbm_traffic.json,
                                                                                         <round> = NAME up to its first "_")

Durations come from the un-instrumented ``rocprofv3 --kernel-trace --stats`` run (STATS_CSV): counter passes
serialise dispatches and run at a lower clock.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of wide
coalesced reads at 64 bytes; MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact for 16-byte stores.  Both count the
L2's memory-side requests, Infinity-Cache hits included - "traffic" is what left the L2, not what reached HBM.
"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0].replace(", BF16>", ">")  # (round 4: the tile kernel's element type is a template parameter)
    # round 5: conv_tile_kernel<..., BF16, WK, SIMPLE> - the plain instantiation keeps its old name, the others say what they are;
    # conv_thin3_kernel<..., PS> likewise (paired stores)
    n = re.sub(r"(conv_tile_kernel<[^>]*), BF16, 1, 0>", r"\1>", n)
    n = re.sub(r"(conv_tile_kernel<[^>]*), BF16, (\d+), (\d+)>", lambda m: m.group(1) + (f", WK{m.group(2)}" if m.group(2) != "1" else "")
               + (", SIMPLE" if m.group(3) != "0" else "") + ">", n)
    n = re.sub(r"(conv_thin3_kernel<[^>]*), false>", r"\1>", n)
    n = re.sub(r"(conv_thin3_kernel<[^>]*), true>", r"\1, PS>", n)
    return n


def load(path, counters):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def main(name, stats_csv):
    fetch = load(f"gpurun_out/{name}_fetch.csv", {"FETCH_SIZE"})
    write = load(f"gpurun_out/{name}_write.csv", {"WRITE_SIZE"})
    mfma = load(f"gpurun_out/{name}_mfma.csv", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES",
                                                "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"})
    dur = {short(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(stats_csv))}
    m = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
    # algorithmic bytes per launch at C3' (B=1, LR 32x32x128, HR 128^3, bf16) where one number describes all launches
    v, V = 32 * 32 * 128, 128 ** 3
    algo = {
        "conv_tile_kernel<8, 1, 4, 9, 2, false, false>": 2 * V * 144 * 2 + 125 * 144 * 144 * 2,
        # LFF forward: 256 channels in, 128 out; the block shortcut is the lane's own K fragment (no second read); every
        # third launch (last block of an RRDB) also reads the RRDB shortcut: average over the launches
        "conv1x1_v2_kernel<8, 8, false, true>": v * (256 + 128) * 2 + 256 * 128 * 2 + v * 128 * 2 // 3,
        "conv_slide_fwd_kernel<5, 5, 18>": V * (144 * 2 + 15 * 4) + 125 * 144 * 3 * 2,
        "conv_slide_dgrad_kernel<5, 5, 9, 0>": V * (16 * 2 + 2 * 144 * 2) + 125 * 144 * 3 * 2,
        "conv1x1_v2_kernel<16, 4, true, true>": v * (128 + 128 + 256 + 32) * 2 + 256 * 128 * 2,
        "wgrad_tile_kernel<8, 1, 8, true>": v * (256 + 128) * 2,
        "wgrad_tile_kernel<3, 16, 1, true>": 2 * V * 144 * 2,
        "physics_stats_kernel": V * 7 * 4,
        "physics_residual_kernel": V * (7 + 9) * 4,
        "physics_adjoint_kernel": V * (9 + 7 + 3) * 4,
        "plane_sum_kernel": V * 3 * 4,
        "zfold_kernel": V * (15 + 3) * 4,
        # conv_thin.hip (round 4): terrain convs, feature conv, the discriminator's first conv (two samples per launch)
        "conv_thin3_kernel<8, 1, 1, 2, 32, 8>": V * 17 * 2,
        "conv_thin3_kernel<16, 1, 1, 2, 32, 8>": V * 32 * 2,
        "conv_thin3_kernel<8, 8, 1, 4, 16, 8, PS>": v * 132 * 2,
        "conv_thin3_kernel<8, 1, 2, 2, 32, 4>": 2 * V * 35 * 2,
    }
    rows = []
    for k, (us, calls) in dur.items():
        f = 2 * m(fetch.get(k, {}).get("FETCH_SIZE", [])) * 1024  # KB -> bytes, gfx950 correction
        w = m(write.get(k, {}).get("WRITE_SIZE", [])) * 1024
        c = mfma.get(k, {})
        busy, gui = m(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), m(c.get("GRBM_GUI_ACTIVE", []))
        util = busy / (gui / 8 * 1024) if gui == gui and gui > 0 else float("nan")
        wc = m(c.get("SQ_WAVE_CYCLES", []))
        rows.append((us * calls, k, calls, us, f, w, (f + w) / (us * 1e-6) / 1e12 if us else 0, util,
                     m(c.get("SQ_WAIT_ANY", [])) / wc if wc == wc and wc else float("nan"), algo.get(k)))
    rows.sort(reverse=True)
    print(f"# {name}: per-kernel L2-side traffic (2 x FETCH_SIZE + WRITE_SIZE, bytes per launch), rate over the")
    print("# un-instrumented duration, MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs),")
    print("# share of wave-cycles parked in s_waitcnt / barriers (SQ_WAIT_ANY).  algo = algorithmic bytes per launch.")
    print(f"{'kernel':58s} {'calls':>6s} {'avg us':>9s} {'fetch MB':>10s} {'write MB':>10s} {'TB/s':>6s} {'mfma':>6s} "
          f"{'wait':>6s} {'algo MB':>9s} {'algo TB/s':>9s}")
    for tot, k, calls, us, f, w, rate, util, wait, ab in rows[:60]:
        a = f"{ab / 1e6:9.1f} {ab / (us * 1e-6) / 1e12:9.2f}" if ab else f"{'':9s} {'':9s}"
        print(f"{k[:58]:58s} {calls:6d} {us:9.1f} {f / 1e6:10.1f} {w / 1e6:10.1f} {rate:6.2f} {util:6.2f} {wait:6.2f} {a}")
    traffic = {}
    for key, k in (("hr0", "conv_tile_kernel<8, 1, 4, 9, 2, false, false>"), ("lff_fwd", "conv1x1_v2_kernel<8, 8, false, true>"),
                   ("hr1_fwd", "conv_slide_fwd_kernel<5, 5, 18>"), ("hr1_dgrad", "conv_slide_dgrad_kernel<5, 5, 9, 0>"),
                   ("terrain0_fwd", "conv_thin3_kernel<8, 1, 1, 2, 32, 8>"),
                   # (forward and input gradient of terrain_convs.1 are launches of ONE kernel: their mean)
                   ("terrain1_fwd", "conv_thin3_kernel<16, 1, 1, 2, 32, 8>"),
                   ("terrain1_dgrad", "conv_thin3_kernel<16, 1, 1, 2, 32, 8>"),
                   ("feature_fwd", "conv_thin3_kernel<8, 8, 1, 4, 16, 8, PS>"), ("d0_fwd", "conv_thin3_kernel<8, 1, 2, 2, 32, 4>")):
        r = [x for x in rows if x[1] == k]
        if r:
            traffic[key] = r[0][4] + r[0][5]
    import subprocess
    try:  # the commit these counter passes were taken at (bench.py prints it beside `traffic`)
        traffic["_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        traffic["_commit"] = None
    traffic["source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '{name}', 2 x FETCH + WRITE, bytes per launch"
    json.dump(traffic, open(f"profiles/{name.split('_')[0]}_hbm_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
