#!/usr/bin/env python
"""Do the kernels of two HIP streams actually run at the same time?

    python tools/tuning/overlap.py kernel_trace.csv [--side-substr wgrad_tile_kernel]

Reads a rocprofv3 --kernel-trace CSV of a run with WSR_WGRAD_STREAM >= 2, splits the launches by Queue_Id, and reports for
the side queue (the one that only carries filter gradients / the ordered reduce): how much of its kernel time lies
inside the busy time of the main queue (overlap), and what happened to the main queue's kernels that ran next to it
(mean duration per kernel name while overlapped vs alone).  Time-slicing shows as: overlapped main kernels take about
their own time PLUS the side kernel's share - no net gain."""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    byq = defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append(r)
    qs = sorted(byq, key=lambda q: -len(byq[q]))
    print("queues:", {q: len(byq[q]) for q in qs})
    if len(qs) < 2:
        print("one queue only: nothing ran on a second stream")
        return
    main_q, side_q = qs[0], qs[1]
    side = sorted(byq[side_q], key=lambda r: r["s"])
    mainr = sorted(byq[main_q], key=lambda r: r["s"])
    # overlap of every main kernel with the union of side intervals
    iv = [(r["s"], r["e"]) for r in side]
    j = 0
    alone, shared = defaultdict(list), defaultdict(list)
    side_total = sum(e - s for s, e in iv)
    ov_total = 0
    for r in mainr:
        while j < len(iv) and iv[j][1] <= r["s"]:
            j += 1
        k, ov = j, 0
        while k < len(iv) and iv[k][0] < r["e"]:
            ov += max(0, min(r["e"], iv[k][1]) - max(r["s"], iv[k][0]))
            k += 1
        ov_total += ov
        d = r["e"] - r["s"]
        (shared if ov > 0.5 * d else alone)[short(r["Kernel_Name"])].append(d)
    print(f"side queue: {len(side)} kernels, {side_total / 1e6:.2f} ms; of it inside main-queue kernels: "
          f"{ov_total / 1e6:.2f} ms ({100.0 * ov_total / max(side_total, 1):.0f} %)")
    print(f"{'main-queue kernel':62s} {'alone us':>9s} {'n':>5s} {'next to side us':>16s} {'n':>5s} {'ratio':>6s}")
    for n in sorted(set(alone) | set(shared), key=lambda n: -sum(shared.get(n, [0]))):
        a, s_ = alone.get(n, []), shared.get(n, [])
        if not s_ or not a:
            continue
        ma, ms = sum(a) / len(a) / 1e3, sum(s_) / len(s_) / 1e3
        print(f"{n:62s} {ma:9.1f} {len(a):5d} {ms:16.1f} {len(s_):5d} {ms / ma:6.2f}")
    sd = defaultdict(list)
    for r in side:
        sd[short(r["Kernel_Name"])].append(r["e"] - r["s"])
    for n, v in sd.items():
        print(f"side: {n:56s} {sum(v) / len(v) / 1e3:9.1f} us x {len(v)}")
    span = max(r["e"] for r in rows) - min(r["s"] for r in rows)
    print(f"span {span / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
