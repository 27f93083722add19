#!/usr/bin/env python
"""Registers / LDS / scratch of every kernel in the built library (code-object metadata, no GPU needed):

    python tools/tuning/kernel_resources.py [path/to/libwindsr_hip.so] [name filter]

Reads the .hip_fatbin section, un-bundles the gfx950 code objects and prints their AMDGPU metadata notes.
Used to reason about co-residency: 160 KB of LDS and 512 VGPRs per SIMD lane per CU."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else \
        os.path.join(ROOT, "gan_sr_wind_field_amd", "csrc", "libwindsr_hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        # the section is a sequence of clang offload bundles, each holding ELF code objects: cut at the ELF magics
        starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01", blob)]
        rows = []
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(blob)
            co = os.path.join(td, f"co{i}.o")
            open(co, "wb").write(blob[s:e])
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                def g(f):
                    m = re.search(r"\." + f + r":\s+(\S+)", k)
                    return m.group(1) if m else "?"
                rows.append((g("symbol").replace(".kd", ""), k.split()[0], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                             g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
        names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True,
                               text=True).stdout.splitlines()
        for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
            n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""), flags=re.S)
            if flt and flt not in n:
                continue
            print(f"agpr={r[1]:>3} vgpr={r[2]:>3} sgpr={r[3]:>3} lds={r[4]:>6} scratch={r[5]:>4} wg={r[6]:>4}  {n[:120]}")


if __name__ == "__main__":
    main()
