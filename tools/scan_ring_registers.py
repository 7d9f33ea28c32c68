#!/usr/bin/env python3
"""tools/scan_ring_registers.py -- does the compiler keep its hands off the weight ring?

The kernels that inline the hand-scheduled k-loop (hz_mlp_dev.h: k_search*, k_mlp_recurrent16) keep weight fragments in flight in
the fixed registers v[96:127], named only inside inline asm; the kernels are compiled with amdgpu_num_vgpr(96), which is a
budget, not a guarantee: under register pressure the allocator was seen to place an address computation in v[96:99] (r03, the
in-turn 32-row kernel).  This compiles the two translation units to assembly and fails if any instruction OUTSIDE an inline-asm
block of those kernels names a register >= v96.  __graft_entry__.build() runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hanabizero_amd", "csrc")
PAT = re.compile(r"\bv(9[6-9]|1[01][0-9]|12[0-7])\b|v\[(9[6-9]|1[01][0-9]|12[0-7]):")


def scan():
    bad, seen = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for unit in ("hz_search.hip", "hz_mlp.hip"):
            out = os.path.join(tmp, unit + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                   "-fhip-fp32-correctly-rounded-divide-sqrt", "-w", "-I" + SRC, "-I" + os.path.join(ROOT, "include"),
                                   "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, unit)], stderr=subprocess.DEVNULL)
            txt = open(out).read()
            for name in re.findall(r"^(_Z\w*k_(?:search|mlp_recurrent16)\w*):", txt, re.M):
                i = txt.index("\n" + name + ":")
                body = txt[i:txt.index("s_endpgm", i)]
                seen += 1
                inasm = False
                for line in body.split("\n"):
                    if "ASMSTART" in line:
                        inasm = True
                    if "ASMEND" in line:
                        inasm = False
                        continue
                    if not inasm and PAT.search(line):
                        bad.append((name, line.strip()))
    return seen, bad


if __name__ == "__main__":
    seen, bad = scan()
    for name, line in bad:
        print("%s: %s" % (name, line))
    print("%d kernels scanned, %d compiler-generated instructions touch v96..v127" % (seen, len(bad)))
    sys.exit(1 if bad or seen < 12 else 0)
