#!/usr/bin/env python3
"""tools/scan_ring_registers.py -- does the compiler keep its hands off the weight ring?

The kernels that inline the hand-scheduled k-loop (hz_mlp_dev.h: k_search*, k_mlp_recurrent16) keep weight fragments in flight in
the fixed registers v[96:127], named only inside inline asm; the kernels are compiled with amdgpu_num_vgpr(96), which is a
budget, not a guarantee: under register pressure the allocator was seen to place an address computation in v[96:99] (r03, the
in-turn 32-row kernel).  This compiles the two translation units to assembly -- with the compiler and the flags csrc/Makefile builds the library with
(`make print-flags`) -- and fails if any instruction OUTSIDE an inline-asm
block of those kernels names a register >= v96.  __graft_entry__.build() runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hanabizero_amd", "csrc")
PAT = re.compile(r"\bv(9[6-9]|1[01][0-9]|12[0-7])\b|v\[(9[6-9]|1[01][0-9]|12[0-7]):")


def build_flags():
    """(hipcc, flags) the library itself is built with, asked of csrc/Makefile (`make print-flags`): one flag list, not two."""
    ask = lambda target: subprocess.check_output(["make", "-s", "--no-print-directory", "-C", SRC, target], text=True).split()
    hipcc, flags = ask("print-hipcc"), ask("print-flags")
    assert len(hipcc) == 1 and "--offload-arch=gfx950" in flags and "-O3" in flags, (hipcc, flags)
    return hipcc[0], flags


def scan():
    bad, seen = [], 0
    hipcc, flags = build_flags()
    with tempfile.TemporaryDirectory() as tmp:
        for unit in ("hz_search.hip", "hz_mlp.hip"):
            out = os.path.join(tmp, unit + ".s")
            run = subprocess.run([hipcc] + flags + ["-w", "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, unit)],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if run.returncode != 0:
                raise RuntimeError("scan_ring_registers: %s did not compile to assembly:\n%s" % (unit, run.stdout[-4000:]))
            txt = open(out).read()
            # (k_search_pairs, the fp16-pair build, runs the compiler-scheduled k-loop: no ring in fixed registers, all 128 are the compiler's)
            for name in re.findall(r"^(_Z\w*k_(?:search(?!_pairs)|mlp_recurrent16)\w*):", txt, re.M):
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
