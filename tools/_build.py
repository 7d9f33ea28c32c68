"""tools/_build.py -- scratch builds of the HIP library for the diagnostic tools: every source file of the product
(hanabizero_amd/csrc/Makefile's SRCS) compiled with the product's flags, plus per-file extra defines (a diagnostic switch belongs
to ONE translation unit: its device-side stamp arrays are defined in headers), linked into gpurun_out/<name>."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hanabizero_amd", "csrc")


def sources():
    mk = open(os.path.join(SRC, "Makefile")).read()
    line = re.search(r"^SRCS := (.*)$", mk, re.M).group(1)
    return [os.path.join(SRC, t.replace("$(HERE)", "")) for t in line.split()]


def build(name, extra, common=()):
    """extra: {file basename: [flags]}; common: flags for every file.  Returns the path of the shared library."""
    out = os.path.join(ROOT, "gpurun_out", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-fhip-fp32-correctly-rounded-divide-sqrt", "-w", "-I" + SRC, "-I" + os.path.join(ROOT, "include")] + list(common)
    objs, procs = [], []
    for f in sources():
        o = os.path.join(ROOT, "gpurun_out", "%s.%s.o" % (os.path.basename(f), name))
        procs.append(subprocess.Popen(base + list(extra.get(os.path.basename(f), [])) + ["-c", "-o", o, f]))
        objs.append(o)
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out
