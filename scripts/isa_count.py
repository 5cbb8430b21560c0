#!/usr/bin/env python3
"""Instruction audit of a kernel's loops: compiles one .hip file to gfx950 assembly and counts, for every
loop of the named kernel with more than 100 instructions, the instructions by mnemonic (VALU total first).
usage: isa_count.py itsxpress_amd/csrc/k_float.hip _ZN4itsx13k_filters_fwdILi12EEEvNS_9FloatArgsEi
       (kernel names: grep '^_Z.*:' on the .s file this writes to /tmp/isa_count/)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    os.makedirs("/tmp/isa_count", exist_ok=True)
    out = os.path.join("/tmp/isa_count", os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fgpu-flush-denormals-to-zero",
                    "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    s = [i for i, l in enumerate(lines) if l.startswith(kernel + ":")][0]
    e = s
    while not lines[e].startswith(".Lfunc_end"):
        e += 1
    labels = {}
    for i in range(s, e):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    for i in range(s, e):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        a = labels[m.group(1)]
        c = collections.Counter()
        for l in lines[a:i + 1]:
            mm = re.match(r"\s+([a-z_0-9]+)\s", l)
            if mm:
                c[mm.group(1)] += 1
        tot = sum(c.values())
        if tot < 100:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("loop at lines %d-%d: %d instructions, %d VALU" % (a, i, tot, valu))
        print("  " + ", ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda x: -x[1])[:24]))
    for l in lines[s:e + 40]:
        if re.match(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy):", l):
            print(l)


if __name__ == "__main__":
    main()
