#!/usr/bin/env python3
"""Static check of the gfx950 ISA of every HIP kernel for SERIALISED global loads (no GPU needed).

    python tools/isa_wait_scan.py [file.hip ...]          default: every .hip under mq_det_amd/csrc

For each kernel: number of global / buffer loads, number of `s_waitcnt vmcnt(0)`, and how many loads are followed by a
vmcnt(0) within a few instructions before the next load is issued ("immediate waits").  A load that is waited on at once is a
full memory round trip during which the wave issues nothing; a kernel whose hot part consists of such pairs is latency-bound no
matter how little data it moves.  How round 2 used it: the MSDeformAttn gather (one guarded load per bilinear corner: 64
serialised round trips per query, 2.38 -> 1.03 ms per launch once the loads were made unconditional and issued together), the
Swin MLP prologue / epilogue (one round trip per 32 channels; `if (live)` / `if (p.delta)` inside the loop and a possible alias
between `out` and `x` kept the compiler from hoisting the loads), the window attention (q / k / v rows block by block, bias rows
one by one behind a wave-uniform branch).  Typical causes: a branch or a store to a possibly aliasing pointer between two loads,
a conversion of the loaded value in the same basic block as the load, per-iteration `if (in_range)` guards.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def scan(path, window=8):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(path))
        txt = open(asm).read()
    rows = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, lines = m.group(1), m.group(2).split("\n")
        loads = [i for i, l in enumerate(lines) if re.search(r"\b(global_load|buffer_load)", l)]
        waits = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt.*vmcnt\(0\)", l)]
        if len(loads) < 4:
            continue
        imm = 0
        for k, i in enumerate(loads):
            nxt = loads[k + 1] if k + 1 < len(loads) else len(lines)
            w = [j for j in waits if i < j < nxt]
            imm += bool(w and w[0] - i <= window)
        rows.append((name, len(loads), len(waits), imm))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "mq_det_amd", "csrc", "*.hip")))
    for f in files:
        for name, nl, nw, imm in scan(os.path.abspath(f)):
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            flag = "  <-- look at this one" if imm >= max(4, nl // 3) else ""
            print(f"{os.path.basename(f):18s} loads={nl:3d} vmcnt(0)={nw:3d} immediate={imm:3d}  {demangled[:90]}{flag}")


if __name__ == "__main__":
    main()
