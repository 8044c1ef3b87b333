#!/usr/bin/env python3
"""Static view of what a kernel's loops WAIT for (no GPU): compiles one .hip to gfx950 ISA and prints, per basic block of a kernel, the
sequence of memory instructions, MFMAs, barriers and s_waitcnt values, or a per-kernel histogram of the waits inside the MFMA loops.

    python tools/isa_loop_waits.py dcn_fused.hip --kernel 'dcn_igemm8_kernel<16, 0, 1, false, false, true>' [-D MQ_F32]
    python tools/isa_loop_waits.py swin_mlp2.hip --hist

Legend of the sequence: D = global -> LDS copy (buffer_load ... lds / global_load_lds), G = global load, r / w = ds_read / ds_write,
M = MFMA, [vm(n)] [lg(n)] = s_waitcnt vmcnt / lgkmcnt, |B| = s_barrier, <br> = branch.

How round 6 used it: a loop whose every wait is vmcnt(0) / lgkmcnt(0) has no counted waits -- for the DCNv2 kernel with
LDS-copied weights and the Swin MLP kernel the cause was the FLAT-encoded `global_load_lds`, which hipcc's wait-count pass books as a flat
access ("pending flat": from then on every wait is forced to zero); `r [lg(0)] M M M M` repeated is a fragment read that is waited for on
the spot (window-attention projections: a two-deep ring of reads)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = {"dcn_fused.hip": ["-fno-slp-vectorize"]}


def isa(src, defs=()):
    path = src if os.path.isabs(src) else os.path.join(ROOT, "mq_det_amd", "csrc", src)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA.get(os.path.basename(path), []), *[f"-D{d}" for d in defs],
                        "-S", "--cuda-device-only", path, "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(txt):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        yield dem, m.group(2)


def sequence(body):
    out = []
    for l in (x.strip() for x in body.split("\n")):
        if not l or l.startswith(";"):
            continue
        if re.match(r"\.LBB\d+_\d+:", l):
            out.append("\n" + l.split(";")[0].strip())
        elif "global_load_lds" in l or (l.startswith("buffer_load") and l.endswith("lds")):
            out.append("D")
        elif l.startswith(("global_load", "buffer_load")):
            out.append("G")
        elif l.startswith("ds_read"):
            out.append("r")
        elif l.startswith("ds_write"):
            out.append("w")
        elif "mfma" in l:
            out.append("M")
        elif l.startswith("s_waitcnt"):
            out.append("[" + l.replace("s_waitcnt ", "").replace("vmcnt", "vm").replace("lgkmcnt", "lg") + "]")
        elif l.startswith("s_barrier"):
            out.append("|B|")
        elif l.startswith("s_cbranch"):
            out.append("<br>")
    return " ".join(out)


def histogram(body, min_mfma=8):
    lg, vm, nm = collections.Counter(), collections.Counter(), 0
    for b in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M):
        k = len(re.findall(r"v_mfma", b))
        if k < min_mfma:
            continue
        nm += k
        for w in re.findall(r"s_waitcnt ([^\n]*)", b):
            for c, v in re.findall(r"(vmcnt|lgkmcnt)\((\d+)\)", w):
                (lg if c == "lgkmcnt" else vm)[int(v)] += 1
    return nm, lg, vm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--kernel", default="", help="substring of the demangled kernel name")
    ap.add_argument("-D", dest="defs", action="append", default=[])
    ap.add_argument("--hist", action="store_true")
    a = ap.parse_args()
    for name, body in kernels(isa(a.src, a.defs)):
        if a.kernel and a.kernel not in name:
            continue
        if a.hist:
            nm, lg, vm = histogram(body)
            if nm:
                print(f"{name[:88]:90s} mfma={nm:4d}  lgkmcnt(0) {lg[0]:3d} of {sum(lg.values()):3d}   vmcnt(0) {vm[0]:3d} of {sum(vm.values()):3d}")
        else:
            print("== " + name)
            print(sequence(body))


if __name__ == "__main__":
    main()
