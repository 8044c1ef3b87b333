"""TEST INFRASTRUCTURE ONLY: compile the kernel sources of mq_det_amd/csrc/ for the HOST against tests/simt/include (a stand-in
for the HIP runtime that runs every thread as a fiber) into tests/simt/_build/libmqdet_simt.so -- the same C ABI
(include/mqdet_hip.h) over host pointers.  Two textual rewrites, nothing else:

  extern __shared__ [aligned] T name[];   ->   T* name = (T*)simt::dyn_smem();
  asm volatile( ... )                      ->   SIMT_ASM( ... )        (gfx950 assembly: waits, cache warm-up loads -- no effect on results)

The product never loads this library (mq_det_amd/ops.py binds mq_det_amd/lib/libmqdet_hip.so only and raises without a GPU)."""
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mq_det_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libmqdet_simt.so")
CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-fno-strict-aliasing", "-Wno-everything",
         # the device accepts 16-byte global loads at 8-byte alignment (e.g. rows of 64-bit mask words); x86 movaps does not
         "-fmax-type-align=2",
         "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include")]

# not built with fp32 operands: the MQ-GroundingDINO / query-extraction operators (their inputs may already be fp32 -- the two template
# arguments would coincide) and the superseded first Swin MLP kernel
F32_SKIP = ("roi_align.cpp", "nms2.cpp", "post2.cpp")
_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w\s]*?)\s+(\w+)\s*\[\s*\]\s*;")




def rewrite(text):
    text = _SHARED.sub(lambda m: f"{m.group(1)}* {m.group(2)} = ({m.group(1)}*)simt::dyn_smem();", text)
    text = text.replace('#include "../../include/', '#include "')                  # -I <repo>/include
    return re.sub(r"\basm\s+volatile\s*\(", "SIMT_ASM(", text)


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    return h.hexdigest()


def build(force=False, verbose=False, sources=None):
    """-> path of the emulation library (rebuilt when any input changed)."""
    os.makedirs(BUILD, exist_ok=True)
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    inputs = [os.path.join(CSRC, f) for f in names] + [os.path.join(HERE, "simt_runtime.cpp"), os.path.join(HERE, "selftest.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                                                        os.path.abspath(__file__)]
    stamp, dig = os.path.join(BUILD, "stamp"), _digest(inputs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    src_dir = os.path.join(BUILD, "src")
    os.makedirs(src_dir, exist_ok=True)
    for f in names:
        with open(os.path.join(CSRC, f)) as fh:
            text = rewrite(fh.read())
        with open(os.path.join(src_dir, f.replace(".hip", ".cpp")), "w") as fh:
            fh.write(text)
    procs, objs = [], []
    units = [os.path.join(src_dir, f.replace(".hip", ".cpp")) for f in names if f.endswith(".hip")] + [os.path.join(HERE, "simt_runtime.cpp"), os.path.join(HERE, "selftest.cpp")]
    for u in units:
        # like mq_det_amd/build.py: every kernel source twice, fp16 and -DMQ_BF16 (the *_bf16 entry points)
        for suffix, defs in (("", []), ("_bf16", ["-DMQ_BF16"]), ("_f32", ["-DMQ_F32"])):
            if suffix and os.path.basename(u) in ("api.cpp", "simt_runtime.cpp", "selftest.cpp"):
                continue
            if suffix == "_f32" and os.path.basename(u) in F32_SKIP:
                continue
            obj = os.path.join(BUILD, os.path.basename(u).replace(".cpp", suffix + ".o"))
            objs.append(obj)
            procs.append((u + suffix, subprocess.Popen([CXX, *FLAGS, *defs, "-I", src_dir, "-c", u, "-o", obj], stdout=subprocess.PIPE,
                                                       stderr=subprocess.STDOUT)))
    for u, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"simt build failed on {u}:\n{out.decode()[-6000:]}")
    subprocess.check_call([CXX, "-shared", "-fPIC", *objs, "-o", LIB, "-lm"])
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
