"""Build libmqdet_hip.so (gfx950) in-tree with hipcc.  `python -m mq_det_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libmqdet_hip.so")
SOURCES = ["api.hip", "attn.hip", "attn_resident.hip", "attn_text.hip", "bert_attn.hip", "vlfuse_attn.hip", "window_attn.hip", "patch_embed.hip", "gcp.hip", "gcp_fused.hip", "conv_igemm.hip", "conv_small.hip", "conv_small2.hip", "conv_small3.hip", "dcn_fused.hip", "layernorm.hip", "layernorm2.hip", "dyconv.hip", "post.hip", "post2.hip", "align_fused.hip", "nms2.hip", "roi_align.hip", "swin_mlp2.hip", "msda.hip"]
# no fp32-operand twin (include/mqdet_hip.h MQ_F32_TWIN): operators whose inputs may already be fp32, and the sources that only hold fp32 / integer code (one copy, in the fp16 unit)
F32_SKIP = ("roi_align.hip", "nms2.hip", "post2.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# dcn_fused.hip: without the SLP vectoriser the bilinear blend compiles to v_fma_mix_f32 / v_fma_mixlo_f16 (fp16 operands,
# fp32 accumulate, no separate converts) instead of cvt + v_pk_fma_f32 -- 40 % fewer VALU cycles next to the MFMAs
EXTRA_FLAGS = {"dcn_fused.hip": ["-fno-slp-vectorize"],
               # swin_mlp2.hip: the GELU is dealt out over the MFMA steps in pieces; the SLP vectoriser would merge pieces of different
               # steps into v_pk_*_f32 (slower than two scalar ops beside MFMAs, and it undoes the placement)
               "swin_mlp2.hip": ["-fno-slp-vectorize"],
               # attn_resident.hip: the S^T accumulators are consumed by VALU code (softmax): keep them in VGPRs (no v_accvgpr_read per
               # logit) and keep the scalar f32 softmax arithmetic unpacked (packed f32 VALU beside MFMAs costs more than it saves)
               "attn_resident.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               "attn_text.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               "bert_attn.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]}


STAMP = os.path.join(LIB_DIR, "libmqdet_hip.stamp")


def _digest():
    """Content hash of everything the library is built from (sources, headers, this file's flags) -- not modification times: a checkout, a
    copy to the GPU box or a `touch` must neither force nor hide a rebuild."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "mqdet_hip.h"), os.path.abspath(__file__)]
    for d in deps:
        if os.path.isfile(d):
            with open(d, "rb") as f:
                h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _digest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library.  Returns the library path."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    digest = _digest()        # of the sources as they are NOW: an edit made while hipcc runs must leave the stamp stale (round 6: it did not)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    # every kernel source three times: fp16 (the entry points of include/mqdet_hip.h), -DMQ_BF16 (the same kernels with bf16 operands
    # under the *_bf16 entry points, csrc/common.h) and -DMQ_F32 (fp32 operands, *_f32: the precise mode); api.hip (the ABI version) once
    for src in SOURCES:
        for suffix, defs in (("", []), ("_bf16", ["-DMQ_BF16"]), ("_f32", ["-DMQ_F32"])):
            if suffix and src == "api.hip":
                continue
            if suffix == "_f32" and src in F32_SKIP:
                continue
            obj = os.path.join(LIB_DIR, src.replace(".hip", suffix + ".o"))
            objs.append(obj)
            procs.append((src + suffix, subprocess.Popen([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *defs, "-c", os.path.join(CSRC, src), "-o", obj],
                                                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    with open(STAMP, "w") as f:
        f.write(digest)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
