"""CPU oracle for the MQ-Det / GLIP vision-language inference forward.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 *restatement* of the
reference algorithm (YifanXu74/MQ-Det @ /root/reference), written function by function
from the reference files cited in each docstring.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / reported CPU baseline -- never as part of the product path
(``mq_det_amd``), which must fail loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * Swin, FPN, GCP blocks (MaskedCrossAttention / GatedCrossAttentionBlock / PreSelect),
    BiMultiHeadAttention / BiAttentionBlock, the clamped BERT layer, DYReLU, DyConv wiring,
    VLDyHead, BoxCoder, anchors and the ATSS post-processor are pinned against outputs of the
    reference's own Python classes imported in the build container (stubs for the
    missing third-party modules only) -- fixtures in tests/golden/, generator
    oracle/gen_golden.py.
  * DCNv2 (modulated deformable conv) and ml_nms exist in the reference only as CUDA
    (.cu) sources that cannot be built here: restated from the .cu files and checked
    by known-answer properties (zero offsets == F.conv2d, integer shifts, brute-force
    NMS).  Those two ops are "parity unpinned" in the strict sense.
"""
from .spec import Spec, tiny_spec, glip_t_spec, glip_l_spec, tiny_l_spec  # noqa: F401
