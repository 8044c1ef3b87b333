"""mq_det_amd -- MI355X (gfx950) native implementation of the MQ-Det / GLIP vision-language
inference forward (Swin + BERT/GCP + VLDyHead + ATSS post-processing) behind the reference's
`build_detection_model(cfg)` / `GeneralizedVLRCNN_New.forward(images, captions=, positive_map=)` API.

The hot ops are hand-written HIP kernels in csrc/ (C ABI: include/mqdet_hip.h, ctypes binding: ops.py);
PyTorch is used for device memory, streams, library GEMMs/convs and torch.distributed only.
"""
# Process environment the forward is TUNED for (never set by importing this package -- ADVICE r4: an import must not change the queue
# behaviour of unrelated HIP users of the process, and the variable has no effect once the HIP runtime is initialised).  The forward runs
# its text / level / image chains as parallel branches of one HIP graph; the ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4), and branches that share a queue serialise.  8 queues: +2.8 % / +3.4 % end to end in two A/B runs on the MI355X
# (profiles/r04_call5_lanes_ab.txt; 16 queues lose 30 %).  bench.py and INTEGRATION.md's launch line export it before the first device call.
RECOMMENDED_ENV = {"GPU_MAX_HW_QUEUES": "8"}

from .config import CfgNode, get_cfg  # noqa: F401
from .structures import BoxList, ImageList, to_image_list, cat_boxlist  # noqa: F401


def build_detection_model(cfg, **kwargs):
    """Same entry point as maskrcnn_benchmark.modeling.detector.build_detection_model
    (reference modeling/detector/__init__.py:9-14)."""
    from .modeling.detector import build_detection_model as _b
    return _b(cfg, **kwargs)
