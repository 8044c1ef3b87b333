"""mq_det_amd -- MI355X (gfx950) native implementation of the MQ-Det / GLIP vision-language
inference forward (Swin + BERT/GCP + VLDyHead + ATSS post-processing) behind the reference's
`build_detection_model(cfg)` / `GeneralizedVLRCNN_New.forward(images, captions=, positive_map=)` API.

The hot ops are hand-written HIP kernels in csrc/ (C ABI: include/mqdet_hip.h, ctypes binding: ops.py);
PyTorch is used for device memory, streams, library GEMMs/convs and torch.distributed only.
"""
from .config import CfgNode, get_cfg  # noqa: F401
from .structures import BoxList, ImageList, to_image_list, cat_boxlist  # noqa: F401


def build_detection_model(cfg, **kwargs):
    """Same entry point as maskrcnn_benchmark.modeling.detector.build_detection_model
    (reference modeling/detector/__init__.py:9-14)."""
    from .modeling.detector import build_detection_model as _b
    return _b(cfg, **kwargs)
