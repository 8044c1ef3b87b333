"""ctypes access to oracle/_ref/libmqdet_ref.so -- the reference's own device kernels built by oracle/build_ref.py
(test infrastructure only; GPU tensors in, GPU tensors out).  Host-side arithmetic around each kernel restates the
reference's C++ wrapper and cites it.  Never imported by mq_det_amd."""
import ctypes
import os

import torch

from .build_ref import OUT

_LIB = None
_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def available():
    return os.path.exists(OUT)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError(f"{OUT} not found: run `python -m oracle.build_ref` where /root/reference is mounted "
                               "(__graft_entry__.build() does); the file travels to the GPU box with the snapshot")
        L = ctypes.CDLL(OUT)
        L.ref_dcn_im2col.argtypes = [_vp] * 4 + [_i] * 8 + [_vp]
        L.ref_ml_nms.argtypes = [_vp, _i, _f, _vp]
        L.ref_roi_align.argtypes = [_vp] * 3 + [_i] * 6 + [_f, _i, _vp]
        L.ref_ms_deform_im2col.argtypes = [_vp] * 6 + [_i] * 7 + [_vp]
        for fn in (L.ref_dcn_im2col, L.ref_ml_nms, L.ref_roi_align, L.ref_ms_deform_im2col, L.ref_abi_version):
            fn.restype = _i
        _LIB = L
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dcn_v2(x, offset_buf, mask_buf, weight, bias, stride):
    """modulated_deform_conv_cuda_forward (csrc/cuda/deform_conv_cuda.cu:496-575) for kernel 3, pad 1, dilation 1,
    groups 1, deformable_groups 1: per sample, the reference's im2col kernel fills columns [C*9, Ho*Wo], then
    output[b] = weight.flatten(1) @ columns (+ bias).  The per-sample offset / mask buffers are handed to the kernel as
    raw pointers exactly like `offset[b]`, `mask[b]` (any spatial dims: flat indexing by the OUTPUT dims, quirk 1)."""
    L = lib()
    B, C, H, W = x.shape
    O = weight.shape[0]
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x, offset_buf, mask_buf = x.float().contiguous(), offset_buf.float().contiguous(), mask_buf.float().contiguous()
    assert offset_buf[0].numel() >= 18 * Ho * Wo and mask_buf[0].numel() >= 9 * Ho * Wo
    out = torch.empty(B, O, Ho, Wo, dtype=torch.float32, device=x.device)
    w2 = weight.float().reshape(O, C * 9)
    for b in range(B):
        col = torch.zeros(C * 9, Ho * Wo, dtype=torch.float32, device=x.device)
        rc = L.ref_dcn_im2col(_p(x[b]), _p(offset_buf[b]), _p(mask_buf[b]), _p(col), 1, C, H, W, Ho, Wo, 1, stride, _stream())
        assert rc == 0, rc
        out[b] = (w2 @ col).reshape(O, Ho, Wo)
    if bias is not None:
        out += bias.float().reshape(1, O, 1, 1)
    return out


def ml_nms(boxes, scores, labels, thresh):
    """ml_nms_cuda (csrc/cuda/ml_nms.cu:78-149): sort by score (descending), bitmask kernel, host sweep, indices of the
    kept boxes in ascending order of the ORIGINAL positions' sort rank -> returns original indices, ascending."""
    L = lib()
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.sort(scores, 0, descending=True)[1]
    b6 = torch.cat([boxes.float(), scores.float()[:, None], labels.float()[:, None]], 1)[order].contiguous()
    keep = (ctypes.c_ubyte * n)()
    torch.cuda.synchronize()
    k = L.ref_ml_nms(_p(b6), n, float(thresh), ctypes.cast(keep, ctypes.c_void_p))
    assert k >= 0
    km = torch.tensor(list(keep), dtype=torch.bool)
    return torch.sort(order.cpu()[km])[0]


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio):
    """ROIAlign_forward_cuda (csrc/cuda/ROIAlign_cuda.cu:262-306), the legacy (aligned=False) operator.
    feat [N, C, H, W] fp32, rois [R, 5] = (batch index, x1, y1, x2, y2) -> [R, C, PH, PW]."""
    L = lib()
    feat, rois = feat.float().contiguous(), rois.float().contiguous()
    N, C, H, W = feat.shape
    R = rois.shape[0]
    PH, PW = (out_size, out_size) if isinstance(out_size, int) else out_size
    out = torch.empty(R, C, PH, PW, dtype=torch.float32, device=feat.device)
    rc = L.ref_roi_align(_p(feat), _p(rois), _p(out), R, C, H, W, PH, PW, float(spatial_scale), int(sampling_ratio), _stream())
    assert rc == 0, rc
    return out


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """ms_deform_attn_cuda_forward (csrc_groundingdino/MsDeformAttn/ms_deform_attn_cuda.cu:21-81) with
    im2col_step = batch: value [B, S, heads, ch], shapes [L, 2] int64, level_start [L] int64,
    loc [B, Q, heads, L, P, 2], attn [B, Q, heads, L, P] -> [B, Q, heads * ch]."""
    Lb = lib()
    value, loc, attn = value.float().contiguous(), sampling_locations.float().contiguous(), attention_weights.float().contiguous()
    B, S, M, D = value.shape
    _, Q, _, Lv, P, _ = loc.shape
    shapes = spatial_shapes.to(device=value.device, dtype=torch.int64).contiguous()
    start = level_start_index.to(device=value.device, dtype=torch.int64).contiguous()
    out = torch.empty(B, Q, M * D, dtype=torch.float32, device=value.device)
    rc = Lb.ref_ms_deform_im2col(_p(value), _p(shapes), _p(start), _p(loc), _p(attn), _p(out), B, S, M, D, Lv, Q, P, _stream())
    assert rc == 0, rc
    return out
