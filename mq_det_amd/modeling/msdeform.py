"""Multi-scale deformable attention of the MQ-GroundingDINO path on the HIP kernel (mq_msdeform_attn_fwd).

Reference: groundingdino_new/models/GroundingDINO/ms_deform_attn.py:232-359 (MultiScaleDeformableAttention.forward; used by
the deformable encoder layers transformer.py:482-596 and the decoder's cross-attention :868-927) and the `_C.ms_deform_attn_forward`
operator behind it.  MI355X-first differences: the value projection's fp16 output is gathered directly (the reference casts
value, locations and weights to fp32 first, :330-336), the sampling-offset and attention-weight projections are ONE GEMM, the
gather kernel reads one contiguous 64-byte row per (head, corner) and accumulates in fp32.
Operator-level form (materialised sampling locations, one module at a time): what the INTEGRATION.md stub for
`groundingdino_new._C.ms_deform_attn_forward` binds.  The MQ-GroundingDINO model itself (modeling/gdino_pipeline.py) uses the
fused-query kernel mq_msdeform_attn_q_fwd instead."""
import torch
import torch.nn.functional as F

from .. import ops


def pack_msda(sd, p, device, dtype=torch.float16):
    """Inference weights of one MultiScaleDeformableAttention module: fp16 casts, offsets | weights as one projection."""
    def h(n):
        return sd[p + n].detach().to(device=device, dtype=dtype).contiguous()
    return {"value.w": h(".value_proj.weight"), "value.b": h(".value_proj.bias"),
            "qproj.w": torch.cat([h(".sampling_offsets.weight"), h(".attention_weights.weight")], 0).contiguous(),
            "qproj.b": torch.cat([h(".sampling_offsets.bias"), h(".attention_weights.bias")], 0).contiguous(),
            "out.w": h(".output_proj.weight"), "out.b": h(".output_proj.bias")}


def ms_deform_attn(W, query, value, reference_points, spatial_shapes, key_padding_mask=None, query_pos=None, heads=8, levels=4,
                   points=4):
    """query [B, Q, C] fp16, value [B, S, C] fp16 or None (-> query), reference_points [B, Q, L, 2 | 4] fp32 normalised,
    spatial_shapes: list of (H, W), key_padding_mask [B, S] bool or None -> [B, Q, C] fp16."""
    if value is None:
        value = query
    if query_pos is not None:
        query = query + query_pos
    B, Q, C = query.shape
    S = value.shape[1]
    v = F.linear(value, W["value.w"], W["value.b"])
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    n = heads * levels * points
    qp = F.linear(query, W["qproj.w"], W["qproj.b"]).float()
    off = qp[..., :2 * n].reshape(B, Q, heads, levels, points, 2)
    aw = qp[..., 2 * n:].reshape(B, Q, heads, levels * points).softmax(-1).reshape(B, Q, heads, levels, points)
    shp = torch.tensor([[h, w] for h, w in spatial_shapes], dtype=torch.float32, device=query.device)
    rp = reference_points.float()
    if rp.shape[-1] == 2:
        norm = torch.stack([shp[:, 1], shp[:, 0]], -1)
        loc = rp[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    elif rp.shape[-1] == 4:
        loc = rp[:, :, None, :, None, :2] + off / points * rp[:, :, None, :, None, 2:] * 0.5
    else:
        raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {rp.shape[-1]} instead.")
    out = ops.ms_deform_attn(v.view(B, S, heads, C // heads), spatial_shapes, loc.contiguous(), aw.contiguous())
    return F.linear(out, W["out.w"], W["out.b"])
