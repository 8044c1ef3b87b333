"""Oracle: multi-scale deformable attention of the MQ-GroundingDINO path (test infrastructure, see oracle/__init__.py).

Restates groundingdino_new/models/GroundingDINO/ms_deform_attn.py:93-133 (`multi_scale_deformable_attn_pytorch`, the
pure-torch statement of the CUDA operator: per level F.grid_sample(bilinear, zeros, align_corners=False) on locations
2*loc - 1, weighted sum over levels x points) and :232-359 (`MultiScaleDeformableAttention.forward`, batch_first: value
projection, key-padding zeroing, sampling offsets and softmax-ed attention weights from the query, sampling locations from
2-d reference points or 4-d reference boxes, output projection).
PINS: tests/golden/msda.npz holds outputs of the reference's own `multi_scale_deformable_attn_pytorch` (executed in place by
oracle/gen_golden.py); on the GPU both this restatement and the HIP kernel are also compared with the reference's CUDA kernel
(ms_deform_im2col_cuda.cuh:237-299 built by oracle/build_ref.py)."""
import torch
import torch.nn.functional as F


def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B, S, M, D]; spatial_shapes [(H, W)] * L; loc [B, Q, M, L, P, 2]; attn [B, Q, M, L, P] -> [B, Q, M * D]."""
    B, _, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes]
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(B * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                          # [B*M, Q, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = attention_weights.transpose(1, 2).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, M * D, Q)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn(sd, p, query, value, reference_points, spatial_shapes, key_padding_mask=None, query_pos=None,
                   heads=8, levels=4, points=4):
    """MultiScaleDeformableAttention.forward (batch_first).  query [B, Q, C], value [B, S, C] (None -> query),
    reference_points [B, Q, L, 2 or 4] normalised, key_padding_mask [B, S] bool (True = padding)."""
    if value is None:
        value = query
    if query_pos is not None:
        query = query + query_pos
    B, Q, C = query.shape
    S = value.shape[1]
    v = F.linear(value, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"])
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    v = v.view(B, S, heads, -1)
    off = F.linear(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"]).view(B, Q, heads, levels, points, 2)
    aw = F.linear(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"]).view(B, Q, heads, levels * points)
    aw = aw.softmax(-1).view(B, Q, heads, levels, points)
    shp = torch.tensor([[h, w] for h, w in spatial_shapes], dtype=query.dtype)
    if reference_points.shape[-1] == 2:
        norm = torch.stack([shp[:, 1], shp[:, 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / points * reference_points[:, :, None, :, None, 2:] * 0.5
    out = ms_deform_attn_core(v, spatial_shapes, loc, aw)
    return F.linear(out, sd[p + ".output_proj.weight"], sd[p + ".output_proj.bias"])


def make_msda_weights(C=256, heads=8, levels=4, points=4, seed=0, prefix="attn"):
    g = torch.Generator().manual_seed(seed)
    n = heads * levels * points
    return {prefix + ".value_proj.weight": torch.randn(C, C, generator=g) / C ** 0.5, prefix + ".value_proj.bias": torch.randn(C, generator=g) * 0.05,
            prefix + ".sampling_offsets.weight": torch.randn(2 * n, C, generator=g) * 0.05, prefix + ".sampling_offsets.bias": torch.randn(2 * n, generator=g) * 2.0,
            prefix + ".attention_weights.weight": torch.randn(n, C, generator=g) * 0.1, prefix + ".attention_weights.bias": torch.randn(n, generator=g) * 0.1,
            prefix + ".output_proj.weight": torch.randn(C, C, generator=g) / C ** 0.5, prefix + ".output_proj.bias": torch.randn(C, generator=g) * 0.05}
