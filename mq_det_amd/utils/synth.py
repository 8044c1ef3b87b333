"""Seeded synthetic weights / inputs for benchmarking (no checkpoints or datasets exist offline).

`randomize_(model)` re-draws every parameter so that all branches of the network are numerically live
(the reference zero-initialises the GCP gates, the layer-scale and the DCN offset conv): fan-in scaled normals
for weights, ~N(1, 0.1) norm scales, small biases, O(1 px) deformable offsets, alignment bias chosen so that a
few percent of the (location, class) scores clear the 0.05 threshold and the post-processing sees real work.
"""
import math

import torch


@torch.no_grad()
def randomize_(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        shape = tuple(p.shape)
        if name.endswith("ff_gate"):
            p.fill_(0.3)
        elif name.endswith("log_scale"):
            p.fill_(0.0)
        elif name.endswith("bias0"):
            p.fill_(-6.0)
        elif name.endswith(".scale"):
            p.fill_(1.0)
        elif "gamma_" in name:
            p.copy_(torch.randn(shape, generator=g) * 0.05 + 0.5)
        elif name.endswith("relative_position_bias_table"):
            p.copy_(torch.randn(shape, generator=g) * 0.5)
        elif "embeddings" in name and name.endswith("weight") and p.dim() == 2:
            p.copy_(torch.randn(shape, generator=g) * 0.3)
        elif p.dim() == 1:
            if name.endswith("weight") and ("norm" in name.lower() or ".bn." in name):
                p.copy_(torch.randn(shape, generator=g) * 0.1 + 1.0)
            else:
                p.copy_(torch.randn(shape, generator=g) * 0.05)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 40.0 if "dot_product_projection_text" in name else (1.5 if ("qkv" in name or "query" in name or "key" in name or "to_q" in name or "to_kv" in name or "v_proj" in name or "l_proj" in name) else 1.0)
            p.copy_(torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in)))
    model._invalidate()
    return model


def synthetic_bank(labels, channels=256, k=5, seed=1):
    g = torch.Generator().manual_seed(seed)
    return {int(l): torch.randn(k, 1, channels, generator=g) for l in labels}
