"""Run-to-run bitwise reproducibility of the HIP ops and of the pipeline stages (GPU box diagnostic)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_checks as pc  # noqa: E402
from mq_det_amd import ops  # noqa: E402
from mq_det_amd.modeling import pipeline  # noqa: E402
from mq_det_amd.structures import ImageList  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rep(name, fn, n=4):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    flat = [o if torch.is_tensor(o) else torch.cat([t.float().reshape(-1) for t in o]) for o in outs]
    same = all(torch.equal(flat[0], f) for f in flat[1:])
    md = max(float((flat[0].float() - f.float()).abs().max()) for f in flat[1:])
    print(f"[{'SAME' if same else 'DIFF'}] {name:<60s} max|d|={md:.3e}", flush=True)
    return same


def attn_case(B, H, D, Nq, Nk, nsplit, shared=False):
    q = torch.randn(B, Nq, H * D, generator=g).half().to(dev)
    k = torch.randn(B, Nk, H * D, generator=g).half().to(dev)
    vt = torch.randn(B, H * D, (Nk + 7) // 8 * 8, generator=g).half().to(dev)
    return lambda: ops.attention(q, k, vt, H, D, scale=1.0 / 16, clamp=50000.0, nsplit=nsplit, nk=Nk)


rep("attn D=256 Nq=22400 Nk=256", attn_case(2, 8, 256, 22400, 256, 1))
rep("attn D=256 Nq=256 Nk=22400 nsplit=6", attn_case(2, 8, 256, 256, 22400, 6))
rep("attn D=256 Nq=256 Nk=22400 nsplit=1", attn_case(1, 8, 256, 256, 22400, 1))
rep("attn D=64 bert", attn_case(8, 12, 64, 256, 256, 1))
rep("attn D=32 preselect nsplit=6", attn_case(8, 8, 32, 200, 5577, 6))

spec, sd, cfg, model, P = pc.tiny(dev)
qkv = torch.randn(2, 100, 168, 3 * 96, generator=g).half().to(dev)
b = "backbone.body.layers.0.blocks.1.attn"
rep("window attn shift", lambda: ops.window_attention(qkv, P[b + ".qkv.bias"], P[b + ".rel_bias"], 3, 7, 3))
x = torch.randn(2, 50, 84, 256, generator=g).half().to(dev)
om = (torch.randn(2, 27, 50, 84, generator=g) * 1.5).to(dev)
rep("dcn im2col", lambda: ops.dcn_im2col(x, om, 1)[0])
w = (torch.randn(256, 2304, generator=g) / 48).half().to(dev)
cols = ops.dcn_im2col(x, om, 1)[0]
rep("F.linear dcn gemm", lambda: F.linear(cols, w))
feats = [torch.randn(2, 256, h, ww, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
         for h, ww in [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]]
rep("dyconv layer", lambda: pipeline.dyconv(P, cfg, "rpn.head.dyhead_tower.2", feats))
l = torch.randn(2, 256, 768, generator=g).half().to(dev)
kb = torch.zeros(2, 256, device=dev)
kb[:, 40:] = -1e30
rep("vl_fuse layer (image side)", lambda: pipeline.vl_fuse(P, "rpn.head.dyhead_tower.0.b_attn", feats, l, kb)[0])
rep("vl_fuse layer (text side)", lambda: pipeline.vl_fuse(P, "rpn.head.dyhead_tower.0.b_attn", feats, l, kb)[1])
rep("bert layer", lambda: pipeline.bert_layer(P, "rpn.head.dyhead_tower.1", l, kb, True))
img = torch.randn(2, 3, 320, 448, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
rep("swin", lambda: pipeline.swin_forward(P, cfg, img))
c = pipeline.swin_forward(P, cfg, img)
rep("fpn (HIP implicit-GEMM convs)", lambda: pipeline.fpn_forward(P, c))

images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
model.load_query_bank(bank)
il = ImageList(images.to(dev), sizes)
kw = dict(captions=None, positive_map=pm, input_ids=ids.to(dev), attention_mask=am.to(dev), return_raw=True)
raws = [model(il, **kw) for _ in range(3)]
for key, f in (("fpn", lambda r: r["feats"]), ("lang hidden", lambda r: [r["lang"]["hidden"]]),
               ("head feats", lambda r: r["head"]["feats"]), ("head hidden", lambda r: [r["head"]["hidden"]]),
               ("dot", lambda r: r["head"]["dot"]), ("bbox", lambda r: r["head"]["bbox_reg"]),
               ("post scores", lambda r: [r["post"]["scores"]]), ("post boxes", lambda r: [r["post"]["boxes"]]),
               ("pre_nms scores", lambda r: [r["post"]["pre_nms"]["scores"]]), ("keep", lambda r: [r["post"]["pre_nms"]["keep"].float()]),
               ("counts", lambda r: [r["post"]["counts"].float()])):
    vals = [torch.cat([t.float().reshape(-1) for t in f(r)]) for r in raws]
    same = all(torch.equal(vals[0], v) for v in vals[1:])
    md = max(float((vals[0] - v).abs().max()) for v in vals[1:])
    print(f"[{'SAME' if same else 'DIFF'}] full model stage: {key:<40s} max|d|={md:.3e}", flush=True)
