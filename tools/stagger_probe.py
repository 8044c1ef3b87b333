"""GPU probe: does the staggered program capture into a HIP graph?  Runs the tiny model; prints OK / the exception.  (round 4, call 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity_checks as pc
from mq_det_amd.structures import ImageList
dev = torch.device("cuda:0")
spec, sd, cfg, model, P = pc.tiny(dev)
if os.environ.get("PROBE_LEVEL_STREAMS") == "0":
    cfg.MODEL.DYHEAD.LEVEL_STREAMS = False
images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
model.load_query_bank(bank)
imgs = torch.cat([images, torch.flip(images, dims=[3])])
il = ImageList(imgs.to(dev), list(sizes) * 2)
kw = dict(captions=None, positive_map=pm, input_ids=ids.repeat(2, 1).to(dev), attention_mask=am.repeat(2, 1).to(dev))
model.backbone_cache, model.use_hip_graph, model.micro_batches = False, True, int(os.environ.get("MQ_MICRO_BATCHES", "2"))
model.clear_caches()
outs = [model(il, **kw) for _ in range(3)]
torch.cuda.synchronize()
print("PROBE_OK", [len(o) for o in outs[2]], {k[0]: e.get("stage") for k, e in model._graphs.items()}, flush=True)
