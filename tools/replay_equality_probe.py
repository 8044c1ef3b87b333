#!/usr/bin/env python3
"""GPU-box diagnostic: is the HIP-graph replay of a forward BIT-EQUAL to the eager forward, and to a second replay?
Prints one line per comparison for the tiny model and (with --bench) the benchmark configuration at B = 2.  Run it under the environment
settings to compare (TORCH_BLAS_PREFER_HIPBLASLT=0, HIPBLASLT_WORKSPACE_SIZE, ...): the library GEMMs are the only kernels of the step whose
selection can differ between an eager call and a call under stream capture."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_checks as pc  # noqa: E402
from mq_det_amd.structures import ImageList  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("MQ_PROBE_BLAS"):
    torch.backends.cuda.preferred_blas_library(os.environ["MQ_PROBE_BLAS"])
spec, sd, cfg, model, P = pc.tiny(dev)
images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
model.load_query_bank(bank)
il = ImageList(images.to(dev), sizes)
kw = dict(captions=None, positive_map=pm, input_ids=ids.to(dev), attention_mask=am.to(dev))


def run(graph):
    model.use_hip_graph = graph
    model(il, **kw)
    return model.last_packed.clone()


def cmp(name, a, b):
    same = torch.equal(a, b)
    d = float((a - b).abs().max())
    print(f"[{'SAME' if same else 'DIFF'}] {name:<48s} max|d| = {d:.3e}", flush=True)


model.use_hip_graph = False
model.clear_caches()
e1, e2 = run(False), run(False)
cmp("eager vs eager", e1, e2)
model.clear_caches()
outs = [run(True) for _ in range(5)]            # eager (warm), capture, replay, replay, replay
captured = any(e.get("stage") == 2 for e in model._graphs.values())
print("captured:", captured, "blas:", torch.backends.cuda.preferred_blas_library(), flush=True)
cmp("graph call 1 (eager warm-up) vs eager", outs[0], e1)
cmp("graph call 2 (capture + first replay) vs eager", outs[1], e1)
cmp("replay vs eager", outs[3], e1)
cmp("replay vs replay", outs[3], outs[4])
