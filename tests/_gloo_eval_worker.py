"""world-2 gloo worker: tensor evaluator gather (mq_det_amd.evaluation) vs the reference's own LvisEvaluatorFixedAP +
utils/mdetr_dist.all_gather executed in place (pickle all_gather on the CPU group, MDETR_CPU_REDUCE=1)."""
import os
import sys
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mq_det_amd import parallel                      # noqa: E402
from mq_det_amd.evaluation import FixedAPAccumulator  # noqa: E402
from oracle import _refload                          # noqa: E402

rank, local, world = parallel.init_distributed("gloo")
assert world == 2
os.environ["MDETR_CPU_REDUCE"] = "1"
md = _refload.reference_functions("maskrcnn_benchmark/utils/mdetr_dist.py", ["_get_global_gloo_group", "all_gather", "get_world_size", "is_dist_avail_and_initialized"],
                                  {"dist": dist, "io": __import__("io"), "functools": __import__("functools")})
ev = _refload.reference_classes("maskrcnn_benchmark/data/datasets/evaluation/lvis/lvis_eval.py", ["LvisEvaluatorFixedAP"], ["_merge_lists"],
                                {"LVIS": object, "dist": types.SimpleNamespace(all_gather=md["all_gather"])})
ref = ev["LvisEvaluatorFixedAP"](gt=None, topk=7)
acc = FixedAPAccumulator(topk=7, prune_at=40)
g = torch.Generator().manual_seed(100 + rank)
for it in range(6):                                   # six "images" per rank, 30 detections each, 5 categories, score ties
    n = 30
    img = 1000 * rank + it
    labels = torch.randint(1, 6, (n,), generator=g)
    scores = (torch.randint(0, 50, (n,), generator=g).float() / 50.0)
    boxes = torch.rand(n, 4, generator=g) * 100
    anns = [{"image_id": img, "category_id": int(labels[k]), "bbox": boxes[k].tolist(), "score": float(scores[k])} for k in range(n)]
    ref.update(anns, is_cur_results=True)
    acc.update(torch.full((n,), float(img)), labels, scores, boxes)
ref.synchronize_between_processes()
acc.synchronize_between_processes()
mine = acc.by_cat()
assert set(mine) == set(ref.by_cat), (set(mine), set(ref.by_cat))
for cat, anns in ref.by_cat.items():
    a = [(x["image_id"], round(x["score"], 6), [round(v, 3) for v in x["bbox"]]) for x in anns]
    b = [(x["image_id"], round(x["score"], 6), [round(v, 3) for v in x["bbox"]]) for x in mine[cat]]
    assert a == b, (cat, a[:3], b[:3])
    assert len(a) <= 14
dist.barrier()
print("EVAL_GATHER_OK", rank, sum(len(v) for v in mine.values()), flush=True)
dist.destroy_process_group()
