"""RCCL worker (one process per GPU, backend "nccl" = RCCL on ROCm) for the two exchanges of the path, at ANY world size incl. 1:
  * parallel.gather_detections / OverlappedGather  -- the fixed-shape [B, K, 6] all-gather of the data path;
  * evaluation.FixedAPAccumulator.synchronize_between_processes -- the evaluator gather (reference
    data/datasets/evaluation/lvis/lvis_eval.py:766-808 + utils/mdetr_dist.py:32-89), state on the DEVICE.
Every rank rebuilds all ranks' inputs from their seeds and compares what RCCL delivered with that expectation; the accumulator itself is pinned
to the reference's LvisEvaluatorFixedAP by the gloo world-2 CPU test.  `force=True`: the collectives are issued even in a one-rank group."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mq_det_amd import parallel                       # noqa: E402
from mq_det_amd.evaluation import FixedAPAccumulator  # noqa: E402


def rank_packed(r, dev):
    g = torch.Generator().manual_seed(500 + r)
    B, K = 3, 316
    boxes = torch.rand(B, K, 4, generator=g) * 800 + 1000 * r
    scores = torch.rand(B, K, generator=g) + 0.01
    scores[:, K - 7 * (r + 1):] = -1.0                 # ragged: different number of valid detections per rank
    labels = torch.randint(1, 1204, (B, K), generator=g)
    return parallel.pack_detections(boxes, scores, labels).to(dev)


def rank_detections(r):
    g = torch.Generator().manual_seed(100 + r)
    out = []
    for it in range(6):                                # six images per rank, 300 detections each, 40 categories, score ties
        n = 300
        out.append((torch.full((n,), float(1000 * r + it)), torch.randint(1, 41, (n,), generator=g),
                    torch.randint(0, 50, (n,), generator=g).float() / 50.0, torch.rand(n, 4, generator=g) * 100))
    return out


def run(dev, world, rank):
    mine = rank_packed(rank, dev)
    allp = parallel.gather_detections(mine, force=True)
    want = torch.cat([rank_packed(r, dev) for r in range(world)])
    assert allp.device.type == dev.type and allp.shape == want.shape and torch.equal(allp, want)
    og = parallel.OverlappedGather()
    steps = [mine + 10 * s for s in range(3)]
    got = [og.submit(p_) for p_ in steps] + [og.flush()]
    assert got[0] is None
    for s, g_ in enumerate(got[1:]):
        if dev.type == "cuda":
            torch.cuda.synchronize()
        assert torch.equal(g_, want + 10 * s if world > 1 else steps[s])
    # evaluator: per-rank top-k on the device, exchanged over RCCL, against the CPU accumulator fed with the same rows rank by rank
    acc = FixedAPAccumulator(topk=50, device=dev, prune_at=700)
    for img, lab, sc, bx in rank_detections(rank):
        acc.update(img.to(dev), lab.to(dev), sc.to(dev), bx.to(dev))
    acc.synchronize_between_processes(force=True)
    assert acc.rows.device.type == dev.type
    parts = []
    for r in range(world):
        a = FixedAPAccumulator(topk=50, device="cpu", prune_at=700)
        for img, lab, sc, bx in rank_detections(r):
            a.update(img, lab, sc, bx)
        a._fold()
        parts.append(a.rows)
    want_rows = torch.cat(parts)
    assert torch.equal(acc.rows.cpu(), want_rows), (acc.rows.shape, want_rows.shape)
    cats = acc.by_cat()
    assert len(cats) == 40 and all(len(v) <= 50 * world for v in cats.values())
    return len(want_rows)


if __name__ == "__main__":
    backend = os.environ.get("MQ_WORKER_BACKEND", "nccl")              # "gloo": the same worker on CPU tensors (a dry run of this script without a GPU)
    rank, local, world = parallel.init_distributed(backend)
    if world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=0, world_size=1)
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    n = run(dev, world, rank)
    dist.barrier()
    print("RCCL_GATHER_OK", rank, world, n, flush=True)
    dist.destroy_process_group()
