"""Data-parallel inference helpers: one process per GPU, RCCL over xGMI via torch.distributed ("nccl" on ROCm).

The MQ-Det forward shards by image (contiguous blocks per rank like the reference sampler,
data/samplers/distributed.py:57-64); the only exchange is the gather of detections.  The reference pickles
variable-length BoxLists and runs two all_gathers of uint8 tensors (utils/comm.py:61-101); here each rank
contributes one fixed-shape [B_local, K, 6] fp32 block (x1, y1, x2, y2, score, label; score <= 0 = empty
slot) in a single all_gather_into_tensor -- 7.2 KB / image, latency-bound on any fabric.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_items, rank, world):
    """Contiguous block of ceil(n/world) items per rank, wrapping around at the end (reference sampler)."""
    per = -(-n_items // world)
    return [(rank * per + i) % n_items for i in range(per)]


def pack_detections(boxes, scores, labels):
    """[B,K,4], [B,K], [B,K] -> [B,K,6] fp32."""
    return torch.cat([boxes.float(), scores.float()[..., None], labels.float()[..., None]], -1).contiguous()


def gather_detections(packed, force=False):
    """All ranks receive [world * B_local, K, 6] (rank-major order).  Single fixed-shape collective.  `force`: issue the collective even in a
    one-rank group (the one-GPU RCCL test of tests/test_gpu_parity.py; a one-rank job otherwise skips it)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return packed
    world = dist.get_world_size()
    out = packed.new_empty((world * packed.shape[0],) + tuple(packed.shape[1:]))
    dist.all_gather_into_tensor(out, packed.contiguous())
    return out


class OverlappedGather:
    """The same fixed-shape gather, one step behind: `submit(packed_k)` starts the collective of step k asynchronously (RCCL runs it on its
    own stream; xGMI is idle during a forward) and hands back the gathered block of step k - 1, so the ~10 us collective and its launch
    latency sit under the next forward instead of behind every step.  `flush()` returns the last block.  One rank: a pass-through.
    The evaluator consumes detections image by image (engine/inference.py:643-648) -- a one-step delay changes nothing it can observe."""

    def __init__(self):
        self._pending = None                      # (work handle or None, out tensor, the packed tensor kept alive)

    def _finish(self):
        if self._pending is None:
            return None
        work, out, _keep = self._pending
        self._pending = None
        if work is not None:
            work.wait()                           # stream-level wait on GPU backends, a host wait on gloo
        return out

    def submit(self, packed):
        prev = self._finish()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            self._pending = (None, packed, packed)
            return prev
        world = dist.get_world_size()
        src = packed.contiguous()
        out = src.new_empty((world * src.shape[0],) + tuple(src.shape[1:]))
        work = dist.all_gather_into_tensor(out, src, async_op=True)
        self._pending = (work, out, src)
        return prev

    def flush(self):
        return self._finish()


def unpack_detections(packed):
    """[N,K,6] -> list of dicts with the non-empty rows."""
    res = []
    for p in packed:
        keep = p[:, 4] > 0
        res.append({"boxes": p[keep, :4], "scores": p[keep, 4], "labels": p[keep, 5].long()})
    return res
