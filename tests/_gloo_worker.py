import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mq_det_amd import parallel  # noqa: E402

rank, local, world = parallel.init_distributed("gloo")
assert world == 2
B, K = 3, 5
boxes = torch.arange(B * K * 4, dtype=torch.float32).reshape(B, K, 4) + 1000 * rank
scores = torch.rand(B, K) + 0.01
scores[:, K - rank - 1:] = -1.0                 # ragged: different number of valid detections per rank
labels = torch.full((B, K), rank + 1)
packed = parallel.pack_detections(boxes, scores, labels)
allp = parallel.gather_detections(packed)
assert allp.shape == (world * B, K, 6)
assert torch.equal(allp[rank * B:(rank + 1) * B], packed)
other = 1 - rank
assert float(allp[other * B, 0, 0]) == 1000 * other and float(allp[other * B, 0, 5]) == other + 1
dets = parallel.unpack_detections(allp)
assert len(dets) == world * B and len(dets[0]["boxes"]) == K - 1 and len(dets[B]["boxes"]) == K - 2
assert parallel.shard_range(10, rank, world) == [rank * 5 + i for i in range(5)]
assert parallel.shard_range(5, 1, 2) == [3, 4, 0]
# the same collective one step behind the "forward" (parallel.OverlappedGather, bench.py --overlap-gather): submit(step k) returns the gathered
# block of step k - 1, flush() the last one; every block equals the synchronous gather of its step
og = parallel.OverlappedGather()
steps = [packed + 10 * s for s in range(4)]
got = [og.submit(p_) for p_ in steps] + [og.flush()]
assert got[0] is None and og.flush() is None
for s, g_ in enumerate(got[1:]):
    assert torch.equal(g_, parallel.gather_detections(steps[s]))
dist.barrier()
print("GATHER_OK", rank, flush=True)
dist.destroy_process_group()
