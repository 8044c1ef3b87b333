"""Evaluator-side accumulation and gather of detections as TENSORS.

Reference: `LvisEvaluatorFixedAP` (data/datasets/evaluation/lvis/lvis_eval.py:766-808) keeps, per category, the `topk`
(10 000) best detections seen so far as Python lists of dicts (`update`: per-category sort + `_merge_lists`, :752-763) and
exchanges them between ranks by pickling the whole dict through two all_gathers of uint8 tensors
(`synchronize_between_processes` :801-808 -> utils/mdetr_dist.py:32-89) -- the only large message of the eval pipeline
(10^2 - 10^3 MB of pickle).  Here the same state is one [n, 7] fp32 tensor per rank
(image id, category id, score, x, y, w, h) that stays on the device:
  * `update`      appends rows; when the buffer outgrows `prune_at` rows it is cut back to the per-category top-k with two
                  stable sorts (score descending, then category) and a rank-within-category mask -- ties keep the earlier
                  detection, like `_merge_lists`;
  * `synchronize_between_processes`  one all_gather of the row counts and one fixed-shape `all_gather_into_tensor` of the
                  padded rows over RCCL (28 bytes per detection instead of a pickled dict) -- like the reference the
                  per-rank lists are CONCATENATED, not re-trimmed (lvis_eval.py:803-806);
  * `by_cat`      the reference's `{category: [ {"image_id", "category_id", "bbox", "score"} ]}` view for the LVIS API.
Image ids must be exactly representable in fp32 (< 2^24; LVIS / COCO ids are < 600 000)."""
from collections import defaultdict

import torch
import torch.distributed as dist


class FixedAPAccumulator:
    def __init__(self, topk=10000, device="cpu", prune_at=None):
        self.topk = int(topk)
        self.device = torch.device(device)
        self.prune_at = prune_at
        self.rows = torch.zeros(0, 7, dtype=torch.float32, device=self.device)
        self._pending = []
        self._npending = 0

    # ------------------------------------------------------------------ accumulate
    def update(self, image_ids, labels, scores, boxes_xywh):
        """Detections of one or more images: image_ids [n] / labels [n] / scores [n] / boxes [n, 4] (x, y, w, h)."""
        n = len(scores)
        if n == 0:
            return
        r = torch.empty(n, 7, dtype=torch.float32, device=self.device)
        r[:, 0] = torch.as_tensor(image_ids, device=self.device).float()
        r[:, 1] = torch.as_tensor(labels, device=self.device).float()
        r[:, 2] = torch.as_tensor(scores, device=self.device).float()
        r[:, 3:] = torch.as_tensor(boxes_xywh, device=self.device).float()
        self._pending.append(r)
        self._npending += n
        limit = self.prune_at if self.prune_at is not None else 4 * max(len(self.rows), 1 << 16)
        if self._npending >= limit:
            self._fold()

    def update_boxlist(self, image_id, boxlist):
        """One reference-style prediction (engine/inference.py:643-648: `output.bbox` in xyxy after `resize_box` to the original
        image size) -> rows with the bbox exactly as `LvisEvaluatorFixedAP.prepare` builds it (lvis_eval.py:810-835 through
        `convert_to_xywh` :998-1000): (xmin, ymin, xmax - xmin, ymax - ymin) -- NO legacy +1 (that is BoxList.convert("xywh"),
        which the LVIS path never calls)."""
        if boxlist.mode != "xyxy":
            boxlist = boxlist.convert("xyxy")
        b = boxlist.bbox
        n = len(b)
        xywh = torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), 1) if n else b.reshape(0, 4)
        self.update(torch.full((n,), float(image_id)), boxlist.get_field("labels"), boxlist.get_field("scores"), xywh)

    def _fold(self):
        if self._pending:
            self.rows = self._prune(torch.cat([self.rows] + self._pending))
            self._pending, self._npending = [], 0

    def _prune(self, rows):
        """Per-category top-k, each category's rows in descending score order, ties in arrival order."""
        if len(rows) == 0:
            return rows
        o = torch.sort(rows[:, 2], descending=True, stable=True)[1]
        rows = rows[o]
        o = torch.sort(rows[:, 1], stable=True)[1]
        rows = rows[o]
        cat = rows[:, 1]
        first = torch.ones(len(rows), dtype=torch.bool, device=rows.device)
        first[1:] = cat[1:] != cat[:-1]
        start = torch.cummax(torch.where(first, torch.arange(len(rows), device=rows.device), torch.zeros((), dtype=torch.long, device=rows.device)), 0)[0]
        rank = torch.arange(len(rows), device=rows.device) - start
        return rows[rank < self.topk]

    # ------------------------------------------------------------------ exchange
    def synchronize_between_processes(self, group=None, force=False):
        """`force`: run the two collectives even in a one-rank group (the one-GPU RCCL test; a one-rank job otherwise has nothing to exchange)."""
        self._fold()
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
            return
        world = dist.get_world_size(group)
        n = torch.tensor([len(self.rows)], dtype=torch.long, device=self.device)
        sizes = torch.empty(world, dtype=torch.long, device=self.device)
        dist.all_gather_into_tensor(sizes, n, group=group)
        sizes = sizes.tolist()
        m = max(max(sizes), 1)
        pad = torch.zeros(m, 7, dtype=torch.float32, device=self.device)
        pad[:len(self.rows)] = self.rows
        out = torch.empty(world * m, 7, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        self.rows = torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])     # concatenated, like the reference

    # ------------------------------------------------------------------ reference view
    def by_cat(self):
        self._fold()
        rows = self.rows.cpu()
        out = defaultdict(list)
        for img, cat, sc, x, y, w, h in rows.tolist():
            out[int(cat)].append({"image_id": int(img), "category_id": int(cat), "bbox": [x, y, w, h], "score": sc})
        return out
