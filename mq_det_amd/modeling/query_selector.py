"""Vision-query selection (host-side glue, tiny compute).

Reference: modeling/query_selector/query_selector.py:8-116 (bank `{label: Tensor[n, scales, C]}`, <= k rows per
label, `[V, T]` 0/1 attention mask, `pad_sequence`) and generalized_vl_rcnn_new.py:295-305 (label -> token map).
MI355X-first difference: besides the dense mask the selector emits the gather index `idx[b, t, :]` (the
vision rows each text token may attend to) that the reference re-derives from the mask with a top-k trick in
EACH of the 6 GCP layers (modeling_bert_new.py:40-63,162-184); it is built once, on the host, from the
positive_map the caller already holds -- no device work, no sync.
"""
import os

import numpy as np
import torch
from torch import nn


def labels_and_maps(positive_map, T, device="cpu", dtype=torch.float32):
    labels = [k for k, v in positive_map.items() if len(v) != 0]
    m = torch.zeros(len(labels), T, dtype=dtype, device=device)
    for j, lab in enumerate(labels):
        m[j, list(positive_map[lab])] = 1
    return labels, m / (m.sum(-1)[:, None] + 1e-6)


def build_token_index(positive_map, labels, device):
    """[L, MT] int32 token positions per label (-1 padded) + [L] int32 label ids, for the scoring kernel."""
    mt = max(1, max((len(positive_map[l]) for l in labels), default=1))
    idx = torch.full((max(len(labels), 1), mt), -1, dtype=torch.int32)
    for j, l in enumerate(labels):
        toks = positive_map[l] if not isinstance(positive_map[l], int) else [positive_map[l]]
        idx[j, :len(toks)] = torch.tensor(toks, dtype=torch.int32)
    ids = torch.tensor(list(labels) if labels else [0], dtype=torch.int32)
    return idx.to(device, non_blocking=True), ids.to(device, non_blocking=True)


class QuerySelector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.num_query_per_class = cfg.VISION_QUERY.NUM_QUERY_PER_CLASS
        # VISION_QUERY.REFERENCE_RNG_STREAM (not a reference key; default False): consume numpy's global generator once per label
        # on EVERY forward exactly like the reference (query_selector.py:74) -- also when every draw is the identity -- so that a
        # run seeded like a reference run keeps the SAME generator state call for call.  Off: identity draws are
        # skipped and memoised (same selections, no per-forward host work), and the generator state then differs from a
        # reference run's as soon as one caption did not need a real draw.
        self.reference_rng = bool(cfg.VISION_QUERY.get("REFERENCE_RNG_STREAM", False))
        self.query_bank = None
        self._dev_bank = {}
        self._sel_cache = {}
        path = cfg.VISION_QUERY.QUERY_BANK_PATH
        if path:
            if not os.path.exists(path):
                raise FileNotFoundError(f"query bank path {path} not exists")
            self.load_query_bank(path)

    def load_query_bank(self, bank):
        """`bank`: path to a torch-saved dict or the dict itself: {label: Tensor[n, scales, C]}."""
        self.query_bank = torch.load(bank, map_location="cpu") if isinstance(bank, (str, os.PathLike)) else bank
        self._dev_bank = {}
        self._sel_cache = {}

    def _candidates(self, label):
        """Bank entry of a label; None when the label has no vision query.  Reference banks are `defaultdict(list)`
        (engine/inference.py:401): a label without queries reads as `[]` and contributes no vision rows
        (query_selector.py:77-78, `isinstance(candidate_queries, list)`); a plain dict without the key is treated alike."""
        bank = self.query_bank
        cand = bank[label] if (label in bank or hasattr(bank, "default_factory")) else None
        if cand is None or isinstance(cand, (list, tuple)) or len(cand) == 0:
            return None
        return cand

    def deterministic(self, labels):
        """True when no label of `labels` holds more bank rows than NUM_QUERY_PER_CLASS: the reference's eval-mode draw
        `sorted(np.random.choice(len, n, replace=False))` (query_selector.py:74-76) is then the identity, and the
        selection may be memoised.  Otherwise every forward draws again, from numpy's global generator like the reference.
        (With VISION_QUERY.REFERENCE_RNG_STREAM nothing is memoised: every label of every forward consumes its draw.)"""
        if self.reference_rng:
            return False
        for lab in labels:
            cand = self._candidates(lab)
            if cand is not None and len(cand) > self.num_query_per_class:
                return False
        return True

    def _rows(self, label, device, dtype, draw=False):
        """[n * scales, C] vision rows of one label (n = min(len, k)); None when the label has none.
        draw: consume numpy's global generator exactly like the reference does for EVERY label (:74), also when the draw is
        the identity -- used when some label of the caption holds more than k rows, so that the same seed gives the same
        selection as the reference."""
        cand = self._candidates(label)
        k = self.num_query_per_class
        if draw:
            n_tot = 0 if cand is None else len(cand)
            idx = sorted(np.random.choice(n_tot, min(n_tot, k), replace=False).tolist())
            if cand is None:
                return None
            if n_tot > k:
                return cand[idx].flatten(0, 1).to(device=device, dtype=dtype)
        if cand is None:
            return None
        key = (label, device, dtype)
        if key not in self._dev_bank:
            self._dev_bank[key] = cand[:k].flatten(0, 1).to(device=device, dtype=dtype)
        return self._dev_bank[key]

    def _width(self):
        for v in self.query_bank.values():
            if torch.is_tensor(v) and v.numel():
                return v.shape[-1]
        return 1

    def select(self, batched_labels, batched_positive_maps, T, device, dtype):
        """-> vision [B, V, C] (zero padded), idx [B, T, S] int32 (-1 padded).  Labels without bank rows are skipped
        (text-only for that label, like the reference)."""
        per_image, tok_lists = [], []
        for labels, pmap in zip(batched_labels, batched_positive_maps):
            rows, owners = [], [[] for _ in range(T)]
            base = 0
            draw = not self.deterministic(labels)
            for lab in labels:
                r = self._rows(lab, device, dtype, draw)
                if r is None:
                    continue
                rows.append(r)
                for t in pmap[lab]:
                    owners[t].extend(range(base, base + r.shape[0]))
                base += r.shape[0]
            per_image.append(torch.cat(rows) if rows else torch.zeros(0, self._width(), device=device, dtype=dtype))
            tok_lists.append(owners)
        vision = torch.nn.utils.rnn.pad_sequence(per_image, batch_first=True)
        S = max(1, max(len(o) for owners in tok_lists for o in owners))
        idx = torch.full((len(tok_lists), T, S), -1, dtype=torch.int32)
        for b, owners in enumerate(tok_lists):
            for t, o in enumerate(owners):
                if o:
                    idx[b, t, :len(o)] = torch.tensor(sorted(o), dtype=torch.int32)
        return vision.contiguous(), idx.to(device, non_blocking=True)

    def select_cached(self, key, labels, positive_map, B, T, device, dtype):
        """Memoised `select` for B images sharing one caption (eval protocol): no per-call host or device work."""
        if not self.deterministic(labels):
            return self.select([labels] * B, [positive_map] * B, T, device, dtype)
        k = (key, B, T, str(device), dtype)
        hit = self._sel_cache.get(k)
        if hit is None:
            if len(self._sel_cache) > 128:
                self._sel_cache.clear()
            from .graph_runner import memoised
            hit = self._sel_cache[k] = memoised(self.select([labels] * B, [positive_map] * B, T, device, dtype))
        return hit

    def forward(self, batched_label_list, batched_location_map, batched_pos_labels=None):
        """Reference-compatible output (queries, 0/1 attention masks [B,V,T], has_vision_query)."""
        if self.query_bank is None:
            return None, None, None
        qs, ms, has = [], [], []
        for labels, maps in zip(batched_label_list, batched_location_map):
            q_img, m_img = [], []
            flags = []
            draw = not self.deterministic(labels)
            for lab, loc in zip(labels, maps):
                r = self._rows(lab, loc.device, loc.dtype, draw)
                flags.append(0 if r is None else 1)
                if r is None:
                    continue
                q_img.append(r)
                m_img.append(loc[None].expand(r.shape[0], -1))
            qs.append(torch.cat(q_img))
            ms.append(torch.cat(m_img))
            has.append(flags)
        q = torch.nn.utils.rnn.pad_sequence(qs, batch_first=True)
        m = torch.nn.utils.rnn.pad_sequence(ms, batch_first=True).clone()
        m[m != 0] = 1
        return q, m, has


def build_query_selector(cfg):
    return QuerySelector(cfg)
