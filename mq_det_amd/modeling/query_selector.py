"""Vision-query selection (host-side glue, tiny compute).

Reference: modeling/query_selector/query_selector.py:8-116 (bank `{label: Tensor[n, scales, C]}`, <= k rows per
label, `[V, T]` 0/1 attention mask, `pad_sequence`) and generalized_vl_rcnn_new.py:295-305 (label -> token map).
MI355X-first difference: besides the dense mask the selector emits the gather index `idx[b, t, :]` (the
vision rows each text token may attend to) that the reference re-derives from the mask with a top-k trick in
EACH of the 6 GCP layers (modeling_bert_new.py:40-63,162-184); it is built once, on the host, from the
positive_map the caller already holds -- no device work, no sync.
"""
import os

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

    def _rows(self, label, device, dtype):
        key = (label, device, dtype)
        if key not in self._dev_bank:
            cand = self.query_bank[label]
            n = min(len(cand), self.num_query_per_class)
            # eval: sorted(np.random.choice(len, n, replace=False)) -- deterministic when len == n (the published
            # banks); otherwise we take the first n rows (documented deviation)
            self._dev_bank[key] = cand[:n].flatten(0, 1).to(device=device, dtype=dtype)
        return self._dev_bank[key]

    def select(self, batched_labels, batched_positive_maps, T, device, dtype):
        """-> vision [B, V, C] (zero padded), idx [B, T, S] int32 (-1 padded)."""
        per_image, tok_lists = [], []
        for labels, pmap in zip(batched_labels, batched_positive_maps):
            rows, owners = [], [[] for _ in range(T)]
            base = 0
            for lab in labels:
                r = self._rows(lab, device, dtype)
                rows.append(r)
                for t in pmap[lab]:
                    owners[t].extend(range(base, base + r.shape[0]))
                base += r.shape[0]
            per_image.append(torch.cat(rows) if rows else torch.zeros(0, 1, device=device, dtype=dtype))
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
        k = (key, B, T, str(device), dtype)
        hit = self._sel_cache.get(k)
        if hit is None:
            if len(self._sel_cache) > 128:
                self._sel_cache.clear()
            hit = self._sel_cache[k] = self.select([labels] * B, [positive_map] * B, T, device, dtype)
        return hit

    def forward(self, batched_label_list, batched_location_map, batched_pos_labels=None):
        """Reference-compatible output (queries, 0/1 attention masks [B,V,T], has_vision_query)."""
        if self.query_bank is None:
            return None, None, None
        qs, ms, has = [], [], []
        for labels, maps in zip(batched_label_list, batched_location_map):
            q_img, m_img = [], []
            for lab, loc in zip(labels, maps):
                r = self._rows(lab, loc.device, loc.dtype)
                q_img.append(r)
                m_img.append(loc[None].expand(r.shape[0], -1))
            qs.append(torch.cat(q_img))
            ms.append(torch.cat(m_img))
            has.append([1] * len(labels))
        q = torch.nn.utils.rnn.pad_sequence(qs, batch_first=True)
        m = torch.nn.utils.rnn.pad_sequence(ms, batch_first=True).clone()
        m[m != 0] = 1
        return q, m, has


def build_query_selector(cfg):
    return QuerySelector(cfg)
