"""Oracle: BERT-base language backbone with the Gated Class-scalable Perceiver plug-in
(test infrastructure, see oracle/__init__.py).

Restates
  * HF transformers `BertEmbeddings` / `BertLayer` eval arithmetic (third-party, unpinned in the
    reference's requirements.txt:12; written spec = SURVEY.md Appendix A and the reference's own
    clamped copy maskrcnn_benchmark/modeling/rpn/modeling_bert.py:39-272),
  * maskrcnn_benchmark/modeling/language_backbone/modeling_bert_new.py:40-63 (index padding),
    :95-102 (gather), :115-126 (FeedForward), :128-248 (MaskedCrossAttention), :250-374
    (GatedCrossAttentionBlock), :377-448 (PreSelect), :457-519 (embeddings), :545-639 (encoder),
  * maskrcnn_benchmark/modeling/language_backbone/bert_model_new.py:39-104 (wrapper outputs).
"""
import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def clamp5e4(x):
    """rpn/modeling_bert.py:34-36."""
    return torch.clamp(x, min=-50000, max=50000)


# --------------------------------------------------------------------------- BERT
def bert_embeddings(sd, p, input_ids, eps=1e-12, position_ids=None):
    """modeling_bert_new.py:487-517: word + token_type(0) + absolute position, LayerNorm.  position_ids [B,T]: the
    per-sub-sentence positions MQ-GroundingDINO passes (groundingdino.py:546-549); default 0..T-1."""
    T = input_ids.shape[1]
    e = sd[p + ".word_embeddings.weight"][input_ids]
    e = e + sd[p + ".token_type_embeddings.weight"][0][None, None]
    e = e + (sd[p + ".position_embeddings.weight"][:T][None] if position_ids is None
             else sd[p + ".position_embeddings.weight"][position_ids])
    return _ln(sd, p + ".LayerNorm", e, eps)


def extended_mask(attention_mask, dtype=torch.float32):
    """HF get_extended_attention_mask: additive key-padding mask [B,1,1,T]; a [B,T,T] mask (the sub-sentence block mask of
    MQ-GroundingDINO, bertwarper.py:140-144) becomes [B,1,T,T]."""
    m = (attention_mask[:, None, :, :] if attention_mask.dim() == 3 else attention_mask[:, None, None, :]).to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


def bert_self_attention(sd, p, x, ext_mask, heads, clamp):
    """rpn/modeling_bert.py:71-176 (clamp=True) == HF BertSelfAttention (clamp=False)."""
    B, T, C = x.shape
    hd = C // heads

    def split(t):
        return t.reshape(B, T, heads, hd).permute(0, 2, 1, 3)
    q, k, v = (split(_lin(sd, f"{p}.{n}", x)) for n in ("query", "key", "value"))
    s = (q @ k.transpose(-1, -2)) / hd ** 0.5
    if clamp:
        s = clamp5e4(s)
    s = s + ext_mask
    ctx = s.softmax(-1) @ v
    return ctx.permute(0, 2, 1, 3).reshape(B, T, C)


def bert_layer(sd, p, x, ext_mask, heads, eps=1e-12, clamp=False):
    """One post-LN BERT layer.  clamp=True adds the +-50000 clamps of the VLDyHead copy
    (rpn/modeling_bert.py:137-142, 253-255, 269-271)."""
    ctx = bert_self_attention(sd, p + ".attention.self", x, ext_mask, heads, clamp)
    a = _ln(sd, p + ".attention.output.LayerNorm", _lin(sd, p + ".attention.output.dense", ctx) + x, eps)
    h = _lin(sd, p + ".intermediate.dense", a)
    if clamp:
        h = clamp5e4(h)
    h = F.gelu(h)
    if clamp:
        h = clamp5e4(h)
    o = _lin(sd, p + ".output.dense", h)
    if clamp:
        o = clamp5e4(o)
    o = _ln(sd, p + ".output.LayerNorm", o + a, eps)
    if clamp:
        o = clamp5e4(o)
    return o


# --------------------------------------------------------------------------- GCP
def feed_forward(sd, p, x):
    """modeling_bert_new.py:115-126: LN -> Linear(no bias) -> GELU -> Linear(no bias)."""
    h = _ln(sd, p + ".norm", x, 1e-5)
    return F.linear(F.gelu(F.linear(h, sd[p + ".linear1.weight"])), sd[p + ".linear2.weight"])


def padded_nonzero_index(a):
    """modeling_bert_new.py:40-63.  a: [B, M, N] 0/1 -> [B, M, S] indices of the non-zeros of
    each row in ascending order, padded with N; S = max count over the whole batch."""
    N = a.shape[-1]
    nz = a != 0
    S = int(nz.sum(-1).max())
    idx = torch.where(nz, torch.arange(N), torch.tensor(N))
    return idx.topk(k=S, dim=-1, largest=False).values[:, :, :S]


def masked_cross_attention(sd, p, x, vision, mask, heads, dim_head, sparse):
    """modeling_bert_new.py:186-248.
    x: [B, T, Cx] (queries), vision: [B, V, Cv] (keys/values), mask: [B, V, T] 0/1 or None.
    sparse=True reproduces `_construct_sparse_inputs` (:162-184): every text token attends to its
    own <=S gathered vision rows (pad index V -> appended all-zero row)."""
    B = x.shape[0]
    if sparse:
        V, C = vision.shape[1], vision.shape[2]
        vis = torch.cat([vision, vision.new_zeros(B, 1, C)], 1)
        idx = padded_nonzero_index(mask.transpose(2, 1))                 # [B, T, S]
        T, S = idx.shape[1], idx.shape[2]
        flat = (idx + torch.arange(B)[:, None, None] * (V + 1)).reshape(-1)
        vision = vis.reshape(-1, C)[flat].reshape(B * T, S, C)           # [B*T, S, C]
        x = x.reshape(B * T, 1, -1)
        mask = (idx != V).reshape(B * T, S, 1)                           # (b, v, t) convention
    x = _ln(sd, p + ".norm", x, 1e-5)
    vision = _ln(sd, p + ".norm_kv", vision.to(x.dtype), 1e-5)
    q = F.linear(x, sd[p + ".to_q.weight"])
    k, v = F.linear(vision, sd[p + ".to_kv.weight"]).chunk(2, -1)

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, dim_head).permute(0, 2, 1, 3)
    q, k, v = split(q) * dim_head ** -0.5, split(k), split(v)
    sim = q @ k.transpose(-1, -2)                                         # [b, h, t, v]
    if mask is not None and mask.numel() > 0:
        add = torch.zeros(mask.shape)
        add[mask == 0] = -1e4                                             # (:219-222)
        sim = sim + add.transpose(1, 2)[:, None]
    attn = sim.softmax(-1)
    if mask is not None and mask.numel() > 0:
        attn = attn * mask.transpose(1, 2)[:, None].to(attn.dtype)        # (:227-231)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(q.shape[0], q.shape[2], heads * dim_head)
    if sparse:
        out = out.reshape(B, -1, heads * dim_head)
    return F.linear(out, sd[p + ".to_out.weight"])


def gated_cross_attention_block(sd, p, x, vision, mask, spec):
    """modeling_bert_new.py:298-374 with CONDITION_GATE, NONLINEAR_GATE, NO_CAT (mq-glip-t.yaml)."""
    sup = masked_cross_attention(sd, p + ".attn", x, vision, mask, spec.gcp_heads, spec.gcp_dim_head, True)
    gate = feed_forward(sd, p + ".attn_gate", sup).tanh()                 # [B, T, 1]
    x = sup * gate + x
    return feed_forward(sd, p + ".ff", x) * sd[p + ".ff_gate"].tanh() + x


def pre_select(sd, p, vision, image, spec):
    """modeling_bert_new.py:398-409, 433-448: vision queries cross-attend to the pooled image tokens."""
    vision, image = vision * spec.vision_scale, image * spec.vision_scale
    for i in range(spec.pre_layers):
        q = f"{p}.layers.{i}"
        att = masked_cross_attention(sd, q + ".image_condition", vision, image, None,
                                     spec.gcp_heads, spec.pre_dim_head, False)
        res = F.linear(vision, sd[q + ".res_mapping.weight"]) if (q + ".res_mapping.weight") in sd else vision
        vision = att + res
        vision = feed_forward(sd, q + ".ff", vision) + vision
    return vision


def qv_bert(sd, p, input_ids, attention_mask, vision, images, vision_mask, spec, position_ids=None):
    """QVBertModel.forward (modeling_bert_new.py:690-848; the same arithmetic through BertModelWarper.forward,
    groundingdino_new/models/GroundingDINO/bertwarper.py:60-215) -> list of the hidden states of every
    layer (output_hidden_states[1:]).  `p` is e.g. 'language_backbone.body.model'."""
    x = bert_embeddings(sd, p + ".embeddings", input_ids, spec.bert_eps, position_ids)
    ext = extended_mask(attention_mask)
    use_vq = vision is not None and images is not None and vision.numel() > 0
    if use_vq:
        vision = pre_select(sd, p + ".pre_select", vision, images, spec)
    hidden = []
    for i in range(spec.bert_layers):
        if use_vq and i >= spec.qv_start:
            x = gated_cross_attention_block(sd, f"{p}.encoder.qv_layer.{i - spec.qv_start}",
                                            x, vision, vision_mask, spec)
        x = bert_layer(sd, f"{p}.encoder.layer.{i}", x, ext, spec.bert_heads, spec.bert_eps)
        hidden.append(x)
    return hidden


def language_backbone(sd, p, input_ids, attention_mask, vision, images, vision_mask, spec):
    """bert_model_new.BertEncoder.forward (bert_model_new.py:39-104).  `p`='language_backbone.body'."""
    hidden = qv_bert(sd, p + ".model", input_ids, attention_mask, vision, images, vision_mask, spec)
    n = spec.n_lang_layers
    feats = torch.stack(hidden[-n:], 1).mean(1) / n                       # (:61-66) incl. the /n quirk
    m = attention_mask.unsqueeze(-1).float()
    embedded = feats * m
    aggregate = embedded.sum(1) / attention_mask.sum(-1, keepdim=True).float()
    return {"aggregate": aggregate, "embedded": embedded, "masks": attention_mask, "hidden": hidden[-1]}
