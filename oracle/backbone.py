"""Oracle: Swin-T/L v1 backbone + FPN (test infrastructure, see oracle/__init__.py).

Restates maskrcnn_benchmark/modeling/backbone/swint.py:34-61,111-142,186-242,258-284,
347-386,412-428,591-615 and maskrcnn_benchmark/modeling/backbone/fpn.py:59-129,137-154
(+ backbone/__init__.py:55-74 for which stages feed the FPN) as pure functions over a flat
state_dict that uses the reference's parameter names.
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def rel_pos_index(ws):
    """swint.py:90-101 -- index into the (2ws-1)^2 bias table for every (query, key) pair."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)])            # [2, N]
    d = pos[:, :, None] - pos[:, None, :] + (ws - 1)               # [2, N, N] in [0, 2ws-2]
    return d[0] * (2 * ws - 1) + d[1]


def to_windows(x, ws):
    """[B,Hp,Wp,C] -> [B*nW, ws*ws, C]  (swint.py:34-45)."""
    B, Hp, Wp, C = x.shape
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def from_windows(w, ws, B, Hp, Wp):
    """inverse of to_windows (swint.py:48-61)."""
    C = w.shape[-1]
    w = w.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return w.reshape(B, Hp, Wp, C)


def shift_mask(Hp, Wp, ws, shift):
    """SW-MSA additive mask, -100 between different regions (swint.py:354-373)."""
    region = torch.zeros(1, Hp, Wp, 1)
    bounds_h = ((0, Hp - ws), (Hp - ws, Hp - shift), (Hp - shift, Hp))
    bounds_w = ((0, Wp - ws), (Wp - ws, Wp - shift), (Wp - shift, Wp))
    k = 0
    for h0, h1 in bounds_h:
        for w0, w1 in bounds_w:
            region[:, h0:h1, w0:w1, :] = k
            k += 1
    r = to_windows(region, ws).squeeze(-1)                          # [nW, N]
    diff = r[:, None, :] - r[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention(sd, p, xw, heads, ws, mask):
    """swint.py:111-142.  xw: [B*nW, N, C]; mask: [nW, N, N] or None."""
    Bw, N, C = xw.shape
    hd = C // heads
    qkv = _lin(sd, p + ".qkv", xw).reshape(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-1, -2)                                   # [Bw, heads, N, N]
    table = sd[p + ".relative_position_bias_table"]                  # [(2ws-1)^2, heads]
    bias = table[rel_pos_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(Bw // nW, nW, heads, N, N) + mask[None, :, None]).reshape(Bw, heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    return _lin(sd, p + ".proj", out)


def swin_block(sd, p, x, H, W, heads, ws, shift, mask):
    """swint.py:186-242 (pad AFTER norm1 -> pad tokens are exact zeros into qkv)."""
    B, L, C = x.shape
    y = _ln(sd, p + ".norm1", x).reshape(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    a = window_attention(sd, p + ".attn", to_windows(y, ws), heads, ws, mask if shift > 0 else None)
    y = from_windows(a, ws, B, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :H, :W, :].reshape(B, L, C)
    x = x + y
    h = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x))))
    return x + h


def patch_merging(sd, p, x, H, W):
    """swint.py:258-284."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.reshape(B, -1, 4 * C)
    return F.linear(_ln(sd, p + ".norm", x), sd[p + ".reduction.weight"])


def swin_forward(sd, p, img, spec):
    """swint.py:591-615 -> [c2, c3, c4, c5] NCHW.  `p` is e.g. 'backbone.body'."""
    ws = spec.window
    _, _, H0, W0 = img.shape
    if W0 % 4:
        img = F.pad(img, (0, 4 - W0 % 4))
    if H0 % 4:
        img = F.pad(img, (0, 0, 0, 4 - H0 % 4))
    x = F.conv2d(img, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=4)
    B, C, H, W = x.shape
    x = _ln(sd, p + ".patch_embed.norm", x.flatten(2).transpose(1, 2))
    outs = []
    for i, (depth, heads) in enumerate(zip(spec.swin_depths, spec.swin_heads)):
        Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
        mask = shift_mask(Hp, Wp, ws, ws // 2)
        for j in range(depth):
            x = swin_block(sd, f"{p}.layers.{i}.blocks.{j}", x, H, W, heads, ws,
                           0 if j % 2 == 0 else ws // 2, mask)
        # per-stage output norm; norm0 is Identity for *-RETINANET (swint.py:544-552)
        y = x if i == 0 else _ln(sd, f"{p}.norm{i}", x)
        outs.append(y.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        if i < len(spec.swin_depths) - 1:
            x = patch_merging(sd, f"{p}.layers.{i}.downsample", x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


def fpn_forward(sd, p, feats):
    """fpn.py:59-129 with in_channels_list=[0,c3,c4,c5] (c2 unused), no GN/ReLU, P6/P7 from P5
    (LastLevelP6P7.use_P5, fpn.py:148-154).  `p` is e.g. 'backbone.fpn'."""
    def conv(name, x, stride=1, pad=0):
        return F.conv2d(x, sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], stride=stride, padding=pad)
    c3, c4, c5 = feats[-3], feats[-2], feats[-1]
    inner = conv("fpn_inner4", c5)
    results = [conv("fpn_layer4", inner, pad=1)]
    for feat, idx in ((c4, 3), (c3, 2)):
        lat = conv(f"fpn_inner{idx}", feat)
        inner = lat + F.interpolate(inner, size=lat.shape[-2:], mode="nearest")
        results.insert(0, conv(f"fpn_layer{idx}", inner, pad=1))
    p6 = conv("top_blocks.p6", results[-1], stride=2, pad=1)
    p7 = conv("top_blocks.p7", F.relu(p6), stride=2, pad=1)
    return results + [p6, p7]
