"""GPU diagnosis (round 3, call 4): which kernel selection moves the full-depth error ladder of the 141-token case?  The fp32 oracle runs once;
the product is rebuilt per selection (ops.configure() + a fresh plan) and the floor ratios of a few stages are printed."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_checks as pc  # noqa: E402
from mq_det_amd import ops  # noqa: E402

ops.load_library()
dev = torch.device("cuda:0")
KEEP = ("swin c5", "fpn p3", "language hidden", "head layer 0: image tokens after VLFuse", "head layer 2: image tokens after DyConv", "head layer 5: text hidden",
        "head layer 5: image tokens after DyConv", "bbox_reg lvl0", "centerness lvl0", "dot-product logits lvl0", "class scores lvl0", "dot-product logits lvl1")
SELS = [("default", {}),
        ("round-2 set", {"MQ_LN_VARIANT": "1", "MQ_OFFSET_CONV_VARIANT": "1", "MQ_PATCH_MERGE_FUSED": "0", "MQ_FPN_VIA_DCN": "0", "MQ_NMS_EARLY_STOP": "0",
                         "MQ_ATTN_RESIDENT": "0", "MQ_SWIN_MLP_VARIANT": "1", "MQ_ALIGN_FUSED": "0"}),
        ("streaming attention", {"MQ_ATTN_RESIDENT": "0"}), ("unfused heads", {"MQ_ALIGN_FUSED": "0"}), ("swin mlp v1", {"MQ_SWIN_MLP_VARIANT": "1"}),
        ("fpn convs own kernel", {"MQ_FPN_VIA_DCN": "0"}), ("swin mlp2 table", {"MQ_SWIN_MLP2_FLAGS": "2"})]
for name, env in SELS:
    for k in list(os.environ):
        if k.startswith("MQ_") and k not in ("MQ_LADDER_OUT",):
            del os.environ[k]
    os.environ.update(env)
    ops.configure()
    for key in [k for k in pc._CACHE if isinstance(k, tuple) and k and k[0] == "bench"]:
        del pc._CACHE[key]
    t0 = time.time()
    res = pc.check_benchmark_config(dev, "long", ((800, 1333),))
    print(f"== {name} ({time.time() - t0:.0f} s): " + "  ".join(
        f"{[k for k in KEEP if k in r['name']][0][:22]}={r.get('ratio_mean', 0):.2f}" for r in res if any(k in r["name"] for k in KEEP)), flush=True)
