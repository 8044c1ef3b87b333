#!/bin/bash
# Round 3, GPU call 16 (the last 90 seconds): mq_add_upsample_nearest on the device -- equality with F.interpolate + add at the FPN's sizes,
# timing of both, and the Swin + FPN parity check with it on.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp MQ_FPN_TOPDOWN_FUSED=1
timeout 70 python - <<'PY'
import sys, torch, torch.nn.functional as F
sys.path.insert(0, "tests")
from mq_det_amd import ops
ops.load_library(); ops.configure(None)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for (H, W), (Hc, Wc) in (((100, 168), (50, 84)), ((50, 84), (25, 42)), ((13, 21), (7, 11))):
    lat = torch.randn(8, H, W, 256, generator=g).half().to(dev); co = torch.randn(8, Hc, Wc, 256, generator=g).half().to(dev)
    ref = (lat + F.interpolate(co.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)).contiguous()
    got = ops.add_upsample_nearest_(lat.clone(), co)
    t_new = timeit(lambda: ops.add_upsample_nearest_(lat, co))
    t_old = timeit(lambda: (lat + F.interpolate(co.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)).contiguous())
    print(f"{H}x{W} <- {Hc}x{Wc}: equal={torch.equal(got, ref)}  fused {t_new*1e3:.1f} us  interpolate+add {t_old*1e3:.1f} us")
import parity_checks as pc
res = pc.check_swin_fpn(dev)
print("check_swin_fpn with FPN_TOPDOWN_FUSED=1:", "ALL_OK" if all(r["ok"] for r in res) else "FAILED", ops.KERNELS["FPN_TOPDOWN_FUSED"])
PY
