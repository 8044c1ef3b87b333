"""Timing probe for the fused DCNv2 kernel on the P3 / P5 shapes of the benchmark (debug tool, not part of the product)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mq_det_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (H, W) in ((100, 168), (25, 42), (7, 11)):
    x = torch.randn(8, H, W, 256, generator=g).half().to(dev)
    om = (torch.randn(8, 27, H, W, generator=g) * 1.0).to(dev)
    w = (torch.randn(256, 2304, generator=g) / 48).half().to(dev)
    b = torch.randn(256, generator=g).half().to(dev)
    for _ in range(3):
        ops.dcnv2(x, om, w, b, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dcnv2(x, om, w, b, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f"variant={os.environ.get('MQ_DCN_VARIANT', '0')} HxW={H}x{W}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us / launch", flush=True)
