"""GPU probe (pure torch): which multi-stream topology does hipStreamEndCapture / hipGraphInstantiate survive?  One variant per process."""
import os, sys, time, torch
v = os.environ.get("PROBE", "lanes")
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
nl = int(os.environ.get("PROBE_LANES", "2")); ns = int(os.environ.get("PROBE_SIDE", "5"))
lanes = [torch.cuda.Stream() for _ in range(nl)]
side = [[torch.cuda.Stream() for _ in range(ns)] for _ in range(nl)]
def prog(x):
    main = torch.cuda.current_stream(); outs = []; done = None
    if v in ("prefork", "prefork_joinall"):                      # every stream enters the capture from the ORIGIN stream first
        for m in range(nl):
            for t in side[m]: t.wait_stream(main)
    for m in range(nl):
        s = lanes[m]; s.wait_stream(main)
        with torch.cuda.stream(s):
            if done is not None and v in ("lanes", "flat_event", "prefork", "prefork_joinall", "nested_joinall"): s.wait_event(done)
            y = x[m::nl] @ a
            if v in ("lanes", "flat_event", "noevent", "prefork", "prefork_joinall", "nested_joinall"):
                done = torch.cuda.Event(); done.record(s)
            parts = []
            if v in ("lanes", "noevent", "nested_noevent", "prefork", "prefork_joinall", "nested_joinall"):
                for t in side[m]: t.wait_stream(s)
                for t in side[m]:
                    with torch.cuda.stream(t): parts.append((y @ a).relu())
                for t in side[m]: s.wait_stream(t)
            else:
                parts = [(y @ a).relu() for _ in range(ns)]
            val, idx = torch.topk(sum(parts).float(), 100, dim=1)
            outs.append(val)
    for s in lanes: main.wait_stream(s)
    if v.endswith("joinall"):
        for m in range(nl):
            for t in side[m]: main.wait_stream(t)
    return torch.cat(outs)
x = torch.randn(512, 2048, device=dev, dtype=torch.float16)
if v != "twographs":
    prog(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = prog(x)
    g.replay(); torch.cuda.synchronize()
    print("PROBE_OK", v, nl, ns, float(out.sum()), flush=True)
else:
    # per lane two graphs (stage 1: a big GEMM chain; stage 2: a chain of small kernels), replayed on the lane's stream with a stagger event
    big = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    def st1(x):
        y = x
        for _ in range(6): y = (y @ big) * 0.01
        return y
    def st2(y):
        z = y[:64]
        for _ in range(200): z = z * 1.0001 + 0.1
        return z
    xs = [torch.randn(8192, 8192, device=dev, dtype=torch.float16) for _ in range(2)]
    graphs = []
    for m in range(2):
        st2(st1(xs[m])); torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1): y = st1(xs[m])
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2): z = st2(y)
        graphs.append((g1, g2, z))
    def run(staggered):
        main = torch.cuda.current_stream(); done = None
        for m in range(2):
            s = lanes[m] if staggered else main
            if staggered: s.wait_stream(main)
            with torch.cuda.stream(s):
                if done is not None and staggered: s.wait_event(done)
                graphs[m][0].replay()
                done = torch.cuda.Event(); done.record(s)
                graphs[m][1].replay()
        if staggered:
            for s in lanes: main.wait_stream(s)
    for mode in (False, True, False, True):
        run(mode); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): run(mode)
        torch.cuda.synchronize()
        print("PROBE_OK twographs staggered=%s ms=%.3f" % (mode, (time.perf_counter() - t0) * 100), flush=True)
