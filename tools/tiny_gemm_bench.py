"""Diagnostic: device time of the M = 100 GEMMs (time-embedding / one-token-context projections) via a kernel trace-free
proxy: N back-to-back launches inside a captured HIP graph, replayed."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
for (M, N, K) in ((100, 960, 512), (100, 960, 960), (100, 576, 512), (100, 384, 384), (100, 192, 768), (128, 960, 512), (256, 960, 512), (100, 960, 64)):
    A = torch.randint(-100, 100, (M, K), dtype=torch.int8, device=dev)
    W = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    sc = torch.rand(N, device=dev); b = torch.rand(N, device=dev)
    out = torch.empty(M, N, device=dev)
    run = lambda: ops.qgemm_i8(A, W, M, N, K, sc, b, out)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                run()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.time(); g.replay(); torch.cuda.synchronize(); dt = time.time() - t0
    print("M=%d N=%d K=%d: %.1f us per launch in a graph of 200" % (M, N, K, dt / 200 * 1e6))
