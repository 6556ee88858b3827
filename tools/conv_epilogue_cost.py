"""Diagnostic: what each epilogue option of edadm_qconv3_i8_direct costs (row add, residual, GroupNorm partials)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B, H, Cin, N in ((100, 64, 384, 192), (100, 64, 192, 192), (100, 32, 384, 384)):
    a = torch.randint(-128, 128, (B, H, H, Cin), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, 3, 3, Cin), dtype=torch.int8, device=dev)
    wdc = ops.conv3_pack_w(w, N, Cin)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    M = B * H * H
    out = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
    ra = torch.randn(B, N, device=dev)
    ws = torch.empty(M // 64, N, 2, device=dev)
    for name, kw in (("plain", {}), ("rowadd", dict(rowadd=ra, rows_per_batch=H * H)), ("gn", dict(gn_ws=ws)),
                     ("rowadd+gn", dict(rowadd=ra, rows_per_batch=H * H, gn_ws=ws)), ("res", dict(residual=res)), ("res+gn", dict(residual=res, gn_ws=ws))):
        print(M, N, 9 * Cin, name, "%.1f us" % t(lambda: ops.qconv3_i8_direct(a, wdc, B, H, H, Cin, N, 0, sc, bs, out, **kw)))
