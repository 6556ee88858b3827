"""Diagnostic: in-kernel cycle stamps of k_conv3_direct (libedadm_stamps.so, `make -C eda-dm_amd/csrc stamps`): per wave,
averaged over all waves of a launch -- prologue (entry -> first barrier passed), waits (vmcnt + barrier in front of the
other steps), compute (the rest of the main loop), epilogue (incl. draining its stores)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EDADM_LIB_PATH"] = os.path.join(ROOT, "eda-dm_amd", "csrc", os.environ.get("STAMPS_LIB", "libedadm_stamps.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, lib
dev = torch.device("cuda", 0)
L = lib.load()


def case(B, H, Cin, N, residual):
    W = H
    a = torch.randint(-128, 128, (B, H, W, Cin), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, 3, 3, Cin), dtype=torch.int8, device=dev)
    wdc = ops.conv3_pack_w(w, N, Cin)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    res = torch.randn(B * H * W, N, device=dev) if residual else None
    out = torch.empty(B * H * W, N, device=dev)
    f = lambda: ops.qconv3_i8_direct(a, wdc, B, H, W, Cin, N, 0, sc, bs, out, residual=res)
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(3):
        f(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    n = buf[4]
    M, K = B * H * W, 9 * Cin
    us = e0.elapsed_time(e1) * 1e3
    mfma = (Cin // 64) * 3 * 36 * 32 * 2            # cycles per wave at the MFMA's rate (two waves share a SIMD)
    print("M=%6d N=%4d K=%5d res=%d | %6.1f us %6.0f TOP/s | per wave ticks: setup=%5.0f prologue=%5.0f waits=%6.0f (vmcnt %6.0f) compute=%6.0f epilogue=%6.0f "
          "total=%6.0f | MFMA-rate cycles %d" % (M, N, K, residual, us, 2.0 * M * N * K / us / 1e6, buf[7] / n, buf[0] / n, buf[1] / n, buf[6] / n,
                                                  buf[2] / n, buf[3] / n, buf[5] / n, mfma))



SHAPES = ((100, 64, 192, 192), (100, 32, 384, 384), (100, 16, 576, 576), (100, 8, 960, 960), (100, 64, 576, 192),
          (100, 8, 1920, 960), (100, 16, 1152, 576))
if os.environ.get("SHAPES"):
    SHAPES = tuple(SHAPES[int(i)] for i in os.environ["SHAPES"].split(","))
for B, H, Cin, N in SHAPES:
    for residual in (0, 1):
        case(B, H, Cin, N, residual)
