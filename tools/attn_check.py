"""Diagnostic: the engine's attention core (quantise -> Q K^T -> softmax + quantise -> P V) stage by stage against torch on
the same integer codes, for several (heads, head dim, queries, keys)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from types import SimpleNamespace
from edadm import ops
from edadm.engine import Engine
dev = torch.device("cuda", 0)


def q_(delta, zp, bits=8):
    return SimpleNamespace(delta=torch.tensor(delta, device=dev), zero_point=torch.tensor(float(zp), device=dev), n_levels=2 ** bits)


def case(B, heads, d, Nq, Nk, v_transposed=False):
    g = torch.Generator().manual_seed(B * 1000 + heads * 100 + d)
    hd = heads * d
    q = torch.randn(B * Nq, hd, generator=g).to(dev)
    k = torch.randn(B * Nk, hd, generator=g).to(dev)
    v = torch.randn(B * Nk, hd, generator=g).to(dev)
    aq, ak, av, aw = q_(0.03, 128), q_(0.031, 127), q_(0.029, 128), q_(1 / 255.0, 0)
    eng = Engine.__new__(Engine)
    eng.dev, eng._attn_cache = dev, {}
    eng.fused_attention = os.environ.get("UNFUSED", "0") != "1"        # K6f (default) or the three-kernel path
    scale = d ** -0.5
    out = eng.attention(q, k, v, B, Nq, Nk, heads, d, aq, ak, av, aw, scale)
    # torch on the same codes
    def codes(x, qz):
        return torch.clamp(torch.round(x / qz.delta) + qz.zero_point, 0, 255) - qz.zero_point
    cq, ck, cv = codes(q, aq), codes(k, ak), codes(v, av)
    sp = lambda t, n: t.reshape(B, n, heads, d).permute(0, 2, 1, 3)
    s = torch.einsum("bhid,bhjd->bhij", sp(cq, Nq).double(), sp(ck, Nk).double()) * float(aq.delta * ak.delta) * scale
    p = torch.softmax(s.float(), -1)
    cp = torch.clamp(torch.round(p / aw.delta) + aw.zero_point, 0, 255) - aw.zero_point
    o = torch.einsum("bhij,bhjd->bhid", cp.double(), sp(cv, Nk).double()) * float(aw.delta * av.delta)
    ref = o.permute(0, 2, 1, 3).reshape(B * Nq, hd).float()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    # stage 1 alone
    qh, kh = ops.quant_f16(q, eng._aq(aq)[0]), ops.quant_f16(k, eng._aq(ak)[0])
    e_codes = max((qh.float() - cq).abs().max().item(), (kh.float() - ck).abs().max().item())
    s_e = ops.gemm_f16_nt(qh, hd, Nq * hd, kh, hd, Nk * hd, B, Nq, Nk, d, float(aq.delta * ak.delta) * scale, inner=heads,
                          strideA_i=d, strideB_i=d)
    e_s = (s_e.reshape(B, heads, Nq, Nk).double() - s).abs().max().item() / s.abs().max().item()
    print("B=%d heads=%d d=%3d Nq=%4d Nk=%4d: codes err %.1f | scores rel err %.2e | output rel err %.2e %s" % (
        B, heads, d, Nq, Nk, e_codes, e_s, err, "  <-- WRONG" if err > 3e-2 else "  (one probability code on a rounding boundary)" if err > 1e-3 else ""))


for args in ((2, 1, 32, 16, 16), (2, 2, 16, 16, 16), (2, 2, 16, 16, 7), (2, 8, 8, 16, 16), (2, 8, 8, 16, 77), (2, 8, 16, 64, 64),
             (2, 8, 40, 64, 64), (2, 8, 40, 64, 77), (2, 8, 80, 16, 77), (2, 8, 160, 16, 77), (2, 4, 8, 16, 16), (2, 4, 24, 32, 32),
             (1, 8, 40, 1024, 1024), (2, 8, 24, 1024, 1024), (2, 8, 96, 64, 64), (1, 8, 40, 4096, 4096), (3, 8, 40, 4096, 77),
             (2, 1, 256, 256, 256), (2, 1, 384, 256, 256), (1, 1, 384, 1024, 1024)):
    case(*args)


def timing(B, heads, d, N, label, Nk=None):
    """fused K6f vs the three-kernel path on coded f16 operands"""
    import time
    g = torch.Generator().manual_seed(1)
    hd = heads * d
    Nk = Nk or N
    mk = lambda n: torch.randint(-120, 120, (B * n, hd), generator=g).to(dev).half()
    qh, kh, vh = mk(N), mk(Nk), mk(Nk)
    aq, ak, av, aw = q_(0.03, 128), q_(0.031, 127), q_(0.029, 128), q_(1 / 255.0, 0)
    res = {}
    for fused in (True, False):
        eng = Engine.__new__(Engine)
        eng.dev, eng._attn_cache, eng.fused_attention = dev, {}, fused
        f = lambda: eng.attention(qh, kh, vh, B, N, Nk, heads, d, aq, ak, av, aw, d ** -0.5, coded=True)
        out = f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        res[fused] = ((time.time() - t0) / 5 * 1e3, out)
    diff = (res[True][1] - res[False][1]).abs()
    print("%s: fused %.3f ms, three kernels %.3f ms (%.2fx); outputs differ in %.5f %% of elements, max %.3g of range" % (
        label, res[True][0], res[False][0], res[False][0] / res[True][0], 100.0 * (diff > 0).float().mean().item(),
        diff.max().item() / res[False][1].abs().max().item()))


timing(8, 8, 40, 4096, "SD 64x64 self-attention (8 rows, 8 heads x 40, 4096 keys)")
timing(8, 8, 80, 1024, "SD 32x32 self-attention (8 rows, 8 heads x 80, 1024 keys)")
timing(100, 8, 24, 1024, "Church 32x32 attention (100 rows, 8 heads x 24, 1024 keys)")
timing(100, 8, 48, 256, "Church 16x16 attention (100 rows, 8 heads x 48, 256 keys)")
timing(8, 8, 40, 4096, "SD 64x64 cross-attention (8 rows, 8 heads x 40, 4096 queries x 77 keys)", Nk=77)
timing(100, 1, 384, 1024, "LDM-4 32x32 self-attention (100 rows, 1 head x 384, 1024 keys)")
timing(50, 1, 384, 1024, "LDM-4 32x32 self-attention, shared half of a guidance pair (50 rows)")
timing(100, 1, 576, 256, "LDM-4 16x16 self-attention (100 rows, 1 head x 576, 256 keys)")
timing(100, 1, 960, 64, "LDM-4 8x8 self-attention (100 rows, 1 head x 960, 64 keys)")


def timing_i8(B, N, label):
    """K6w with int8 scores (edadm_attention_fused_i8qk) against its f16 form on the same codes"""
    import time
    g = torch.Generator().manual_seed(2)
    d = 384
    c = lambda n: torch.randint(0, 256, (B * n, d), generator=g).to(dev)
    cq, ck, cv = c(N), c(N), c(N)
    q8, k8 = (cq - 128).to(torch.int8), (ck - 128).to(torch.int8)
    qh, kh, vh = (cq - 128).half(), (ck - 128).half(), (cv - 128).half()
    pqp = ops.qp_tensor([(torch.tensor(1 / 255.0), torch.tensor(0.0), 255)], dev)
    alpha = 0.03 * 0.031 * d ** -0.5 * 0.4
    res = {}
    for name, f in (("int8 scores", lambda: ops.attention_fused_i8qk(q8, k8, vh, B, 1, N, N, d, alpha, 128.0, pqp, 0.029 / 255)),
                    ("f16", lambda: ops.attention_fused(qh, kh, vh, B, 1, N, N, d, alpha, pqp, 0.029 / 255))):
        out = f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        res[name] = ((time.time() - t0) / 10 * 1e3, out)
    diff = (res["int8 scores"][1] - res["f16"][1]).abs()
    print("%s: int8 scores %.3f ms, f16 %.3f ms (%.2fx); outputs differ in %.5f %% of elements" % (
        label, res["int8 scores"][0], res["f16"][0], res["f16"][0] / res["int8 scores"][0], 100.0 * (diff > 0).float().mean().item()))


timing_i8(100, 1024, "LDM-4 32x32 self-attention (100 rows, 1 head x 384, 1024 keys)")
timing_i8(50, 1024, "LDM-4 32x32 self-attention (50 rows)")
