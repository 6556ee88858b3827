"""-m gpu: every HIP kernel, called through the C ABI (edadm.ops -> libedadm.so), against the
oracle / golden vectors.  Integer codes and index work bit-exact; float results within the
tolerance written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import qdiff_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.as_tensor(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from edadm import ops as _ops
    return _ops


def D(t):
    return T(t).float().cuda().contiguous() if not isinstance(t, torch.Tensor) else t.float().cuda().contiguous()


def close(a, b, rtol, atol):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def exact(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------ K1
def test_fake_quant_golden(ops, golden):
    g = golden("g3_uaq_forward")
    x, gy, u = D(g["x"]), D(g["gy"]), D(g["mask_u"])
    for c in sorted({k.split("/")[0] for k in g.files if "/" in k}):
        bits = int(c.split("_")[0][1:])
        delta = torch.tensor([float(c.split("_d")[1].split("_z")[0])], device="cuda")
        zp = torch.tensor([float(c.split("_z")[1])], device="cuda")
        qmax = 2 ** bits - 1
        out, codes = ops.fake_quant_fwd(x, delta, zp, qmax, want_codes=True)
        exact(codes, g[c + "/codes"])            # integer codes: bit-exact
        exact(out, g[c + "/out"])
        gx, gd = ops.fake_quant_bwd(gy, x, delta, zp, qmax)
        exact(gx, g[c + "/gx"])
        close(gd, g[c + "/gdelta"].reshape(1), rtol=2e-5, atol=1e-5)
        out = ops.fake_quant_fwd(x, delta, zp, qmax, u=u, prob=0.5)
        exact(out, g[c + "/train_out"])
        gx, gd = ops.fake_quant_bwd(gy, x, delta, zp, qmax, u=u, prob=0.5)
        exact(gx, g[c + "/train_gx"])
        close(gd, g[c + "/train_gdelta"].reshape(1), rtol=2e-5, atol=1e-5)


def test_fake_quant_per_channel_and_rng(ops):
    gen = torch.Generator().manual_seed(5)
    w = torch.randn(24, 10, 3, 3, generator=gen) * 0.3
    delta = torch.rand(24, generator=gen) * 0.05 + 0.01
    zp = torch.randint(0, 16, (24,), generator=gen).float()
    ref, codes = O.fake_quant_fwd(w, delta.view(-1, 1, 1, 1), zp.view(-1, 1, 1, 1), 16)
    out, c = ops.fake_quant_fwd(D(w), D(delta), D(zp), 15, inner=90, want_codes=True)
    exact(c, codes), exact(out, ref)
    # odd sizes take the scalar path
    w2 = torch.randn(7, 5, generator=gen)
    ref, codes = O.fake_quant_fwd(w2, delta[:7].view(-1, 1), zp[:7].view(-1, 1), 16)
    out, c = ops.fake_quant_fwd(D(w2), D(delta[:7]), D(zp[:7]), 15, inner=5, want_codes=True)
    exact(c, codes), exact(out, ref)
    # in-kernel RNG: about `prob` of the elements quantised, same seed -> same mask in fwd and bwd
    x = torch.randn(1 << 16, generator=gen).cuda()
    d1, z1 = torch.tensor([0.05], device="cuda"), torch.tensor([128.0], device="cuda")
    full = ops.fake_quant_fwd(x, d1, z1, 255)
    mixed = ops.fake_quant_fwd(x, d1, z1, 255, prob=0.5, seed=77)
    took_q = (mixed == full) & (full != x)
    frac = took_q.float().sum() / (full != x).float().sum()
    assert 0.48 < float(frac) < 0.52
    gx, _ = ops.fake_quant_bwd(torch.ones_like(x), x, d1, z1, 255, prob=0.5, seed=77)
    assert torch.equal(mixed, ops.fake_quant_fwd(x, d1, z1, 255, prob=0.5, seed=77))
    assert mixed.ne(ops.fake_quant_fwd(x, d1, z1, 255, prob=0.5, seed=78)).any()


# ------------------------------------------------------------------ K2
def test_adaround_golden(ops, golden):
    g = golden("g4_adaround")
    for c, bits in (("conv", 4), ("lin", 8)):
        w, delta, zp = D(g[c + "/w"]), D(g[c + "/delta"]).reshape(-1), D(g[c + "/zero_point"]).reshape(-1)
        qmax = 2 ** bits - 1
        a0 = ops.adaround_init_alpha(w, delta)
        close(a0, g[c + "/alpha0"], rtol=2e-6, atol=2e-6)
        gy = D(g[c + "/gy"])
        for tag, alpha in (("", D(g[c + "/alpha0"])), ("1", D(g[c + "/alpha1"]))):
            out = torch.empty_like(w)
            ops.adaround_fwd(w, alpha, out, delta, zp, qmax, True)
            close(out, g[c + "/soft_out" + tag], rtol=1e-6, atol=1e-7)
            close(ops.adaround_bwd(gy, w, alpha, delta, zp, qmax), g[c + "/galpha" + tag], rtol=2e-5, atol=1e-9)
        out = torch.empty_like(w)
        ops.adaround_fwd(w, D(g[c + "/alpha1"]), out, delta, zp, qmax, False)
        exact(out, g[c + "/hard_out1"])           # hard rounding decides integer weights: bit-exact


def test_adaround_split_views(ops):
    gen = torch.Generator().manual_seed(9)
    w = (torch.randn(16, 24, 3, 3, generator=gen) * 0.2).cuda()
    out = torch.zeros_like(w)
    for lo, hi in ((0, 8), (8, 24)):
        wv = w[:, lo:hi]
        delta = (torch.rand(16, generator=gen) * 0.03 + 0.01).cuda()
        zp = torch.full((16,), 8.0).cuda()
        alpha = ops.adaround_init_alpha(wv, delta)
        ref_a = O.adaround_init_alpha(wv.cpu(), delta.cpu().view(-1, 1, 1, 1))
        close(alpha, ref_a, rtol=2e-6, atol=2e-6)
        ops.adaround_fwd(wv, alpha, out[:, lo:hi], delta, zp, 15, True)
        ref = O.adaround_fwd(wv.cpu(), ref_a, delta.cpu().view(-1, 1, 1, 1), zp.cpu().view(-1, 1, 1, 1), 16, True)
        close(out[:, lo:hi], ref, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ K3
def test_mse_scores_pick_reference_candidate(ops, golden):
    g = golden("g2_act_init")
    for run in ("two/b8/sym", "pos/b8/sym", "softmax/b8/sym", "two/b4/sym"):
        x = T(g[run + "/step0/x"]).float()
        bits = int(run.split("/")[1][1:])
        n_levels = 2 ** bits
        one = "pos" if x.min() >= 0 else "neg" if x.max() <= 0 else "no"
        mn, mx = torch.aminmax(x)
        xr = torch.max(mn.abs(), mx)
        thres = xr / 100 * torch.arange(1, 101)
        new_min = torch.zeros_like(thres) if one == "pos" else -thres
        new_max = torch.zeros_like(thres) if one == "neg" else thres
        scale = torch.max((new_max - new_min) / float(n_levels - 1), O.EPS)
        zp = torch.clamp(-torch.round(new_min / scale), 0, n_levels - 1)
        mm = ops.minmax(D(x))
        exact(mm, torch.stack([mn, mx]))
        sc = ops.mse_scores_tensor(D(x), D(scale), D(zp), n_levels - 1).cpu()
        xf = x.reshape(1, -1)
        ref = []
        for i in range(100):
            xi = torch.max(torch.min((xf / scale[i]).round(), n_levels - 1 - zp[i]), -zp[i]) * scale[i]
            ref.append((xi - xf).abs().pow(2.4).mean())
        ref = torch.stack(ref)
        close(sc, ref, rtol=5e-5, atol=1e-12)
        ind = int(torch.argmin(sc))
        assert ind == int(torch.argmin(ref))
        delta, z = O.calculate_qparams(new_min[ind], new_max[ind], n_levels)
        exact(delta, g[run + "/step0/delta"]), exact(z, g[run + "/step0/zero_point"])


def test_mse_scores_channel(ops, golden):
    g = golden("g1_weight_init")
    for cname in ("conv_two", "lin_two", "conv_pos", "conv_zero_ch"):
        w = T(g["w/" + cname]).float()
        rows = w.shape[0]
        w2 = w.reshape(rows, -1)
        for bits in (4, 8):
            n_levels = 2 ** bits
            one = "pos" if w.min() >= 0 else "neg" if w.max() <= 0 else "no"
            mn, mx = torch.aminmax(w2, dim=1)
            xr = torch.max(mn.abs(), mx)
            cs, cz, cmin, cmax = [], [], [], []
            for i in range(1, 101):
                thres = xr / 100 * i
                nmin = torch.zeros_like(mn) if one == "pos" else -thres
                nmax = torch.zeros_like(mx) if one == "neg" else thres
                s, z = O.calculate_qparams(nmin, nmax, n_levels)
                cs.append(s), cz.append(z), cmin.append(nmin), cmax.append(nmax)
            sc = ops.mse_scores_channel(D(w2), D(torch.stack(cs)), D(torch.stack(cz)), n_levels - 1).cpu()
            big = torch.full_like(sc, 1000, dtype=torch.long)
            idx = torch.where(sc == sc.min(0, keepdim=True)[0], torch.arange(100).view(-1, 1).expand_as(sc), big).min(0)[0]
            bmin = torch.stack(cmin).gather(0, idx.view(1, -1))[0]
            bmax = torch.stack(cmax).gather(0, idx.view(1, -1))[0]
            delta, z = O.calculate_qparams(bmin, bmax, n_levels)
            key = "%s/b%d/sym" % (cname, bits)
            exact(delta, g[key + "/delta"].reshape(-1)), exact(z, g[key + "/zero_point"].reshape(-1))


# ------------------------------------------------------------------ K7 / K8 / K9 / K10
def test_lp_loss_golden(ops, golden):
    g = golden("g5_loss")
    for c in ("4d", "2d", "3d"):
        p, t = D(g[c + "/pred"]), D(g[c + "/tgt"])
        close(ops.lp_loss_fwd(p, t), g[c + "/loss"].reshape(1), rtol=2e-6, atol=0)
        one = torch.ones(1, device="cuda")
        close(ops.lp_loss_bwd(p, t, one), g[c + "/gpred"], rtol=2e-6, atol=1e-9)


def test_adam_matches_torch(ops):
    gen = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, generator=gen)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-2)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=0.0)
    p, m, v = p0.cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    for it in range(1, 21):
        gr = torch.randn(1000, generator=gen) * (0.1 if it % 3 else 1e-3)
        p_ref.grad = gr.clone()
        lr = 5e-2 * (1 + math.cos(math.pi * (it - 1) / 20)) / 2
        opt.step(), sched.step()
        hyper = torch.tensor([lr / (1 - 0.9 ** it), math.sqrt(1 - 0.999 ** it), 0.9, 0.999], device="cuda")
        ops.adam_step(p, gr.cuda(), m, v, hyper)
        close(p, p_ref.detach(), rtol=2e-6, atol=1e-5)   # 20 steps of <=5e-2 each: within 2e-4 of a step


def test_mix_and_ddim(ops, golden):
    gen = torch.Generator().manual_seed(4)
    a, b, u = (torch.randn(5000, generator=gen) for _ in range(3))
    u = torch.rand(5000, generator=gen)
    exact(ops.mix_where(a.cuda(), b.cuda(), 0.5, u=u.cuda()), torch.where(u < 0.5, a, b))
    g = golden("g10_steps")
    x, c, uc, Wm = T(g["gs/x"]), T(g["ps/c"]), T(g["ps/uc"]), T(g["gs/Wm"])
    app = lambda x_, t_, c_: torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
        + t_.float().view(-1, 1, 1, 1) / 1000.0
    b_ = O.ldm_linear_betas(1000, 0.0015, 0.0195)
    ac = np.cumprod(1.0 - b_, axis=0).astype(np.float32)
    sig, al, alp = O.make_ddim_sampling_parameters(ac, O.make_ddim_timesteps(20, 1000), 0.0)
    idx = g["psq/index"]
    tq = T(g["psq/t"])
    coef = torch.tensor(np.stack([np.sqrt(1 - al[idx]), np.sqrt(al[idx]), np.sqrt(alp[idx]),
                                  np.sqrt(1 - alp[idx] - sig[idx] ** 2), sig[idx]], 1), dtype=torch.float32).cuda()
    xp, p0 = ops.ddim_step(x.cuda(), app(x, tq, c).contiguous().cuda(), app(x, tq, uc).contiguous().cuda(), 3.0, coef,
                            want_x0=True)
    close(xp, g["psq/x_prev"], rtol=3e-5, atol=3e-6)
    close(p0, g["psq/pred_x0"], rtol=3e-5, atol=3e-6)


# ------------------------------------------------------------------ operand producers
def _qp(ops, entries):
    return ops.qp_tensor(entries, torch.device("cuda"))


def test_quant_i8_and_split(ops):
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(300, 64, generator=gen) * 2
    qp = _qp(ops, [(0.03, 128.0, 255), (0.011, 127.0, 255)])
    got = ops.quant_i8(x.cuda(), qp, split=32).cpu().int()
    c0 = torch.clamp(torch.round(x[:, :32] / 0.03) + 128, 0, 255) - 128
    c1 = torch.clamp(torch.round(x[:, 32:] / torch.tensor(0.011)) + 127, 0, 255) - 128
    exact(got, torch.cat([c0, c1], 1).int())
    x3 = torch.randn(50, 3, generator=gen)
    got = ops.quant_i8(x3.cuda(), _qp(ops, [(0.02, 128.0, 255)])).cpu().int()
    exact(got, (torch.clamp(torch.round(x3 / torch.tensor(0.02)) + 128, 0, 255) - 128).int())
    h = ops.quant_f16(x.cuda(), _qp(ops, [(0.03, 127.0, 255)]), premul=0.5).cpu().float()
    exact(h, torch.clamp(torch.round((x * 0.5) / torch.tensor(0.03)) + 127, 0, 255) - 127)


@pytest.mark.parametrize("B,HW,C,eps,silu", [(3, 64, 64, 1e-6, True), (2, 256, 192, 1e-5, True),
                                              (2, 16, 1920, 1e-5, False), (1, 1024, 320, 1e-6, True)])
def test_groupnorm_apply(ops, B, HW, C, eps, silu):
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(B, HW, C, generator=gen) * 2 + 0.3
    gamma, beta = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    ref = F.group_norm(x.permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), eps).permute(0, 2, 1)
    if silu:
        ref = ref * torch.sigmoid(ref)
    stats = ops.groupnorm_stats(x.cuda(), 32, eps)
    qp = _qp(ops, [(0.02, 128.0, 255), (0.05, 127.0, 255), (0.3, 8.0, 15)])
    out, qs = ops.groupnorm_apply(x.cuda(), stats, gamma.cuda(), beta.cuda(), 32, silu, qp=qp, nq=3, want_f32=True)
    close(out, ref, rtol=2e-5, atol=2e-5)           # fp32 GroupNorm vs fp64 reference
    for (d, z, qm), q in zip(((0.02, 128.0, 255), (0.05, 127.0, 255), (0.3, 8.0, 15)), qs):
        mine = q.cpu().int()
        exact(mine, (torch.clamp(torch.round(out.cpu() / torch.tensor(d)) + z, 0, qm) - 128).int())
    # scale-shift form (use_scale_shift_norm, openaimodel.py:263-267)
    ss = torch.randn(B, 2 * C, generator=gen) * 0.2
    out2, _ = ops.groupnorm_apply(x.cuda(), stats, gamma.cuda(), beta.cuda(), 32, False, want_f32=True,
                                  scale_shift=ss.cuda())
    base = F.group_norm(x.permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), eps).permute(0, 2, 1)
    close(out2, base * (1 + ss[:, None, :C].double()) + ss[:, None, C:].double(), rtol=2e-5, atol=2e-5)


def test_layernorm_geglu_silu_softmax(ops):
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(77, 384, generator=gen) * 3
    gamma, beta = torch.randn(384, generator=gen), torch.randn(384, generator=gen)
    ref = F.layer_norm(x.double(), (384,), gamma.double(), beta.double(), 1e-5)
    qp = _qp(ops, [(0.02, 128.0, 255), (0.03, 128.0, 255), (0.04, 127.0, 255)])
    out, qs = ops.layernorm_quant(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, qp=qp, nq=3, want_f32=True)
    close(out, ref, rtol=2e-5, atol=2e-5)
    exact(qs[2].cpu().int(), (torch.clamp(torch.round(out.cpu() / torch.tensor(0.04)) + 127, 0, 255) - 128).int())
    h = torch.randn(40, 256, generator=gen)
    a, gate = h.chunk(2, -1)
    refg = a * F.gelu(gate)
    got = ops.geglu_quant_i8(h.cuda(), _qp(ops, [(0.01, 128.0, 255)])).cpu().int()
    want = (torch.clamp(torch.round(refg / torch.tensor(0.01)) + 128, 0, 255) - 128).int()
    assert (got - want).abs().max() <= 1 and (got != want).float().mean() < 2e-3   # erf/gelu last-ulp ties
    s = torch.randn(9, 768, generator=gen)
    got = ops.silu_quant_i8(s.cuda(), _qp(ops, [(0.01, 30.0, 255)])).cpu().int()
    want = (torch.clamp(torch.round(O.silu(s) / torch.tensor(0.01)) + 30, 0, 255) - 128).int()
    assert (got - want).abs().max() <= 1 and (got != want).float().mean() < 2e-3
    close(ops.silu(s.cuda()), O.silu(s), rtol=1e-6, atol=1e-7)
    sc = torch.randn(130, 200, generator=gen) * 3
    p = torch.softmax(sc, -1)
    got = ops.softmax_quant_f16(sc.cuda(), _qp(ops, [(1 / 255.0, 0.0, 255)]), ldo=208).cpu().float()
    want = torch.clamp(torch.round(p / torch.tensor(1 / 255.0)), 0, 255)
    assert (got[:, :200] - want).abs().max() <= 1 and (got[:, :200] != want).float().mean() < 2e-3
    assert got[:, 200:].abs().max() == 0


def test_layout_helpers(ops):
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(3, 5, 7, 9, generator=gen)
    exact(ops.nchw_to_nhwc(x.cuda()), x.permute(0, 2, 3, 1))
    exact(ops.nhwc_to_nchw(x.permute(0, 2, 3, 1).contiguous().cuda()), x)
    a, b = torch.randn(2, 4, 4, 8, generator=gen), torch.randn(2, 4, 4, 12, generator=gen)
    exact(ops.concat_c(a.cuda(), b.cuda()), torch.cat([a, b], -1))
    exact(ops.upsample2_nhwc(a.cuda()), F.interpolate(a.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1))
    close(ops.avgpool2_nhwc(a.cuda()), F.avg_pool2d(a.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1), 1e-6, 1e-7)
    exact(ops.add(a.cuda(), a.cuda()), a + a)
    h = torch.randn(2, 37, 24, generator=gen).half()
    t = ops.transpose_f16(h.cuda(), 24, 37 * 24, 2, 37, 24, 40).cpu()
    exact(t[:, :, :37], h.permute(0, 2, 1)), exact(t[:, :, 37:], torch.zeros(2, 24, 3).half())
    codes = torch.randint(0, 16, (6, 10), generator=gen)
    packed = (codes.reshape(-1)[0::2] | (codes.reshape(-1)[1::2] << 4)).to(torch.uint8)
    zp = torch.tensor([7., 8., 7., 8., 8., 7.])
    exact(ops.unpack_w4(packed.cuda(), zp.cuda(), 6, 10).cpu().int(), (codes - zp.view(-1, 1).long()).int())


# ------------------------------------------------------------------ K4: int8 MFMA conv / linear
def _int_ref_dense(A, W):
    return A.double() @ W.double().t()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 192), (100, 768, 192), (64, 3, 128), (1000, 384, 1536),
                                    (257, 130, 320)])
def test_qgemm_dense_exact(ops, M, N, K):
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randint(-128, 128, (M, K), generator=gen, dtype=torch.int8)
    W = torch.randint(-8, 9, (N, K), generator=gen, dtype=torch.int8)
    scale = torch.rand(N, generator=gen) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen)
    rowadd = torch.randn((M + 47) // 48, N, generator=gen)
    out = torch.empty(M, N, device="cuda")
    ops.qgemm_i8(A.cuda(), W.cuda(), M, N, K, torch.ones(N).cuda(), torch.zeros(N).cuda(), out)
    exact(out.cpu().double(), _int_ref_dense(A, W))        # integer accumulation: bit-exact
    ops.qgemm_i8(A.cuda(), W.cuda(), M, N, K, scale.cuda(), bias.cuda(), out, rowadd=rowadd.cuda(),
                 rows_per_batch=48, residual=res.cuda())
    ref = _int_ref_dense(A, W) * scale.double() + bias.double() + rowadd.double()[torch.arange(M) // 48] + res.double()
    close(out, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,H,Cin,Cout,KH,stride,pad0,ups", [
    (2, 8, 64, 64, 3, 1, 1, False), (3, 16, 128, 192, 3, 1, 1, False), (2, 8, 64, 96, 3, 2, 0, False),
    (2, 8, 64, 64, 3, 2, 1, False), (2, 4, 128, 64, 3, 1, 1, True), (2, 8, 192, 40, 1, 1, 0, False),
    (2, 8, 32, 64, 3, 1, 1, False), (1, 8, 48, 32, 3, 2, 1, False), (2, 4, 16, 32, 3, 1, 1, True)])
def test_qgemm_conv_exact(ops, B, H, Cin, Cout, KH, stride, pad0, ups):
    gen = torch.Generator().manual_seed(B * H + Cin + Cout + stride)
    W_ = H
    x = torch.randint(-128, 128, (B, H, W_, Cin), generator=gen, dtype=torch.int8)      # NHWC operand
    w = torch.randint(-8, 9, (Cout, KH, KH, Cin), generator=gen, dtype=torch.int8)      # [co][ky][kx][ci]
    padval = -3
    xin = x.permute(0, 3, 1, 2).double()
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    Hl = xin.shape[2]
    if KH == 1:
        Ho = Hl
        xp = xin
    elif stride == 1:
        Ho = Hl
        xp = F.pad(xin, (1, 1, 1, 1), value=padval)
    elif pad0 == 0:       # DDPM Downsample: pad (0,1,0,1) then stride-2 conv (ddim/models/diffusion.py:66-70)
        Ho = Hl // 2
        xp = F.pad(xin, (0, 1, 0, 1), value=padval)
    else:                 # LDM Downsample: stride 2, padding 1 (openaimodel.py:150-152)
        Ho = Hl // 2
        xp = F.pad(xin, (1, 1, 1, 1), value=padval)
    ref = F.conv2d(xp, w.permute(0, 3, 1, 2).double(), stride=stride).permute(0, 2, 3, 1).reshape(-1, Cout)
    M, K = B * Ho * Ho, KH * KH * Cin
    assert ref.shape[0] == M
    geom = ops.make_geom(B, H, W_, Cin, Ho, Ho, KH, KH, stride, pad0, ups, padval)
    out = torch.empty(M, Cout, device="cuda")
    ops.qgemm_i8(x.cuda(), w.reshape(Cout, K).cuda(), M, Cout, K, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(),
                 out, geom=geom)
    exact(out.cpu().double(), ref)


def test_gemm_f16_batched_exact(ops):
    gen = torch.Generator().manual_seed(12)
    for (b, M, N, K) in ((3, 64, 64, 32), (2, 256, 256, 256), (4, 100, 77, 40), (2, 1024, 8, 64)):
        A = torch.randint(-128, 129, (b, M, K), generator=gen).half()
        Bm = torch.randint(-128, 129, (b, N, K), generator=gen).half()
        out = ops.gemm_f16_nt(A.cuda(), K, M * K, Bm.cuda(), K, N * K, b, M, N, K, 0.5)
        ref = torch.einsum("bmk,bnk->bmn", A.double(), Bm.double()) * 0.5
        exact(out.cpu().double(), ref)          # |sum| < 2^24: exact in fp32


def test_im2col_conv_in(ops):
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(2, 8, 8, 3, generator=gen)
    qp = _qp(ops, [(0.02, 127.0, 255)])
    col = ops.im2col_quant_i8(x.cuda(), 64, qp).cpu().int()
    code = (torch.clamp(torch.round(x / torch.tensor(0.02)) + 127, 0, 255) - 128)
    padded = F.pad(code.permute(0, 3, 1, 2), (1, 1, 1, 1), value=-1.0)       # zp-128 = -1 is real zero
    ref = F.unfold(padded, 3).reshape(2, 3, 9, 64).permute(0, 3, 2, 1).reshape(128, 27)   # [m][tap][c]
    exact(col[:, :27], ref.int())
    assert col[:, 27:].abs().max() == 0


def test_gemm_f16_heads_and_qgemm_f16(ops):
    gen = torch.Generator().manual_seed(14)
    B, N, h, d = 2, 96, 4, 40
    q = torch.randint(-128, 129, (B, N, h * d), generator=gen).half()
    k = torch.randint(-128, 129, (B, N, h * d), generator=gen).half()
    out = ops.gemm_f16_nt(q.cuda(), h * d, N * h * d, k.cuda(), h * d, N * h * d, B, N, N, d, 1.0, inner=h,
                          strideA_i=d, strideB_i=d)
    ref = torch.einsum("bnhd,bmhd->bhnm", q.reshape(B, N, h, d).double(), k.reshape(B, N, h, d).double())
    exact(out.cpu().double().reshape(B, h, N, N), ref)
    # f16 operand path of the quantised layer: weights up to +-128 (8-bit, zp 127)
    M, Nn, K = 100, 192, 64
    A = torch.randint(-128, 129, (M, K), generator=gen).half()
    W = torch.randint(-127, 129, (Nn, K), generator=gen).half()
    sc, bs = torch.rand(Nn, generator=gen) * 1e-3, torch.randn(Nn, generator=gen)
    o = torch.empty(M, Nn, device="cuda")
    ops.qgemm_f16(A.cuda(), W.cuda(), M, Nn, K, sc.cuda(), bs.cuda(), o)
    close(o, (A.double() @ W.double().t()) * sc.double() + bs.double(), rtol=1e-6, atol=1e-6)
    # fp32 last-layer convolution
    x = torch.randn(2, 8, 8, 64, generator=gen)
    w = torch.randn(3, 3, 3, 64, generator=gen) * 0.1
    b = torch.randn(3, generator=gen)
    got = ops.conv3x3_f32_smalln(x.cuda(), w.cuda(), b.cuda())
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    close(got, ref, rtol=1e-5, atol=1e-5)


def test_cat_consumers_match_materialised_concat(ops):
    """GroupNorm (stats + apply) and the activation quantiser over an unmaterialised [a | b] channel concatenation
    give the bits of the same kernels run on the concatenated tensor."""
    g = torch.Generator().manual_seed(11)
    for B, H, Ca, Cb, G in ((3, 8, 64, 32, 32), (2, 16, 384, 192, 32), (2, 4, 960, 960, 32)):
        a = torch.randn(B, H, H, Ca, generator=g).cuda()
        b = (torch.randn(B, H, H, Cb, generator=g) * 2 + 0.5).cuda()
        cat = ops.concat_c(a, b)
        C = Ca + Cb
        gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        qp = ops.qp_tensor([(0.03, 120.0, 255.0), (0.05, 131.0, 255.0)], "cuda")
        st0, st1 = ops.groupnorm_stats(cat, G, 1e-5), ops.groupnorm_stats(ops.Cat(a, b), G, 1e-5)
        exact(st0, st1)
        o0, q0 = ops.groupnorm_apply(cat, st0, gamma, beta, G, True, qp=qp, nq=2, want_f32=True)
        o1, q1 = ops.groupnorm_apply(ops.Cat(a, b), st1, gamma, beta, G, True, qp=qp, nq=2, want_f32=True)
        exact(o0, o1), exact(q0[0], q1[0]), exact(q0[1], q1[1])
        for split in (0, Ca):
            exact(ops.quant_i8(cat.reshape(-1, C), qp, split=split), ops.quant_i8(ops.Cat(a, b), qp, split=split))


def test_groupnorm_apply_wide_equals_narrow(ops):
    """k_gn_apply16 (one int8 operand, 16 channels per thread: the sampling form) gives the bits of k_gn_apply (taken when
    an fp32 output is also asked for): plain, concatenated, CFG-pair-periodic second half, raw operand with split."""
    g = torch.Generator().manual_seed(16)
    for B, H, Ca, Cb, rep in ((2, 8, 192, 0, 1), (3, 16, 384, 192, 1), (4, 8, 960, 960, 2), (2, 4, 64, 0, 1),
                              (2, 9, 576, 384, 1), (2, 8, 200, 0, 1)):
        a = (torch.randn(B, H, H, Ca, generator=g) * 1.5).cuda()
        x = a
        if Cb:
            b = (torch.randn(B // rep, H, H, Cb, generator=g) * 2 + 0.5).cuda()
            x = ops.Cat(a, b)
        C = Ca + Cb
        G = 32 if C % 32 == 0 else 8
        gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        qp = ops.qp_tensor([(0.03, 120.0, 255.0)], "cuda")
        rqp = ops.qp_tensor([(0.04, 128.0, 255.0), (0.07, 125.0, 255.0)], "cuda")
        st = ops.groupnorm_stats(x, G, 1e-5)
        for silu in (True, False):
            for split in ((0, Ca) if Cb else (0,)):
                _, qn, rn = ops.groupnorm_apply(x, st, gamma, beta, G, silu, qp=qp, nq=1, want_f32=True, raw_qp=rqp, raw_split=split)
                _, qw, rw = ops.groupnorm_apply(x, st, gamma, beta, G, silu, qp=qp, nq=1, raw_qp=rqp, raw_split=split)
                exact(qn[0], qw[0]), exact(rn, rw)
            if rep == 1:
                _, qn = ops.groupnorm_apply(x, st, gamma, beta, G, silu, qp=qp, nq=1, want_f32=True)
                _, qw = ops.groupnorm_apply(x, st, gamma, beta, G, silu, qp=qp, nq=1)
                exact(qn[0], qw[0])


def test_plms_loop_golden(ops, golden):
    """K9b + PLMSLoop (edadm/sampling.py) against the reference's PLMSSampler run (G14): every intermediate x and
    pred_x0 of the 8 steps, classifier-free guidance 7.5; fp32 elementwise chains: 3e-5."""
    from edadm.sampling import PLMSLoop
    g = golden("g14_plms")
    Wm = T(g["Wm"]).float().cuda()
    unet = lambda x_, t_, c_: torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
        + t_.float().view(-1, 1, 1, 1) / 1000.0
    loop = PLMSLoop(unet, (4, 8, 8), 3, steps=8, scale=float(g["scale"]), use_graph=False)
    np.testing.assert_array_equal(loop.ddim_timesteps, g["ts"])
    inter = {}
    out = loop.sample(D(g["x_T"]), cond=D(g["c"]), uncond=D(g["uc"]), intermediates=inter)
    close(torch.stack(inter["x_inter"]), g["x_inter"][1:], rtol=3e-5, atol=3e-5)
    close(torch.stack(inter["pred_x0"]), g["pred_x0"][1:], rtol=3e-5, atol=3e-5)
    close(out, g["final"], rtol=3e-5, atol=3e-5)
