"""-m gpu: the reference's script flow end to end on tiny networks through the product API
(scripts/sample_diffusion_ddim.py:265-323 and sample_diffusion_ldm_imagenet.py:142-249):
TDAC calibration set -> scale init -> block reconstruction walk -> freeze -> quantised DDIM sampling."""
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import build_cifar, build_ldm, WQ4, AQ8

pytestmark = pytest.mark.gpu


def _recon_kwargs(cali, bs):
    return dict(cali_data=cali, iters=3, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0,
                weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=bs, input_prob=0.5, add_loss=0.8,
                recon_w=True, recon_a=True, keep_gpu=False)


def test_cifar_flow(golden):
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params, recon_block_Qmodel
    from qdiff.utils import seed_everything
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from scripts.calibration import TDAC_cifar_calib_data_generator
    from ddim.functions.denoising import generalized_steps
    g = golden("g13_cifar_unet")
    seed_everything(1234)
    model = build_cifar(g).cuda()
    qnn = QuantModel(model, WQ4, AQ8, sm_abit=8).cuda().eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    betas = torch.linspace(1e-4, 0.02, 1000).cuda()
    seq = [int(s) for s in (np.linspace(0, np.sqrt(1000 * 0.8), 10) ** 2)]
    diffusion = SimpleNamespace(seq=seq, betas=betas, args=SimpleNamespace(eta=0.0))
    cali = TDAC_cifar_calib_data_generator(qnn.model, qnn.model.config, 1.2, 64, 32, torch.device("cuda"), diffusion,
                                           class_cond=False)
    assert cali[0].shape == (64, 3, 16, 16) and cali[1].shape == (64,)
    assert set(int(v) for v in cali[1].cpu()) <= set(seq)
    qnn.model.config.split_shortcut = True
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    qnn = recon_block_Qmodel(None, qnn, cali, _recon_kwargs(cali, 32)).recon()
    assert qnn.block_count == len(g["units"])
    n_ada = sum(isinstance(m, AdaRoundQuantizer) and not m.soft_targets for m in qnn.modules())
    assert n_ada >= len(g["units"])
    qnn.set_quant_state(True, True)
    x = torch.randn(4, 3, 16, 16, device="cuda")
    with torch.no_grad():
        ref = qnn(x, torch.full((4,), 500.0, device="cuda"))        # fake-quant graph
    qnn.freeze()
    with torch.no_grad():
        out = qnn(x, torch.full((4,), 500.0, device="cuda"))        # int8 engine
    err = (out - ref).abs()
    print('cifar flow: engine vs fake-quant graph max %.3f mean %.4f of range %.3f' % (err.max(), err.mean(), ref.abs().max()))
    assert err.max() < 0.2 * ref.abs().max() and err.mean() < 0.02 * ref.abs().max()
    xs, x0 = generalized_steps(x, seq, qnn, betas, eta=0.0)
    assert len(xs) == len(seq) + 1 and torch.isfinite(xs[-1]).all()


def test_ldm_conditional_flow(golden):
    from qdiff import QuantModel
    from qdiff.utils import seed_everything
    from qdiff_control import (set_weight_quantize_params_Conditional, set_act_quantize_params_Conditional,
                               recon_block_Qmodel)
    from edadm.latent import LatentDiffusionLite, ClassEmbedder
    from ldm.models.diffusion.ddim_control import DDIMSampler_control
    from scripts.calibration import TDAC_imagenet_calib_data_generator
    from edadm.state import quant_state_dict, load_quant_state
    g = golden("g13_ldm_imagenet")
    seed_everything(1234)
    unet = build_ldm(g)
    ld = LatentDiffusionLite(unet, linear_start=0.0015, linear_end=0.0195, conditioning_key="crossattn",
                             cond_stage_model=ClassEmbedder(16, n_classes=1001)).cuda().eval()
    qnn = QuantModel(ld.model.diffusion_model, WQ4, AQ8, act_quant_mode="qdiff", sm_abit=8).cuda().eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)
    ld.model.diffusion_model = qnn
    args = SimpleNamespace(scale=3.0, custom_steps=10, ddim_eta=0.0, lamda=1.2, latent_shape=[3, 8, 8],
                           data=torch.randint(0, 1000, (64,)).cuda())
    cali = TDAC_imagenet_calib_data_generator(ld, args, 64, 32, torch.device("cuda"), 10)
    assert cali[0].shape == (64, 3, 8, 8) and cali[3].shape == (64, 1, 16) and cali[4].shape == (64, 1, 16)
    qnn.model.split_shortcut = True
    set_weight_quantize_params_Conditional(ld, cali, args)
    set_act_quantize_params_Conditional(ld, cali, args)
    qnn.set_quant_state(True, True)
    rq = recon_block_Qmodel(args, qnn, cali, _recon_kwargs(cali, 32)).recon()
    assert rq is qnn and qnn.block_count == len(g["units"])
    state = quant_state_dict(qnn)
    assert any(k.endswith("/alpha") for k in state) and any(k.endswith("/split") for k in state)
    qnn.set_quant_state(True, True)
    qnn.freeze()
    sampler = DDIMSampler_control(ld)
    c = ld.get_learned_conditioning({"class_label": args.data[:4]})
    uc = ld.get_learned_conditioning({"class_label": torch.full((4,), 1000).cuda()})
    samples, _ = sampler.sample(S=10, conditioning=c, batch_size=4, shape=[3, 8, 8], verbose=False,
                                unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0)
    assert samples.shape == (4, 3, 8, 8) and torch.isfinite(samples).all()
    # the compiled sampling loop (HIP-graph replay) gives the same latents as the sampler class
    from edadm.sampling import DDIMLoop
    x_T = torch.randn(4, 3, 8, 8, device="cuda")
    a, _ = sampler.sample(S=10, conditioning=c, batch_size=4, shape=[3, 8, 8], verbose=False, x_T=x_T,
                          unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0)
    loop = DDIMLoop(qnn.engine, (3, 8, 8), 4, steps=10, scale=3.0, context_shape=(1, 16))
    b = loop.sample(x_T, c, uc)
    assert (a - b).abs().max() < 1e-4 * max(1.0, float(a.abs().max()))
    # several batches in flight (edadm.sampling.InFlightSampler: one loop per stream over the same engine, graphs captured on
    # separate streams): every batch has the bits of the serial loop, through the sharded driver too
    from edadm.sampling import InFlightSampler
    from edadm.sample_driver import ShardedSampler
    fl = InFlightSampler(lambda cs: DDIMLoop(qnn.engine, (3, 8, 8), 4, steps=10, scale=3.0, context_shape=(1, 16), capture_stream=cs), n=2)
    noises = [torch.randn(4, 3, 8, 8, device="cuda") for _ in range(5)]
    serial = [loop.sample(n_, c, uc).clone() for n_ in noises]
    got = [fl.submit(n_, c, uc) for n_ in noises]
    fl.drain()
    torch.cuda.synchronize()
    for s_, (l_, _) in zip(serial, got):
        assert torch.equal(s_, l_)
    out_serial, out_flight = {}, {}
    ShardedSampler(loop, 7, 20, 4, (3, 8, 8), n_classes=1000).run(lambda i, lab: (c, uc), lambda i, lat: out_serial.__setitem__(i, lat.clone()))
    ShardedSampler(fl, 7, 20, 4, (3, 8, 8), n_classes=1000).run(lambda i, lab: (c, uc), lambda i, lat: out_flight.__setitem__(i, lat.clone()))
    torch.cuda.synchronize()
    assert sorted(out_serial) == sorted(out_flight) == list(range(5))
    for i in out_serial:
        assert torch.equal(out_serial[i], out_flight[i]), i


def test_frozen_state_round_trip(golden, tmp_path):
    """calibration and sampling as separate jobs.  The frozen integer model (packed 4-bit weights,
    scales, folded biases, quantiser tables) written by one process and loaded into an engine built over DIFFERENT
    floating-point weights gives the bits of the original engine; 4-bit layers cost half a byte per weight."""
    import numpy as np
    from helpers import build_ldm, quantize_like_reference
    from edadm.state import save_frozen, load_frozen
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, ctx), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        want = qnn.freeze()(x, t, ctx)
        path = str(tmp_path / "frozen.npz")
        nbytes = save_frozen(qnn, path)
        st = np.load(path)
        n4 = sum(st[k].size * 2 for k in st.files if k.endswith("/w4"))
        nw = n4 + sum(st[k].size for k in st.files if k.endswith("/w") or k.endswith("/w_f32"))
        assert n4 > 0.9 * nw, "the W4 layers should be stored as nibbles"
        print("frozen state: %d bytes for %d weights (%.2f B/weight)" % (nbytes, nw, nbytes / nw))
        # a second model: same topology and quantiser state, other floating-point weights
        other = build_ldm(g)
        for p_ in other.parameters():
            p_.data.add_(0.05 * torch.randn_like(p_))
        qnn2, _, _ = quantize_like_reference(other, g, "ldm")
        qnn2.set_quant_state(True, True)
        eng2 = qnn2.freeze()
        assert not torch.equal(eng2(x, t, ctx), want)
        load_frozen(qnn2, path)
        assert torch.equal(qnn2.engine(x, t, ctx), want)


def test_plms_sampler_class_golden(golden):
    """ldm.models.diffusion.plms.PLMSSampler (reference interface) reproduces the reference's own 8-step PLMS run
    (G14): all x_inter / pred_x0, ts / ts_next bookkeeping, and the quant_unet branch returns the guided prediction."""
    from ldm.models.diffusion.plms import PLMSSampler
    g = golden("g14_plms")
    dev = torch.device("cuda")
    Wm = torch.tensor(g["Wm"]).float().to(dev)
    ac = np.cumprod(1.0 - g["betas"], axis=0)

    class FakeLD:
        num_timesteps = 1000
        betas = torch.tensor(g["betas"], dtype=torch.float32, device=dev)
        alphas_cumprod = torch.tensor(ac, dtype=torch.float32, device=dev)
        alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32, device=dev)
        device = dev

        def apply_model(self, x_, t_, c_):
            return torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
                + t_.float().view(-1, 1, 1, 1) / 1000.0

    s = PLMSSampler(FakeLD())
    x, c, uc = (torch.tensor(g[k]).float().to(dev) for k in ("x_T", "c", "uc"))
    out, inter = s.sample(S=8, batch_size=3, shape=(4, 8, 8), conditioning=c, x_T=x.clone(), verbose=False,
                          unconditional_guidance_scale=float(g["scale"]), unconditional_conditioning=uc)
    tol = dict(rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(torch.stack(inter["x_inter"]).cpu().numpy(), g["x_inter"], **tol)
    np.testing.assert_allclose(torch.stack(inter["pred_x0"][1:]).cpu().numpy(), g["pred_x0"][1:], **tol)
    np.testing.assert_allclose(out.cpu().numpy(), g["final"], **tol)
    assert [int(t[0]) for t in inter["ts"]] == list(np.flip(g["ts"]))
    assert [int(t[0]) for t in inter["ts_next"]] == list(np.flip(g["ts"]))[1:] + [int(g["ts"][0])]
    assert [len(o) for o in inter["old_eps"]] == [0, 1, 2, 3, 3, 3, 3, 3]
    t = torch.full((3,), int(g["ts"][3]), device=dev)
    e = s.sample(S=8, batch_size=3, shape=(4, 8, 8), quant_unet=True, cali_data=(x, t, None, c, uc),
                 unconditional_guidance_scale=float(g["scale"]))
    fl = FakeLD()
    want = fl.apply_model(x, t, uc) + float(g["scale"]) * (fl.apply_model(x, t, c) - fl.apply_model(x, t, uc))
    np.testing.assert_allclose(e.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("which", ["bedroom", "church", "coco_plms", "coco_ddim"])
def test_tdac_generators_ldm(golden, which):
    """the remaining TDAC generators (scripts/calibration.py:156,263,502) on a fixture-sized LDM:
    tuple layout and shapes, finite samples, `t` is the DDIM timestep of `index` for every row, `t_next` (coco) the
    following one, conditioning / unconditional embeddings carried through (the allocation itself is pinned by G9)."""
    from edadm.latent import LatentDiffusionLite
    from qdiff.utils import seed_everything
    import scripts.calibration as cal
    g = golden("g13_ldm_%s" % ("church" if which in ("bedroom", "church") else "imagenet"))
    unet = build_ldm(g)
    coco = which.startswith("coco")

    class Prompts(torch.nn.Module):                     # stand-in text encoder: prompt -> [1, 16] embedding
        def forward(self, prompts):
            return torch.stack([torch.full((1, 16), float(len(p) % 7) * 0.1) for p in prompts]).cuda()

    ld = LatentDiffusionLite(unet, linear_start=0.0015, linear_end=0.0155,
                             conditioning_key="crossattn" if coco else None,
                             cond_stage_model=Prompts() if coco else None).cuda().eval()
    if coco:
        ld.get_learned_conditioning = lambda prompts: Prompts()(prompts)
    C, S = unet.in_channels, 10
    unet.image_size = 8
    N, nb = 32, 16
    if coco:
        args = SimpleNamespace(scale=7.5, custom_steps=S, ddim_eta=0.0, lamda=1.2, C=C, H=64, W=64, f=8, plms=(which == "coco_plms"),
                               list_prompts=["a photo %d" % (i * i) for i in range(N)])
        fn = cal.TDAC_coco_calib_data_generator
    else:
        args = SimpleNamespace(custom_steps=S, eta=0.0, lamda=1.2)
        fn = cal.TDAC_bedroom_calib_data_generator if which == "bedroom" else cal.TDAC_church_calib_data_generator
    seed_everything(7)
    out = fn(ld, args, N, nb, torch.device("cuda"), S)
    assert len(out) == (6 if coco else 3)
    x, t, index = out[0], out[1], out[2]
    assert x.shape == (N, C, 8, 8) and t.shape == (N,) and index.shape == (N,)
    assert torch.isfinite(x).all() and int(index.min()) >= 0 and int(index.max()) <= S - 1
    from edadm.schedule import make_ddim_timesteps
    steps = make_ddim_timesteps("uniform", S, 1000, verbose=False)
    assert all(int(tt) == int(steps[int(i)]) for tt, i in zip(t, index))          # t is the timestep of ddim index
    if coco:
        assert out[3].shape == (N, 1, 16) and out[4].shape == (N, 1, 16) and out[5].shape == (N,)
        nxt = [int(steps[max(int(i) - 1, 0)]) for i in index]
        assert [int(v) for v in out[5]] == nxt


def test_fp_trace_matches_per_unit_passes(golden, monkeypatch):
    """The look-ahead FP activation cache (one FP prefix pass serving several units) returns the tensors of the
    reference's per-unit double pass (qdiff/data_utils.py:112-150), bit for bit, whatever the byte budget."""
    from qdiff import QuantModel
    from qdiff.utils import seed_everything
    from qdiff_control import set_weight_quantize_params_Conditional, set_act_quantize_params_Conditional
    from qdiff_control.data_utils import save_inp_oup_data
    import qdiff.data_utils as du
    from edadm.latent import LatentDiffusionLite, ClassEmbedder
    g = golden("g13_ldm_imagenet")
    seed_everything(7)
    unet = build_ldm(g)
    ld = LatentDiffusionLite(unet, linear_start=0.0015, linear_end=0.0195, conditioning_key="crossattn",
                             cond_stage_model=ClassEmbedder(16, n_classes=1001)).cuda().eval()
    qnn = QuantModel(ld.model.diffusion_model, WQ4, AQ8, act_quant_mode="qdiff", sm_abit=8).cuda().eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)
    ld.model.diffusion_model = qnn
    n = 64
    cali = (torch.randn(n, 3, 8, 8).cuda(), torch.randint(1, 1000, (n,)).cuda(), torch.zeros(n, dtype=torch.long).cuda(),
            torch.randn(n, 1, 16).cuda(), torch.randn(n, 1, 16).cuda())
    args = SimpleNamespace(scale=3.0, custom_steps=10, ddim_eta=0.0, lamda=1.2, latent_shape=[3, 8, 8])
    qnn.model.split_shortcut = True
    set_weight_quantize_params_Conditional(ld, cali, args)
    set_act_quantize_params_Conditional(ld, cali, args)
    units = du.recon_units(qnn)
    assert len(units) == len(g["units"])

    def flat(r):
        out = []
        def walk(v):
            if torch.is_tensor(v):
                out.append(v)
            elif isinstance(v, (list, tuple)):
                for e in v:
                    walk(e)
        walk(r[1:])
        return r[0], out

    from qdiff.quant_layer import QuantModule

    def walk_all(trace_gb, memo_gb):
        """save_inp_oup_data for every unit in walk order; after each unit its weight scales are perturbed, standing
        in for the reconstruction that changes what the units after it see."""
        monkeypatch.setattr(du, "FP_TRACE_GB", float(trace_gb))
        monkeypatch.setattr(du, "Q_MEMO_GB", float(memo_gb))
        du.clear_fp_trace(qnn)
        du.STATS.update(fp_passes=0, fp_captures=0, units_served=0, memo_hits=0)
        undo, res = [], []
        try:
            for u in units:
                res.append(flat(save_inp_oup_data(qnn, u, cali, True, True, batch_size=32, input_prob=True)))
                for m in u.modules():
                    if isinstance(m, QuantModule):
                        d = m.weight_quantizer.delta
                        undo.append((d, d.detach().clone()))
                        d.data.mul_(1.07)
        finally:
            for d, keep in undo:
                d.data.copy_(keep)
        stats = dict(du.STATS)
        left = bool(qnn._fp_trace.store) if getattr(qnn, "_fp_trace", None) is not None else False
        du.clear_fp_trace(qnn)
        return res, stats, left

    ref, st0, _ = walk_all("0", "0")
    assert st0["fp_captures"] == 0 and st0["memo_hits"] == 0
    for trace_gb, memo_gb in (("48", "0"), ("0.0002", "0"), ("48", "64"), ("0.0002", "0.004")):
        got, st, left = walk_all(trace_gb, memo_gb)
        for (rb, tens), (rb2, tens2) in zip(ref, got):
            assert rb2 == rb and len(tens2) == len(tens)
            for a, b in zip(tens, tens2):
                assert a.shape == b.shape and torch.equal(a, b)
        assert st["units_served"] == len(units) and not left
        if trace_gb == "48":
            assert st["fp_captures"] == 1                                   # one FP sweep serves the whole walk
        else:
            assert 1 < st["fp_captures"] < len(units)                       # several look-ahead groups
        assert (st["memo_hits"] > 0) == (memo_gb != "0")
        if memo_gb == "64":
            full_hits = st["memo_hits"]
        elif memo_gb != "0":
            assert st["memo_hits"] < full_hits                              # a partial memo: some units recompute
        print("trace", trace_gb, "memo", memo_gb, st)
