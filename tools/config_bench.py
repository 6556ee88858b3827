"""Full-size runs of BASELINE.json's configs 2, 3 and 5 on one MI355X (the headline, config 4, is bench.py):

    cifar    CIFAR-10 DDIM 32x32 W4A8 (configs/cifar10.yml: ch 128, mult 1-2-2-2, attention at 16x16; 35.7 M parameters),
             batch 500 x 100 quad-skip DDIM steps, eta 0 (scripts/for_cifar.sh, sample_diffusion_ddim.py:265-323)
    church   LSUN-Church LDM-8 256x256 W4A8 (models/ldm/lsun_churches256/config.yaml: ch 192, mult 1-2-2-4-4, legacy 8-head
             attention at every level, scale-shift norm, resblock up/down; 295 M parameters), batch 100, DDIM eta 0
             (scripts/for_church.sh: 500 steps; --steps N runs N of them and reports per-step and extrapolated figures)
    sd       Stable Diffusion v1-4 512x512 W4A8 (v1-inference.yaml: ch 320, mult 1-2-4-4, 8 heads, 77 x 768 context; 860 M
             parameters), 4 prompts x CFG 7.5 (8 rows per call) x 50 PLMS steps (scripts/for_coco.sh, sample_txt2img.py:154-283)

Random-init weights (no checkpoint in the tree), scales from the build's own initialisation on a few synthetic rows (as
bench.py does), then the frozen int8 executor, UNet forward replayed from a HIP graph.  Prints one JSON line per config:
images/s, UNet-call ms, the int8 GEMM group's share of the dense int8 MFMA peak, layer modes.
    python tools/config_bench.py cifar church sd [--batches 2] [--steps N]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

WQ = dict(n_bits=4, symmetric=True, channel_wise=True, scale_method="mse")
AQ = dict(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True, prob=0.5)
I8_PEAK = 5033.0

CHURCH = dict(image_size=32, in_channels=4, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
              num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True, resblock_updown=True)
SD = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
          channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
          use_checkpoint=True, legacy=False)


def _reinit_zero(model, seed):
    g = torch.Generator().manual_seed(seed)
    for prm in model.parameters():
        if float(prm.detach().abs().max()) == 0.0:
            with torch.no_grad():
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)


def _quantise(model, dev, cali, kind, rows):
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
    from qdiff.set_quantize_params_LDM import all_act_quantizers
    qnn = QuantModel(model, WQ, AQ, sm_abit=8, act_quant_mode="qdiff").to(dev).eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    if kind == "cifar":
        qnn.model.config.split_shortcut = True
    else:
        qnn.set_grad_ckpt(False)
        if kind != "sd":                                   # sample_txt2img.py:183-184 sets an attribute nothing reads
            qnn.model.split_shortcut = True
    t0 = time.time()
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=max(rows // 2, 1))
    for q in all_act_quantizers(qnn):
        q.set_inited(True)
    torch.cuda.synchronize()
    qnn.set_quant_state(True, True)
    return qnn, time.time() - t0


def _gemm_group(eng, call):
    """executed flops / summed device time of every int8 GEMM launch of one engine call (each re-played 5x between events)"""
    eng.prof = []
    call()
    torch.cuda.synchronize()
    prof, eng.prof = eng.prof, None

    def ms(run, reps=5):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    rows = [(mode, name, M, N, K, f, ms(run)) for mode, name, M, N, K, f, run, by in prof]
    i8 = [r for r in rows if r[0] == "i8"]
    fl, t = sum(r[5] for r in i8), sum(r[6] for r in i8)
    other = sum(r[6] for r in rows if r[0] != "i8")
    return {"int8_gemm_calls": len(i8), "int8_gemm_gflop": fl / 1e9, "int8_gemm_ms": t, "int8_gemm_tflops": fl / t / 1e9 if t else None,
            "frac_of_int8_mfma_peak": fl / t / 1e9 / I8_PEAK if t else None, "other_contraction_ms": other,
            "slowest": [(n, M, N, K, round(m, 3)) for _, n, M, N, K, _, m in sorted(rows, key=lambda r: -r[6])[:6]]}


def _modes(eng):
    out = {}
    for L in eng.layers.values():
        out[L.mode] = out.get(L.mode, 0) + 1
    return out


def _ldm(kw, dev, seed):
    from edadm.nets.ldm_unet import UNetModel
    from qdiff.utils import seed_everything
    seed_everything(seed)
    m = UNetModel(**kw)
    _reinit_zero(m, seed)
    return m.to(dev).eval()


def cifar_model(dev):
    """configs/cifar10.yml:12-24 of the reference: DDPM UNet ch 128, mult 1-2-2-2, attention at 16 x 16"""
    from edadm.nets.ddpm_unet import Model
    from qdiff.utils import seed_everything
    seed_everything(1234)
    cfg = SimpleNamespace(model=SimpleNamespace(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2,
                                                attn_resolutions=[16], dropout=0.1, resamp_with_conv=True),
                          data=SimpleNamespace(image_size=32), diffusion=SimpleNamespace(num_diffusion_timesteps=1000))
    return Model(cfg).to(dev).eval()


CIFAR_SEQ = [int(s) for s in (np.linspace(0, np.sqrt(1000 * 0.8), 100) ** 2)]


def build(kind, dev):
    """(qnn with quick W4A8 scales, make_inputs(rows) -> (x, t, context or None), shipped rows per UNet call) for configs 2, 3, 5 at
    full size: random-init weights of the reference's architecture, scales from the build's own initialisation on a few synthetic rows."""
    if kind == "cifar":
        model = cifar_model(dev)
        g = torch.Generator().manual_seed(1)
        rows = 32
        cali = (torch.randn(rows, 3, 32, 32, generator=g).to(dev), torch.tensor(np.random.RandomState(0).choice(CIFAR_SEQ, rows)).float().to(dev))
        qnn, t_init = _quantise(model, dev, cali, "cifar", rows)

        def inputs(n, seed=5):
            gg = torch.Generator().manual_seed(seed)
            return (torch.randn(n, 3, 32, 32, generator=gg).to(dev),
                    torch.tensor(np.random.RandomState(seed).choice(CIFAR_SEQ, n)).float().to(dev), None)
        return qnn, inputs, 500, t_init
    if kind == "church":
        model = _ldm(CHURCH, dev, 1235)
        g = torch.Generator().manual_seed(2)
        rows = 16
        ts = np.arange(0, 1000, 2) + 1
        cali = (torch.randn(rows, 4, 32, 32, generator=g).to(dev),
                torch.tensor(ts[np.random.RandomState(0).randint(0, 500, rows)], dtype=torch.long, device=dev))
        qnn, t_init = _quantise(model, dev, cali, "church", rows)

        def inputs(n, seed=5):
            gg = torch.Generator().manual_seed(seed)
            return (torch.randn(n, 4, 32, 32, generator=gg).to(dev),
                    torch.tensor(ts[np.random.RandomState(seed).randint(0, 500, n)], dtype=torch.long, device=dev), None)
        return qnn, inputs, 100, t_init
    model = _ldm(SD, dev, 1236)
    g = torch.Generator().manual_seed(3)
    rows = 4
    ts = np.arange(0, 1000, 20) + 1
    cali = (torch.randn(rows, 4, 64, 64, generator=g).to(dev),
            torch.tensor(ts[np.random.RandomState(0).randint(0, 50, rows)], dtype=torch.long, device=dev),
            torch.randn(rows, 77, 768, generator=g).to(dev))
    qnn, t_init = _quantise(model, dev, cali, "sd", rows)

    def inputs(n, seed=5):
        gg = torch.Generator().manual_seed(seed)
        return (torch.randn(n, 4, 64, 64, generator=gg).to(dev),
                torch.tensor(ts[np.random.RandomState(seed).randint(0, 50, n)], dtype=torch.long, device=dev),
                torch.randn(n, 77, 768, generator=gg).to(dev))
    return qnn, inputs, 8, t_init


def quick_call_numbers(kind, dev):
    """bench.py's `configs` key: one UNet call of config `kind` at its shipped rows per call on the frozen int8 executor -- the eager
    call between HIP events (3 calls after a warm-up) and the int8 GEMM group's share of the dense int8 MFMA peak (cold-cache timing
    as in bench.py is not repeated here: warm replay, 3x per launch)."""
    with torch.no_grad():
        qnn, inputs, rows, t_init = build(kind, dev)
        eng = qnn.freeze()
        x, t, c = inputs(rows)
        eng(x, t, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng(x, t, c)
        e1.record()
        torch.cuda.synchronize()
        grp = _gemm_group(eng, lambda: eng(x, t, c))
    return {"rows_per_call": rows, "unet_call_ms_eager": e0.elapsed_time(e1) / 3, "scale_init_s": t_init, "layer_modes": _modes(eng),
            "int8_gemm_ms": grp["int8_gemm_ms"], "int8_gemm_calls": grp["int8_gemm_calls"],
            "frac_of_int8_mfma_peak": grp["frac_of_int8_mfma_peak"]}


def _timed_batches(mk, nfl, noise, nb, dev, cond=None, uncond=None):
    """seconds per batch over nb timed batches after nfl warm-up ones: serial (nfl = 1) or nfl batches in flight
    (edadm.sampling.InFlightSampler, as bench.py samples the headline configuration)"""
    if nfl > 1:
        from edadm.sampling import InFlightSampler
        fl = InFlightSampler(mk, n=nfl, device=dev)
        run = lambda x: fl.submit(x, cond, uncond)[0]
        fin = fl.drain
    else:
        loop = mk()
        run = lambda x: loop.sample(x, cond, uncond)
        fin = lambda: None
    for i in range(nfl):
        run(noise[i])
    fin()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(nb):
        out = run(noise[nfl + i])
    fin()
    torch.cuda.synchronize()
    return (time.time() - t0) / nb, out


def run_cifar(dev, a):
    from edadm.nets.ddpm_unet import Model
    from edadm.sampling import GraphedUNet
    from ddim.functions.denoising import generalized_steps
    from qdiff.utils import seed_everything
    seed_everything(1234)
    cfg = SimpleNamespace(model=SimpleNamespace(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2,
                                                attn_resolutions=[16], dropout=0.1, resamp_with_conv=True),
                          data=SimpleNamespace(image_size=32), diffusion=SimpleNamespace(num_diffusion_timesteps=1000))
    model = Model(cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    rows = 32
    seq = [int(s) for s in (np.linspace(0, np.sqrt(1000 * 0.8), 100) ** 2)]
    cali = (torch.randn(rows, 3, 32, 32, generator=g).to(dev), torch.tensor(np.random.RandomState(0).choice(seq, rows)).float().to(dev))
    qnn, t_init = _quantise(model, dev, cali, "cifar", rows)
    eng = qnn.freeze()
    B = a.batch or 500
    if a.calls:
        xg, tg = torch.randn(B, 3, 32, 32, device=dev), torch.full((B,), 500.0, device=dev)
        for _ in range(a.calls):
            eng(xg, tg, None)
        torch.cuda.synchronize()
        return {"config": "cifar", "calls": a.calls}
    betas = torch.linspace(1e-4, 2e-2, 1000).to(dev)
    x0 = torch.zeros(B, 3, 32, 32, device=dev)
    unet = GraphedUNet(eng, x0, torch.zeros(B, device=dev), None)
    steps = seq if not a.steps else seq[-a.steps:]

    def batch(seed):
        x = torch.randn(B, 3, 32, 32, generator=torch.Generator(device=dev).manual_seed(seed), device=dev)
        xs, _ = generalized_steps(x, steps, lambda xt, t: unet(xt, t), betas, eta=0.0)
        return xs[-1]

    batch(0)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(a.batches):
        out = batch(1 + i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.batches
    assert bool(torch.isfinite(out).all())
    xg, tg = torch.randn(B, 3, 32, 32, device=dev), torch.full((B,), 500.0, device=dev)
    grp = _gemm_group(eng, lambda: eng(xg, tg, None))
    per_step = dt / len(steps)
    return {"config": "2: CIFAR-10 DDIM 32x32 W4A8, batch %d x %d quad-skip steps" % (B, len(steps)), "images_per_sec": B / (per_step * 100),
            "unet_call_ms": 1e3 * per_step, "steps_run": len(steps), "batch": B, "scale_init_s": t_init, "layer_modes": _modes(eng),
            "gemm_group": grp, "algorithmic_tflops": B * 12.5e9 / per_step / 1e12}


def run_church(dev, a):
    from edadm.sampling import DDIMLoop
    model = _ldm(CHURCH, dev, 1235)
    g = torch.Generator().manual_seed(2)
    rows = 16
    ts = np.arange(0, 1000, 2) + 1
    cali = (torch.randn(rows, 4, 32, 32, generator=g).to(dev), torch.tensor(ts[np.random.RandomState(0).randint(0, 500, rows)], dtype=torch.long, device=dev))
    qnn, t_init = _quantise(model, dev, cali, "church", rows)
    eng = qnn.freeze()
    B = a.batch or 100
    S = a.steps or 20
    if a.calls:
        xg, tg = torch.randn(B, 4, 32, 32, device=dev), torch.full((B,), 501, dtype=torch.long, device=dev)
        for _ in range(a.calls):
            eng(xg, tg, None)
        torch.cuda.synchronize()
        return {"config": "church", "calls": a.calls}
    nfl = int(getattr(a, "inflight", 1))
    mk = lambda cs=None: DDIMLoop(eng, (4, 32, 32), B, steps=S, eta=0.0, scale=1.0, linear_start=0.0015, linear_end=0.0155, context_shape=None,
                                  device=dev, capture_stream=cs)
    nb = a.batches * nfl
    noise = [torch.randn(B, 4, 32, 32, generator=torch.Generator(device=dev).manual_seed(i), device=dev) for i in range(nb + nfl)]
    dt, out = _timed_batches(mk, nfl, noise, nb, dev)
    assert bool(torch.isfinite(out).all())
    xg, tg = torch.randn(B, 4, 32, 32, device=dev), torch.full((B,), 501, dtype=torch.long, device=dev)
    grp = _gemm_group(eng, lambda: eng(xg, tg, None))
    per_step = dt / S
    return {"config": "3: LSUN-Church LDM-8 256x256 W4A8, batch %d, DDIM eta 0 (%d of the shipped 500 steps run)" % (B, S),
            "images_per_sec_at_500_steps": B / (per_step * 500), "images_per_sec_at_%d_steps" % S: B / dt, "unet_call_ms": 1e3 * per_step,
            "batch": B, "scale_init_s": t_init, "layer_modes": _modes(eng), "gemm_group": grp,
            "algorithmic_tflops": B * 38e9 / per_step / 1e12}


def run_sd(dev, a):
    from edadm.sampling import PLMSLoop
    model = _ldm(SD, dev, 1236)
    g = torch.Generator().manual_seed(3)
    rows = 4
    ts = np.arange(0, 1000, 20) + 1
    cali = (torch.randn(rows, 4, 64, 64, generator=g).to(dev), torch.tensor(ts[np.random.RandomState(0).randint(0, 50, rows)], dtype=torch.long, device=dev),
            torch.randn(rows, 77, 768, generator=g).to(dev))
    qnn, t_init = _quantise(model, dev, cali, "sd", rows)
    eng = qnn.freeze()
    B = a.batch or 4
    S = a.steps or 50
    if a.calls:
        xg, tg = torch.randn(2 * B, 4, 64, 64, device=dev), torch.full((2 * B,), 501, dtype=torch.long, device=dev)
        cg = torch.randn(2 * B, 77, 768, device=dev)
        for _ in range(a.calls):
            eng(xg, tg, cg)
        torch.cuda.synchronize()
        return {"config": "sd", "calls": a.calls}
    nfl = int(getattr(a, "inflight", 1))
    mk = lambda cs=None: PLMSLoop(eng, (4, 64, 64), B, steps=S, scale=7.5, context_shape=(77, 768), device=dev, capture_stream=cs)
    gen = torch.Generator(device=dev).manual_seed(9)
    cond = torch.randn(B, 77, 768, generator=gen, device=dev)
    uncond = torch.randn(1, 77, 768, generator=gen, device=dev).expand(B, 77, 768).contiguous()
    nb = a.batches * nfl
    noise = [torch.randn(B, 4, 64, 64, generator=gen, device=dev) for _ in range(nb + nfl)]
    dt, out = _timed_batches(mk, nfl, noise, nb, dev, cond, uncond)
    assert bool(torch.isfinite(out).all())
    xg = torch.randn(2 * B, 4, 64, 64, device=dev)
    tg = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
    cg = torch.cat([uncond, cond])
    grp = _gemm_group(eng, lambda: eng(xg, tg, cg))
    calls = S + 1
    return {"config": "5: Stable Diffusion v1-4 512x512 W4A8, %d prompts x CFG 7.5 (%d rows/call) x %d PLMS steps (%d UNet calls)" % (B, 2 * B, S, calls),
            "images_per_sec": B / dt, "seconds_per_batch": dt, "unet_call_ms": 1e3 * dt / calls, "batch": B, "scale_init_s": t_init,
            "layer_modes": _modes(eng), "gemm_group": grp, "algorithmic_tflops": 2 * B * 2 * 338.6e9 * calls / dt / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", choices=["cifar", "church", "sd"])
    ap.add_argument("--batches", type=int, default=2)
    ap.add_argument("--steps", type=int, default=0, help="sampling steps to run (0: cifar 100, church 20 of 500, sd 50)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=2, help="sample batches in flight (church, sd)")
    ap.add_argument("--calls", type=int, default=0, help="profiling mode (rocprofv3 + tools/prof_diff.py): set up, then only N eager UNet calls")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from edadm import lib
    lib.load()
    for c in a.configs:
        t0 = time.time()
        try:
            with torch.no_grad():
                r = {"cifar": run_cifar, "church": run_church, "sd": run_sd}[c](dev, a)
        except Exception as e:
            import traceback
            r = {"config": c, "error": repr(e), "trace": traceback.format_exc()[-1500:]}
        r["wall_s"] = time.time() - t0
        r["peak_hbm_gb"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
