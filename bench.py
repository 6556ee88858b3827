#!/usr/bin/env python3
"""bench.py -- images/sec of W4A8 LDM-4 ImageNet 256x256 on MI355X, and the full calibration wall-clock (BASELINE.json metric).

A "step" = one batch of 50 images taken through the path the reference's sampling loop takes per batch
(sample_diffusion_ldm_imagenet.py:215-249): 20 DDIM steps x classifier-free guidance 3.0 (100 UNet rows per call) on the frozen
int8 executor, UNet forward replayed from a HIP graph, THEN the VQ-f4 first-stage decode of the batch to 256x256 pixels (on a
second HIP stream, overlapping the next batch's sampling).  `value` = decoded images per second; `sampling_only` (same line) is the
quantised-UNet part alone.  Inputs (noise latents, class-embedding context) are resident in HBM before the timed region; weights
are random-init LDM-4 (cin256-v2 shapes, 400.9 M params) -- no checkpoint / dataset exists in the tree.

At N = 1 the run then measures the OTHER half of the metric in the same process: the whole calibration job at the shipped size --
TDAC calibration set (1024 samples) -> scale initialisation over all 1024 -> 1000 iterations x 80 units of reconstruction --
stage by stage (`calibration.stages`), with the reconstruction loop's roofline (`calibration.h1_roofline`).  ~9 minutes;
`--calib bounded` / `--calib none` shorten the run.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

I8_PEAK_TFLOPS = 5033.0      # gfx950 dense int8 MFMA: 2048 op/clk/SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md)
UNET_GFLOP_PER_ROW = 208.4   # conv/linear/attention matmul FLOPs per sample-forward

LDM4 = dict(image_size=64, in_channels=3, out_channels=3, model_channels=192, attention_resolutions=[8, 4, 2],
            num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_heads=1, use_spatial_transformer=True,
            transformer_depth=1, context_dim=512)
WQ = dict(n_bits=4, symmetric=True, channel_wise=True, scale_method="mse")
AQ = dict(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True, prob=0.5)


def build_quantised_unet(device, calib_rows=16, seed=1234):
    from edadm.nets.ldm_unet import UNetModel
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
    from qdiff.utils import seed_everything
    seed_everything(seed)
    model = UNetModel(**LDM4)
    g = torch.Generator().manual_seed(seed)
    for prm in model.parameters():                      # zero_module convs: give them weights
        if float(prm.detach().abs().max()) == 0.0:
            with torch.no_grad():
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    model = model.to(device).eval()
    sd_cpu = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    qnn = QuantModel(model, WQ, AQ, sm_abit=8, act_quant_mode="qdiff").to(device).eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)
    qnn.model.split_shortcut = True
    ts = np.arange(0, 1000, 50) + 1
    x = torch.randn(calib_rows, 3, 64, 64, generator=g).to(device)
    t = torch.tensor(ts[torch.randint(0, 20, (calib_rows,), generator=g).numpy()], dtype=torch.long, device=device)
    c = torch.randn(calib_rows, 1, 512, generator=g).to(device)
    t0 = time.time()
    set_weight_quantize_params(qnn, (x, t, c))
    torch.cuda.synchronize()
    t1 = time.time()
    from qdiff.set_quantize_params_LDM import all_act_quantizers
    set_act_quantize_params(qnn, (x, t, c), batch_size=calib_rows // 2)
    for q in all_act_quantizers(qnn):
        q.set_inited(True)
    torch.cuda.synchronize()
    t2 = time.time()
    qnn.set_quant_state(True, True)
    return qnn, sd_cpu, dict(weight_init_s=t1 - t0, act_init_s=t2 - t1, calib_rows=calib_rows)


VQF4 = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
            resolution=256, z_channels=3)        # models/first_stage_models/vq-f4/config.yaml (embed_dim 3, 8192 codes)


def time_decoder(eng, dev, B):
    """the VQ-f4 first-stage decoder on its own (55.3 M parameters, 318 GMAC per image, fp32) on the HIP fp32-grade
    kernels, B latents -> B images of 256x256."""
    z = torch.randn(B, 3, 64, 64, device=dev)
    eng(z)                                   # untimed pass at the full batch: allocator pools and code objects warm
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        img = eng(z)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 2
    assert img.shape == (B, 3, 256, 256) and bool(torch.isfinite(img).all())
    return {"images": B, "wall_s": dt, "ms_per_image": 1e3 * dt / B, "tflops_fp32": B * 2 * 318.1e9 / dt / 1e12,
            "config": "VQ-f4 decoder, fp32 in / fp32 out; large convolutions as three f16-MFMA products over two-term f16 "
                      "expansions (fp32-grade, DESIGN.md section 4), the rest on the exact-fp32 MFMA; random-init weights"}


def make_decoder(dev):
    from edadm.nets.vq_decoder import Decoder
    from edadm.decoder import DecoderEngine
    torch.manual_seed(4321)
    return DecoderEngine(Decoder(**VQF4).to(dev).eval(), torch.nn.Conv2d(3, 3, 1).to(dev), codebook=torch.randn(8192, 3, device=dev))


def run_steps(loop, dec, side, noise, cond, uncond, first, last, dev, flight=None):
    """Steps [first, last): sample batch k, decode it on `side` while later batches sample (the loop of
    sample_diffusion_ldm_imagenet.py:215-249 delivers decoded images).  flight: an edadm.sampling.InFlightSampler -- batch k is
    sampled on stream k mod n, n batches in flight; else on the current stream.  dec None: sampling only.  Returns images delivered."""
    prev, imgs = None, 0
    cur = torch.cuda.current_stream(dev)

    def decode(p):
        lat, src = p
        side.wait_stream(src)
        with torch.cuda.stream(side):
            lat.record_stream(side)
            return dec(lat).shape[0]

    for i in range(first, last):
        if prev is not None:
            imgs += decode(prev)
            prev = None
        if flight is not None:
            lat, src = flight.submit(noise[i % len(noise)], cond, uncond)
        else:
            lat, src = loop.sample(noise[i % len(noise)], cond, uncond), cur
        if dec is None:
            imgs += lat.shape[0]
        else:
            prev = (lat, src)
    if prev is not None:
        imgs += decode(prev)
    if flight is not None:
        flight.drain()
    cur.wait_stream(side)
    return imgs


def instrumented_walk(qnn, cali, kwargs, n_calib):
    """The conditional reconstruction walk (qdiff_control.recon_block_Qmodel, every unit of the model) with its phases timed:
    activation caching (save_inp_oup_data), per-sample FP feature maps, steady-state iterations, per-unit setup; the cache budgets
    scaled with the calibration-set size so that a bounded run groups and memoises exactly as the full-size run does."""
    import qdiff_control.block_recon as cb
    import qdiff_control.layer_recon as cl
    from qdiff_control import recon_block_Qmodel
    import qdiff.data_utils as du
    import edadm.recon as er
    iters = kwargs["iters"]
    t_cache = [0.0]
    orig = (cb.save_inp_oup_data, cl.save_inp_oup_data)
    full_gb, memo_gb, feat_gb = du.FP_TRACE_GB, du.Q_MEMO_GB, er.FP_FEAT_GB
    du.FP_TRACE_GB = full_gb * n_calib / 1024
    du.Q_MEMO_GB = memo_gb * n_calib / 1024
    er.FP_FEAT_GB = feat_gb * n_calib / 1024
    er.FP_FEAT_FORCE = iters < 1000          # pays off over the 1000 iterations of the real run: forced on in a bounded one
    du.STATS.update(fp_passes=0, fp_captures=0, units_served=0, memo_hits=0)

    def timed_save(*a, **k):
        torch.cuda.synchronize()
        t0 = time.time()
        r = orig[0](*a, **k)
        torch.cuda.synchronize()
        t_cache[0] += time.time() - t0
        return r

    cb.save_inp_oup_data = cl.save_inp_oup_data = timed_save
    er.TIMING = {"iter_s": 0.0, "iters": 0}
    try:
        qnn.set_quant_state(True, True)
        torch.cuda.synchronize()
        t0 = time.time()
        recon_block_Qmodel(None, qnn, cali, kwargs).recon()
        torch.cuda.synchronize()
        total = time.time() - t0
    finally:
        cb.save_inp_oup_data, cl.save_inp_oup_data = orig
        timing, er.TIMING = er.TIMING, None
        du.FP_TRACE_GB, du.Q_MEMO_GB, er.FP_FEAT_GB, er.FP_FEAT_FORCE = full_gb, memo_gb, feat_gb, False
    loop = total - t_cache[0]
    units = qnn.block_count
    # steady-state seconds of ONE iteration of every unit (iterations after the first of each unit, edadm/recon.py);
    # what is left of the loop time is per-unit setup (AdaRound init, optimiser state, first iterations, graph capture)
    per_iter_all_units = timing["iter_s"] / max(timing["iters"], 1) * units
    feat = timing.get("feat_s", 0.0)
    setup = max(loop - timing["iter_s"] - feat, 0.0)
    flops = timing.get("flops", 0.0)
    tf = flops / max(timing["iter_s"], 1e-9) / 1e12
    return dict(units=units, calib_samples=n_calib, iters_per_unit=iters, wall_s=total, caching_s=t_cache[0],
                loop_s=loop, steady_iterations_s=timing["iter_s"], unit_setup_s=setup, fp_features_s=feat,
                s_per_iteration_all_units=per_iter_all_units,
                per_unit_ms=[{"unit": u, "weights": n, "ms_per_iteration": ms, "positions_per_row": pos, "data_parallel_eligible": pos >= er.DP_MIN_POSITIONS}
                             for (u, n, ms), pos in zip(timing.get("per_unit", []), timing.get("per_unit_positions", []))],
                # steady-state loop seconds of the units whose iterations split over the ranks of a multi-rank job (edadm.recon.DP_LOOP:
                # >= DP_MIN_POSITIONS feature-map positions / tokens per row, the 64 x 64 and 32 x 32 levels) -- measured per unit
                dp_eligible_loop_s=sum(ms * 1e-3 * (iters - 1) for (u, n, ms), pos in zip(timing.get("per_unit", []), timing.get("per_unit_positions", []))
                                       if pos >= er.DP_MIN_POSITIONS),
                dp_stats=dict(er.DP_STATS),
                graphed_units=timing.get("graphed_units", 0),
                h1_roofline={"bound": "mfma", "what": "contractions of the steady-state reconstruction iterations (forward x2-3, input and "
                             "weight gradients; attention products included), executed fp32-equivalent flops counted on the host as they "
                             "are issued / wall time of those iterations (everything in them: elementwise, Adam, gathers)",
                             "executed_fp32_equivalent_pflop": flops / 1e15, "seconds": timing["iter_s"], "achieved": tf, "unit": "TFLOP/s",
                             "peak_f16_three_product": 2516.0 / 3, "frac": tf / (2516.0 / 3),
                             "peak_fp32_mfma": 157.0, "frac_of_fp32_mfma": tf / 157.0},
                fp_features={"s": feat, "units_cached": timing.get("feat_units", 0), "budget_gb_at_1024_samples": feat_gb},
                hbm_budget_scale=du.STATS.get("hbm_scale"),     # < 1: the cache budgets were clamped to the free HBM (min over ranks)
                fp_trace={"budget_gb_at_1024_samples": full_gb, "fp_prefix_sweeps": du.STATS["fp_captures"],
                          "units_served": du.STATS["units_served"], "memo_budget_gb_at_1024_samples": memo_gb,
                          "memo_hits": du.STATS["memo_hits"]},
                extrapolated_full_s={"caching_1024_samples": t_cache[0] * 1024 / n_calib,
                                     "fp_features_1024_samples": feat * 1024 / n_calib,
                                     "loops_1000_iters": setup + per_iter_all_units * 1000,
                                     "total": (t_cache[0] + feat) * 1024 / n_calib + setup + per_iter_all_units * 1000})


SHIPPED_RECON = dict(act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0, weight=0.0001, b_range=(20, 2),
                     warmup=0.2, batch_size=32, input_prob=0.5, add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=False)


def time_calibration(qnn, dev, n_calib=256, iters=40):
    """Bounded form (`--calib bounded`, and the multi-rank leg): the reconstruction walk on the full-size UNet with `n_calib`
    synthetic calibration rows and `iters` iterations per unit, extrapolated linearly to 1024 x 1000."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n_calib, 3, 64, 64, generator=g).to(dev)
    ts = np.arange(0, 1000, 50) + 1
    idx = torch.randint(0, 20, (n_calib,), generator=g)
    t = torch.tensor(ts[idx.numpy()], dtype=torch.long, device=dev)
    cond = torch.randn(n_calib, 1, 512, generator=g).to(dev)
    uncond = torch.randn(1, 1, 512, generator=g).expand(n_calib, 1, 512).contiguous().to(dev)
    cali = (x, t, idx.to(dev), cond, uncond)
    return instrumented_walk(qnn, cali, dict(cali_data=cali, iters=iters, **SHIPPED_RECON), n_calib)


def full_calibration(dev):
    """The whole calibration job at the shipped size (scripts/for_imagenet.sh:15-16 -> sample_diffusion_ldm_imagenet.py:142-199):
    1024 calibration samples in trajectory batches of 64, 20 DDIM steps, CFG 3.0, lambda 1.2; scale initialisation over all 1024;
    1000 iterations x every unit.  A fresh FP model (the job starts from the checkpoint, not from a quantised model)."""
    from scripts import sample_diffusion_ldm_imagenet as H
    args = H.parser().parse_args(["calibrate", "--calib_num_samples", "1024", "--batch_samples", "64", "--iters", "1000"])
    walk_out = {}

    def walk(qnn, cali, kwargs):
        walk_out.update(instrumented_walk(qnn, cali, kwargs, args.calib_num_samples))

    torch.cuda.reset_peak_memory_stats(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    ld, qnn, st = H.calibration_flow(args, dev, walk=walk)
    torch.cuda.synchronize()
    total = time.time() - t0
    r = dict(walk_out)
    r.pop("extrapolated_full_s", None)
    r["stages"] = {"model_build_and_wrap_s": total - sum(st.values()), "tdac_s": st["tdac_s"],
                   "scale_init_s": st["weight_scale_init_s"] + st["act_scale_init_s"],
                   "weight_scale_init_s": st["weight_scale_init_s"], "act_scale_init_s": st["act_scale_init_s"],
                   "caching_s": r["caching_s"], "fp_features_s": r["fp_features_s"], "loop_s": r["steady_iterations_s"] + r["unit_setup_s"],
                   "reconstruction_s": st["reconstruction_s"]}
    r["wall_s"] = st["tdac_s"] + st["weight_scale_init_s"] + st["act_scale_init_s"] + st["reconstruction_s"]
    r["measured_in_this_run"] = True
    r["config"] = "1024 TDAC calibration samples (trajectory batches of 64, 20 DDIM steps, CFG 3.0, lambda 1.2) -> " \
                  "set_{weight,act}_quantize_params_Conditional over all 1024 (batches of 32) -> 1000 iterations x %d units, batch 32, " \
                  "shipped kwargs of sample_diffusion_ldm_imagenet.py:165-196 (input_prob 0.5, quantizer prob 0.5)" % r["units"]
    r["peak_hbm_gb"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    return r


def rank_rows_ceiling(full, st, dp_loop, rest, world):
    """The multi-rank ceiling with the data-parallel iterations priced at what a rank's SMALLER minibatch costs, not at 1 / N of the
    32-row iteration: profiles/*_rank_rows.json (tools/rank_rows.py, committed) holds the measured ms per iteration of the data-parallel
    unit classes at 32 / 16 / 8 / 4 rows on ONE GPU; each eligible unit of THIS run is scaled by the ratio of its class (by positions per
    row, transformer block or not).  The per-iteration gather is priced, not measured (no multi-GPU node): <= 12 MB per rank over one
    xGMI link at 153 GB/s + 45 us of latency and graph A / gather / graph B hand-off."""
    if world != 1:
        return {}
    try:
        names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_rank_rows.json"))
        with open(os.path.join(ROOT, "profiles", names[-1])) as fh:
            rr = json.load(fh)["classes"]
    except Exception:
        return {}
    import edadm.recon as er
    iters = full.get("iters_per_unit", 1000)
    out = {}
    for n in (1, 2, 4, 8):
        rows = str(32 // n)
        t_dp = 0.0
        n_dp_iters = 0
        for u in full.get("per_unit_ms", []):
            if not u["data_parallel_eligible"]:
                continue
            pos = u["positions_per_row"]
            cls = ("tf@1024" if "transformer_blocks" in u["unit"] else ("res@4096" if pos >= 4096 else "res@1024"))
            ratio = rr.get(cls, rr["res@1024"])["ratio_to_32_rows"][rows]
            t_dp += u["ms_per_iteration"] * 1e-3 * (iters - 1) * ratio
            n_dp_iters += iters if n > 1 else 0
        gather = n_dp_iters * (12e6 / 153e9 + 45e-6)
        wall = rest + (st["tdac_s"] + st["caching_s"]) / n + t_dp + gather
        out[str(n)] = full["wall_s"] / wall
    return {"ceiling_measured_rows": out,
            "ceiling_measured_rows_source": "profiles/%s: per-rank iteration cost measured at 32 / N rows on one GPU; gather priced at one xGMI "
                                            "link (12 MB / 153 GB/s + 45 us); still no N > 1 hardware number" % names[-1]}


def time_h1_contraction(dev):
    """H1's dominant contraction on its own: the 3x3 192->192 convolution of the 64x64 level at the reconstruction batch
    (32 rows), forward, on both contraction paths, timed with events on the launch stream.  fp32-equivalent rate =
    2 M K N / t; the three-product path executes 3x those flops on the f16 MFMA (peak 2516 TFLOP/s dense), the
    exact path runs on the fp32 MFMA (peak 157 TFLOP/s)."""
    from edadm import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 64, 64, 192, generator=g).to(dev)
    w = (torch.randn(192, 3, 3, 192, generator=g) * 0.05).to(dev)
    b = torch.randn(192, generator=g).to(dev)
    flops = 2.0 * 32 * 64 * 64 * 9 * 192 * 192
    out = {}
    for name, fn in (("f16_three_product", lambda: ops.conv2d_f16x3_nhwc(x, w, b)), ("exact_fp32_mfma", lambda: ops.conv2d_f32_nhwc(x, w, b))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out[name] = {"ms": ms, "fp32_equivalent_tflops": flops / ms / 1e9}
    t = out["f16_three_product"]
    t["includes"] = "operand scan + f16 expansion of activations and filter + GEMM"
    t["f16_mfma_tflops"] = 3 * t["fp32_equivalent_tflops"]
    t["frac_of_f16_mfma_peak"] = t["f16_mfma_tflops"] / 2516.0
    out["exact_fp32_mfma"]["frac_of_fp32_mfma_peak"] = out["exact_fp32_mfma"]["fp32_equivalent_tflops"] / 157.0
    out["workload"] = "conv3x3 192->192, 32 x 64 x 64 NHWC fp32, forward"
    return out


def cpu_baseline(qnn, sd_cpu):
    """The oracle (CPU restatement of the reference's PyTorch fake-quant path) on this box's host cores.  One image of the metric is
    20 DDIM steps x one CFG-doubled UNet forward (2 rows) = 20 such forwards; timed here: as many of them as fit ~30 s of CPU work
    (at least 2, at most 20 = one whole image), every one a full fake-quant forward at its own timestep of the 20-step schedule --
    the per-forward cost does not depend on the step, so images/s = 1 / (20 x mean forward time); `extrapolation` states the factor."""
    from oracle import qdiff_oracle as O
    from edadm.state import quant_state_dict
    net = O.OUNet(sd_cpu, WQ, AQ, 8, **LDM4)
    net.set_first_last_layer_to_8bit()
    net.disable_network_output_quantization()
    net.split_shortcut = True
    g = torch.Generator().manual_seed(7)
    x, c = torch.randn(2, 3, 64, 64, generator=g), torch.randn(2, 1, 512, generator=g)
    steps = list(np.arange(0, 1000, 50) + 1)[::-1]
    with torch.no_grad():
        net(x, torch.tensor([501, 501]), c)                  # FP pass: creates split quantizers, warms up
    st = {"qp/" + k: v for k, v in quant_state_dict(qnn).items()}
    net.load_qparams(st, prefix="qp/model.")
    net.set_quant_state(True, True)
    n, total = 0, 0.0
    with torch.no_grad():
        while n < 20 and (n < 2 or total < 30.0):
            t0 = time.time()
            net(x, torch.tensor([int(steps[n])] * 2), c)
            total += time.time() - t0
            n += 1
    dt = total / n
    return dict(value=1.0 / (20 * dt), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample="%d of the 20 fake-quant UNet forwards (CFG-doubled, 2 rows) of ONE image, %.1f s of CPU work" % (n, total),
                extrapolation={"forwards_timed": n, "forwards_per_image": 20, "factor": 20.0 / n,
                               "note": "a forward costs the same at every timestep: images/s = 1 / (20 x mean forward time)"})


def calibration_cpu_baseline():
    """The calibration half of the metric on this box's host cores (SURVEY 8(d) CPU plan (ii)-(iv)): the oracle -- the CPU
    restatement of qdiff_control/block_recon.py:133-217, qdiff/quant_layer.py:234-244 and qdiff/data_utils.py:133-139 on the
    reference's own torch operators -- on ONE reconstruction unit of the headline model, the LDM-4 ResBlock 192 -> 384 at 32 x 32 (2.36 M
    AdaRound alphas), at the shipped minibatch of 32 rows and the shipped hyper-parameters: seconds per reconstruction iteration
    (three forwards + backward + two Adam steps), seconds per batch of the activation-scale initialisation (the MSE search of every
    activation quantiser of the unit) and of the FP target forward.  Random-init weights; a bounded sample: two iterations."""
    import random
    from oracle import qdiff_oracle as O
    g = torch.Generator().manual_seed(11)
    shapes = {"in_layers.0.weight": (192,), "in_layers.0.bias": (192,), "in_layers.2.weight": (384, 192, 3, 3), "in_layers.2.bias": (384,),
              "emb_layers.1.weight": (384, 768), "emb_layers.1.bias": (384,), "out_layers.0.weight": (384,), "out_layers.0.bias": (384,),
              "out_layers.3.weight": (384, 384, 3, 3), "out_layers.3.bias": (384,), "skip_connection.weight": (384, 192, 1, 1),
              "skip_connection.bias": (384,)}
    sd = {}
    for k, shp in shapes.items():
        if len(shp) == 1:
            sd["res." + k] = (torch.ones(shp) if k.endswith("0.weight") else torch.zeros(shp)) + 0.1 * torch.randn(shp, generator=g)
        else:
            fan = int(np.prod(shp[1:]))
            sd["res." + k] = torch.randn(shp, generator=g) / math.sqrt(fan)
    unit = O.OResBlock(O._Builder(sd, WQ, AQ, 8), "res", "res", 192, 384)
    rows = 32
    x, emb = torch.randn(rows, 192, 32, 32, generator=g), torch.randn(rows, 768, generator=g)
    xq, eq = x + 0.05 * torch.randn(x.shape, generator=g), emb + 0.02 * torch.randn(emb.shape, generator=g)
    t = {}
    with torch.no_grad():
        unit.set_quant_state(False, False)
        t0 = time.time()
        out_fp = unit(x, emb)
        t["fp_target_forward_s_per_batch"] = time.time() - t0
        unit.set_quant_state(True, False)                      # weight scales: per-channel MSE search on the first quantised forward
        t0 = time.time()
        unit(xq, eq)
        t["weight_scale_init_s"] = time.time() - t0
        for l in unit.layers():
            l.weight_quantizer.inited = True
        unit.set_quant_state(True, True)                       # activation scales: MSE search per quantiser on this batch
        t0 = time.time()
        unit(xq, eq)
        t["act_scale_init_s_per_batch"] = time.time() - t0
        for l in unit.layers():
            for q in l.quantizers():
                q.inited = True
    stamps = [time.time()]

    class _Stop(Exception):
        pass

    def trace(it, wp, ap, loss):
        stamps.append(time.time())
        if len(stamps) == 3:
            raise _Stop

    random.seed(7)
    try:
        O.reconstruct_unit(None, unit, "block", cali=None, iters=1000, act_quant=True, lr_a=SHIPPED_RECON["lr_a"], lr_w=SHIPPED_RECON["lr_w"], p=2.0,
                           batch_size=rows, input_prob=SHIPPED_RECON["input_prob"], add_loss=SHIPPED_RECON["add_loss"], recon_w=True,
                           recon_a=True, caches=(True, (xq, eq), (x, emb), out_fp), trace=trace)
    except _Stop:
        pass
    it_s = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
    per_iter = float(np.mean(it_s))
    return dict(value=per_iter, unit="s per reconstruction iteration (one unit, 32 rows)", cores=torch.get_num_threads(), kind="port",
                sample="LDM-4 ResBlock 192 -> 384 at 32 x 32 (2 359 296 alphas), shipped hyper-parameters (input_prob 0.5, quantizer prob 0.5, add_loss "
                       "0.8): %d iterations timed, %.1f s of CPU work in all" % (len(it_s), sum(it_s) + sum(t.values())),
                iterations_s=it_s, **t,
                note="the job runs 1000 such iterations on each of 80 units (this one is a mid-sized one: 4.97 ms per iteration on the GPU, "
                     "`per_unit_ms`) after 32 batches of scale initialisation and 2 x 32 cached forward batches per unit")


def emit(line):
    """Rank 0's ONE stdout line: the compact form (bench_line.py, <= 6 KB); the detailed dict goes to bench_detail.json."""
    import bench_line
    bench_line.write_detail(line, ROOT)
    sys.stdout.write(bench_line.dumps(line) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent 50-image sample batches in flight per GPU (edadm.sampling.InFlightSampler: batch k on stream k mod n; "
                         "measured 1 / 2 / 3 / 4 / 5 in flight, round 6: 97.5 / 108.4 / 110.3-111.9 / 110.6 / 108.3 decoded, 120.5 / 141.9 / 140.8-141.3 / "
                         "140.3 / 138.5 sampling-only images/s)")
    ap.add_argument("--calib", choices=["full", "bounded", "none"], default=None,
                    help="full (default at N = 1): the whole calibration job at the shipped size, measured (~8 min); with N > 1 every rank runs "
                         "it (TDAC and activation caching sharded) and the line carries the max-over-ranks wall-clock -- not the default there, "
                         "the sampling throughput is; bounded (N = 1): a 256-sample x 40-iteration reconstruction walk extrapolated linearly "
                         "(~1 min); none")
    ap.add_argument("--no-calib", action="store_true", help="= --calib none")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` key (one UNet call of configs 2, 3, 5 at full size, ~1 min)")
    ap.add_argument("--calib-ranks", action="store_true",
                    help="with --gpus N > 1: also time a bounded reconstruction walk with the activation caching sharded over the "
                         "ranks (all_gather_into_tensor of the cached slabs + broadcast of the learned parameters per unit)")
    args = ap.parse_args()
    if args.no_calib:
        args.calib = "none"
    if args.calib is None:
        args.calib = "full" if args.gpus <= 1 else "none"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start N ranks (one process per GPU) through torch.distributed.run as a
        # CHILD process and hand its output through.  Nothing in this process has touched the GPU (an exec or a
        # fork after HIP initialisation is not allowed on this pool).
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
        assert dist.get_world_size() == args.gpus
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                                  # RCCL over xGMI: every rank contributes a 1
        ranks_seen = int(seen.item())
        assert ranks_seen == world, (ranks_seen, world)

    from edadm import lib, ops
    lib.load()                                                # fail loudly if the HIP library is missing
    from edadm.sampling import DDIMLoop

    qnn, sd_cpu, calib = build_quantised_unet(dev)
    eng = qnn.freeze()
    B = args.batch
    mk = lambda cs=None: DDIMLoop(eng, (3, 64, 64), B, steps=20, eta=0.0, scale=3.0, context_shape=(1, 512), device=dev, capture_stream=cs)
    flight = None
    if args.inflight > 1:
        from edadm.sampling import InFlightSampler
        flight = InFlightSampler(mk, n=args.inflight, device=dev)
        loop = flight.loops[0]
    else:
        loop = mk()
    dec = make_decoder(dev)
    side = torch.cuda.Stream(device=dev)
    # a batch is a pure function of (seed, global batch index) (edadm/sample_driver.py): rank r makes the batches
    # {i : i mod world = r} of the one global sequence, so the union over ranks is the same images whatever N is
    from edadm.sample_driver import batch_noise, batch_generator
    n_total = args.steps + args.warmup
    noise = [batch_noise(1234, k * world + rank, (B, 3, 64, 64), dev) for k in range(n_total)]
    gen = batch_generator(1234, 10 ** 9, dev)                  # the class embeddings: shared by all ranks
    cond = torch.randn(B, 1, 512, generator=gen, device=dev)
    uncond = torch.randn(1, 1, 512, generator=gen, device=dev).expand(B, 1, 512).contiguous()

    def flight_now():
        return flight

    def timed(decoder, serial=False):
        """W untimed + exactly K timed steps between barrier + synchronize on both sides; max over ranks.  serial: one batch in
        flight (the loop of the first in-flight slot on the current stream) instead of the InFlightSampler.  (`flight` is read from
        the enclosing scope at call time: a default argument would keep the sampler, its graphs and static buffers alive through the
        calibration job below.)"""
        flight = None if serial else flight_now()
        run_steps(loop, decoder, side, noise, cond, uncond, 0, args.warmup, dev, flight)
        if args.warmup == 0 and decoder is not None:
            with torch.cuda.stream(side):
                decoder(torch.zeros(B, 3, 64, 64, device=dev))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.time()
        n = run_steps(loop, decoder, side, noise, cond, uncond, args.warmup, n_total, dev, flight)
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert n == B * args.steps
        if world > 1:
            dist.barrier()
            tt = torch.tensor([dt], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    elapsed = timed(dec)                      # the headline: sampled AND decoded
    elapsed_unet = timed(None)                # the quantised UNet sampling alone
    # the same K steps with ONE batch in flight (the serial loop of the reference's script): what the kernels alone deliver
    elapsed_one = timed(dec, serial=True) if flight is not None else elapsed
    elapsed_unet_one = timed(None, serial=True) if flight is not None else elapsed_unet
    ops.device_status()                       # a deferred in-kernel failure (persistent-GEMM hand-off timeout) raises here

    # roofline pass: HIP events around every int8 MFMA GEMM launch of one eager UNet call as the sampling loop issues it
    # per step (the context-only cross-attention vectors and the time-embedding rows come in precomputed, as in the step
    # graph)
    x_in = torch.cat([noise[0], noise[0]])
    t_in = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
    c_in = torch.cat([uncond, cond])
    eng.ctx_r = eng.context_branches(c_in)
    eng.emb_r = eng.emb_rows(t_in)           # ... and the step's time-embedding rows from the run's table
    eng.cfg_pair = True                      # x_in is a guidance pair [x, x]: context-independent prefix once per pair
    eng.prof = []
    eng(x_in, t_in, c_in)
    torch.cuda.synchronize()
    prof, eng.prof = eng.prof, None
    eng.ctx_r = eng.emb_r = None
    eng.cfg_pair = False

    # Each recorded launch is timed on its own between two HIP events (on the launch stream), `reps` times, and between two timed
    # launches a 320 MB buffer is rewritten: the operands of the next launch are no longer in L2 (32 MB) or the Infinity Cache
    # (256 MB) -- colder than in situ, where the producer has just written them.  (Round 3 replayed each launch 5x back to back:
    # 15.57 ms for the group against 16.03 ms in the rocprof trace of the real call.)  `roofline` uses these cold numbers.
    flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device=dev)

    def kernel_ms(run, reps=3):
        run()
        total = 0.0
        for r in range(reps):
            flush.fill_(float(r))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            total += e0.elapsed_time(e1)
        return total / reps

    i8 = [(f, kernel_ms(run), sum(v for k, v in by.items() if k != "kind")) for mode, _, _, _, _, f, run, by in prof if mode == "i8"]
    gemm_flop, gemm_ms = sum(r[0] for r in i8), sum(r[1] for r in i8)
    gemm_alg_bytes = sum(r[2] for r in i8)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    loop.unet(x_in, t_in, c_in, step=0)      # new context tensor: its branch graph replays here, outside the timing
    ev0.record()
    loop.unet(x_in, t_in, c_in, step=0)      # one step of the loop: table row copy + step graph
    ev1.record()
    torch.cuda.synchronize()
    unet_ms = ev0.elapsed_time(ev1)

    # HBM bytes per launch of the dominant kernel group from the committed rocprofv3 --pmc passes (tools/prof_round6.sh step 2).  The
    # file names the sources it was measured on (sha256 of csrc/gemm.hip); a file of another build is NOT used: traffic = null
    traffic, traffic_src = None, None
    import hashlib
    with open(os.path.join(ROOT, "eda-dm_amd", "csrc", "gemm.hip"), "rb") as fh:
        gemm_sha = hashlib.sha256(fh.read()).hexdigest()
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gemm_traffic.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                tj = json.load(fh)
            if tj.get("gemm_hip_sha256") != gemm_sha:
                traffic_src = "stale: profiles/%s was measured on another csrc/gemm.hip" % name if traffic_src is None else traffic_src
                continue
            # bytes of all int8 GEMM launches of one UNet call / the GEMM calls timed above (a split or tail-re-tiled
            # layer is two device launches of one call)
            if "hbm_bytes_per_unet_call" in tj:               # round 5 format: kernels selected by the engine's launch list
                traffic = tj["hbm_bytes_per_unet_call"] / max(len(i8), 1)
            else:
                traffic = tj["hbm_bytes_per_launch"] * tj["launches"] / tj.get("unet_calls", 4) / max(len(i8), 1)
            traffic_src = "profiles/%s (rocprofv3 --pmc passes of tools/prof_traffic.sh on this csrc/gemm.hip, committed; not re-measured in this run)" % name
            break
        except Exception:
            pass
    decode = None
    if world == 1:
        try:
            d = time_decoder(dec, dev, B)
            d["roofline"] = {"bound": "mfma", "achieved": 3 * d["tflops_fp32"], "peak": 2516.0, "unit": "TFLOP/s",
                             "frac": 3 * d["tflops_fp32"] / 2516.0,
                             "note": "three f16 MFMA products per fp32 product (fp32-grade result): fp32-equivalent rate x 3 "
                                     "against the dense f16 MFMA peak"}
            decode = d
        except Exception as e:
            decode = {"error": repr(e)}
    calib_mr = None
    if world > 1 and args.calib_ranks:
        # every rank runs the walk (the loop is replicated); the caching batches are sharded and all-gathered
        from edadm import dist as edist
        edist.GATHER_STATS.update(bytes=0, calls=0)
        r = time_calibration(qnn, dev, n_calib=32 * world * 2, iters=3)
        calib_mr = {"ranks": world, "units": r["units"], "calib_samples": r["calib_samples"], "caching_s": r["caching_s"],
                    "caching_s_per_unit": r["caching_s"] / max(r["units"], 1), "gathered_bytes": edist.GATHER_STATS["bytes"],
                    "gather_calls": edist.GATHER_STATS["calls"], "loop_s": r["loop_s"],
                    "transport": "all_gather_into_tensor over RCCL (one slab per cached tensor, in place) + broadcast of alphas / step sizes from rank 0"}
    if rank == 0:
        images = B * args.steps * world
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
        ips, ips_unet = images / elapsed, images / elapsed_unet
        line = {
            "metric": "images/sec W4A8 LDM-4 ImageNet 256x256 (sampled + VQ-f4 decoded); calibration wall-clock in `calibration`",
            "metric_definition": "20 DDIM steps x CFG 3.0 on the int8 executor AND VQ-f4 decode to pixels (what the reference's loop "
                                 "delivers per batch); full calibration wall-clock under `calibration`",
            "value": ips, "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8",
            "dtype_note": "i8 MFMA, i32 accumulate, fp32 epilogues; first-stage decoder fp32-grade (f16 x3 MFMA)",
            "data": "synthetic",
            "config": {"workload": "configs[3]: ImageNet LDM-4 256x256 W4A8, 50-image batches x 20 DDIM steps x CFG 3.0 (100 UNet rows/call), "
                                   "400.9M-param UNet + VQ-f4 decode, random-init weights, inputs resident in HBM",
                       "workload_note": "each batch decoded by "
                                   "the VQ-f4 first stage (55.3M params, fp32) -- issued on a second stream, but NOT hidden: both want the same CUs and "
                                   "the decode is ~20 %% of a step (`first_stage_decode`); %d independent batches are in flight on %d streams "
                                   "(edadm.sampling.InFlightSampler: a second batch's launches fill the idle slots between the ~500 "
                                   "dependent launches of a UNet call; each batch has the bits of the serial loop); the one-token "
                                   "cross-attention vectors (a function of the context alone) and the time-embedding rows of "
                                   "the 20 timesteps are evaluated once per batch inside the timed sample() call, and the "
                                   "attention-free leading blocks (identical for the two halves of a guidance pair) once "
                                   "per pair: all bit-identical to the plain evaluation" % (args.inflight, args.inflight),
                       "batches_in_flight": args.inflight,
                       "images_per_step": B, "ddim_steps": 20, "cfg_scale": 3.0, "parallelism": "dp%d (independent batches, no collective)" % world},
            "sampling_only": {"metric": "images/sec of the quantised UNet sampling path alone (latents, not decoded)", "value": ips_unet,
                              "ms_per_step": 1e3 * elapsed_unet / args.steps, "steps": args.steps, "warmup": args.warmup},
            "one_batch_in_flight": {"value": images / elapsed_one, "sampling_only": images / elapsed_unet_one,
                                    "ms_per_step": 1e3 * elapsed_one / args.steps,
                                    "note": "the same K steps issued serially on one stream (no InFlightSampler): kernels alone"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / I8_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "frac_definition": "dominant kernel GROUP: executed flops of every int8 GEMM launch of one UNet call / their summed "
                                            "device time (each launch timed COLD: 3 repetitions between HIP events, a 320 MB buffer rewritten in front of each so that "
                                            "no operand is left in L2 / the Infinity Cache from the previous one) / dense int8 MFMA peak",
                         "frac_survey_8d": {"definition": "SURVEY 8(d): images/s x algorithmic FLOP/image / peak, on the per-GPU rate",
                                            "unet_flop_per_image": 40 * UNET_GFLOP_PER_ROW * 1e9,
                                            "sampling_only": ips_unet / world * 40 * UNET_GFLOP_PER_ROW * 1e9 / (I8_PEAK_TFLOPS * 1e12),
                                            "sampled_and_decoded": ips / world * 40 * UNET_GFLOP_PER_ROW * 1e9 / (I8_PEAK_TFLOPS * 1e12)},
                         "algorithmic_bytes": gemm_alg_bytes / max(len(i8), 1),
                         "algorithmic_bytes_note": "per GEMM call, like `traffic`: int8 activation tensor + integer weights + output in "
                                                   "its stored type + fp32 residual, each element once (a convolution's input counted "
                                                   "once, not 9x); traffic / algorithmic_bytes = re-read factor",
                         "kernel": "int8 GEMM group, %d launches per UNet call, %.1f GFLOP, %.2f ms (cold)" % (len(i8), gemm_flop / 1e9, gemm_ms),
                         "kernel_note": "edadm_qgemm_i8/_q and edadm_qconv3_i8_direct: k_conv3_direct, k_gemm_nt, k_gemm_p, k_gemm_nt8",
                         "hbm": {"note": "same launches against the HBM roof: PMC bytes per launch x launches / summed time",
                                 "achieved_GBps": (traffic * len(i8) / (gemm_ms * 1e-3) / 1e9) if traffic else None,
                                 "peak_GBps": 8000.0,
                                 "frac": (traffic * len(i8) / (gemm_ms * 1e-3) / 8e12) if traffic else None},
                         "in_flight": {"batches": args.inflight,
                                       "note": "`frac` above prices every launch alone; in the timed loop %d batches share the chip, and the int8 GEMM "
                                               "flops the sampling-only loop sustains over ITS wall time (all kernels of a call included) are" % args.inflight,
                                       "unet_calls_per_s": ips_unet / world / B * 20,
                                       "sustained_int8_gemm_tflops": ips_unet / world / B * 20 * gemm_flop / 1e12,
                                       "frac": ips_unet / world / B * 20 * gemm_flop / 1e12 / I8_PEAK_TFLOPS},
                         "unet_call_ms": unet_ms,
                         "unet_algorithmic_tflops": 2 * B * UNET_GFLOP_PER_ROW / unet_ms},
            "calibration": {"quick_scale_init_of_the_sampling_model": calib},
        }
        if decode is not None:
            line["first_stage_decode"] = decode
        if calib_mr is not None:
            line["calibration"]["multi_rank_bounded_walk"] = calib_mr
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(qnn, sd_cpu)
            except Exception as e:      # the baseline is a report, never a reason to lose the bench line
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": torch.get_num_threads(),
                                        "kind": "port", "sample": "failed: %r" % (e,)}
            try:
                line["calibration"]["cpu_baseline"] = calibration_cpu_baseline()
            except Exception as e:
                line["calibration"]["cpu_baseline"] = {"value": None, "unit": "s per reconstruction iteration", "cores": torch.get_num_threads(),
                                                       "kind": "port", "sample": "failed: %r" % (e,)}
    # ---- configs 2, 3, 5 at full size, timed by THIS run (tools/config_bench.py::quick_call_numbers): one UNet call each at the shipped
    # rows per call on the frozen int8 executor
    if rank == 0 and world == 1 and not args.no_configs:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        cfgs = {}
        try:
            import config_bench as cb
            loop = eng = dec = prof = i8 = flight = None                # the headline's engine, graphs and decoder leave HBM first
            qnn.engine = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            from types import SimpleNamespace
            # the sampling loop of each configuration at full size (tools/config_bench.py: CIFAR 500 x 100 quad-skip DDIM steps,
            # Church 100 x 20 of the shipped 500 steps, Stable Diffusion 4 prompts x CFG x 50 PLMS steps), ONE batch timed after a
            # warm-up batch: images / s, the UNet call's time inside the loop and a roofline per configuration -- the algorithmic
            # rate of the whole loop (images / s x FLOP / image) and the int8 GEMM group's executed rate, both from
            # events / wall time of THIS run
            for kind, fn in (("cifar", cb.run_cifar), ("church", cb.run_church), ("sd", cb.run_sd)):
                t0c = time.time()
                try:
                    with torch.no_grad():
                        r = fn(dev, SimpleNamespace(batch=0, steps=0, calls=0, batches=1, inflight=args.inflight))
                    r["batches_in_flight"] = 1 if kind == "cifar" else args.inflight
                    gg = r.get("gemm_group", {})
                    r["roofline"] = {"bound": "mfma", "achieved": r["algorithmic_tflops"], "peak": I8_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": r["algorithmic_tflops"] / I8_PEAK_TFLOPS,
                                     "definition": "algorithmic conv / linear / attention matmul FLOPs of the loop's UNet calls (SURVEY 8(d) per-row "
                                                   "figures) / wall time of the timed batch / dense int8 MFMA peak",
                                     "int8_gemm_group": {"achieved": gg.get("int8_gemm_tflops"), "frac": gg.get("frac_of_int8_mfma_peak"),
                                                         "launches": gg.get("int8_gemm_calls"), "ms": gg.get("int8_gemm_ms"),
                                                         "definition": "executed flops of every int8 GEMM launch of one UNet call / their summed device "
                                                                       "time (HIP events, warm)"}}
                except Exception as e:
                    r = {"error": repr(e)}
                r["wall_s"] = time.time() - t0c
                cfgs[kind] = r
                gc.collect()
                torch.cuda.empty_cache()
        except Exception as e:
            cfgs["error"] = repr(e)
        line["configs"] = cfgs
    # ---- the calibration job.  N = 1: bounded or full.  N > 1 with --calib full: EVERY rank runs the job -- TDAC trajectory batches and
    # the activation caching are sharded (one all_gather_into_tensor per tensor / cached slab), scale initialisation and the
    # reconstruction loops run replicated with rank 0's learned parameters broadcast after each unit -- and the wall-clock is the
    # max over ranks between two barriers.
    calib_out = {}
    if world > 1 and args.calib == "full":
        # A rank that fails inside the job skips the collectives the others are blocked in.  It cannot tell them, so it leaves with a
        # non-zero status (below) and the launcher tears the group down; rank 0 answers the launcher's SIGTERM by printing the line
        # it has -- sampling numbers plus calibration.error -- and leaves non-zero too: no rank exits 0 after a failed job.
        # ... and, because a rank blocked inside a collective or a stream synchronisation never reaches a Python-level signal handler before
        # the launcher escalates to SIGKILL, rank 0 prints the line it has NOW (sampling numbers, no calibration yet): whatever happens
        # to the job below, the driver finds a parseable last line; a job that completes prints the full line after it.
        if rank == 0:
            line["calibration"]["status"] = "multi-rank job started; this line was printed before it"
            emit(line)
            line["calibration"].pop("status", None)
        import signal

        def _torn_down(signum, frame):
            if rank == 0:
                line["calibration"]["error"] = "a rank failed inside the %d-rank calibration job (its traceback is on stderr); torn down" % world
                emit(line)
            os._exit(14)
        signal.signal(signal.SIGTERM, _torn_down)
    if args.calib != "none" and (world == 1 or args.calib == "full"):
        try:
            if args.calib == "bounded":
                calib_out["reconstruction_bounded"] = time_calibration(qnn, dev)
            else:
                # the sampling model, its engine, graphs and the decoder leave HBM first: the job needs ~175 GB
                loop = eng = dec = prof = i8 = flight = None
                qnn.engine = None
                qnn = None
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                if world > 1:
                    dist.barrier()
                try:
                    full = full_calibration(dev)
                except torch.OutOfMemoryError as oom:
                    # the job peaks near 210 of the 288 GB and the allocator's fragmentation decides whether the last 2.5 GB slab fits:
                    # ONE retry (single rank only: the ranks of a sharded job must agree) from an empty cache with the cache budgets at
                    # 60 % -- more FP sweeps, the same result; the line says so
                    if world > 1:
                        raise
                    sys.stderr.write("calibration: out of memory (%s); retrying once with the cache budgets at 60 %%\n" % str(oom)[:200])
                    import qdiff.data_utils as du
                    import edadm.recon as er
                    gc.collect()
                    torch.cuda.empty_cache()
                    keep = du.FP_TRACE_GB, du.Q_MEMO_GB, er.FP_FEAT_GB
                    du.FP_TRACE_GB, du.Q_MEMO_GB, er.FP_FEAT_GB = 0.6 * keep[0], 0.6 * keep[1], 0.6 * keep[2]
                    try:
                        full = full_calibration(dev)
                    finally:
                        du.FP_TRACE_GB, du.Q_MEMO_GB, er.FP_FEAT_GB = keep
                    full["retried_after_oom_with_budgets_at"] = 0.6
                if world > 1:
                    dist.barrier()
                    tt = torch.tensor([full["wall_s"]] + [full["stages"][k] for k in ("tdac_s", "scale_init_s", "caching_s", "loop_s")],
                                      device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    full["wall_s"] = float(tt[0])
                    full["stages_max_over_ranks"] = dict(zip(("tdac_s", "scale_init_s", "caching_s", "loop_s"), [float(v) for v in tt[1:]]))
                calib_out.update(full)
                calib_out["metric"] = "full calibration+recon wall-clock, W4A8 LDM-4 ImageNet 256x256, %d x MI355X" % world
                calib_out["value_s"] = full["wall_s"]
                st = full["stages"]
                dp_loop = full.get("dp_eligible_loop_s", 0.0)
                sharded = st["tdac_s"] + st["caching_s"] + dp_loop
                rest = full["wall_s"] - sharded
                calib_out["multi_rank"] = {
                    "ranks": world,
                    "sharded_stages": "TDAC trajectory batches (one gather of the calibration latents), activation caching of every unit "
                                      "(one gather per cached slab), and the reconstruction iterations of EVERY unit -- block or single layer -- with >= 1024 "
                                      "positions per row (64 x 64 and 32 x 32 levels: the 32-row minibatch split over the ranks, partial gradient slabs "
                                      "all-gathered and added in rank order, SURVEY 8e(2))",
                    "replicated_stages": "scale initialisation (the activation ranges are an EMA over the batch sequence), the iterations of the "
                                         "units below 1024 positions per row (16 x 16 / 8 x 8 levels, time-embedding layers: launch-latency bound), per-unit set-up",
                    "sharded_s_this_run": sharded, "of_which_data_parallel_loop_s": dp_loop, "replicated_s_this_run": rest,
                    # what N ranks can gain at best with this split: the stages of THIS run, sharded ones divided by N / this N
                    "ceiling": {str(n): (rest + sharded * world) / (rest + sharded * world / n) for n in (1, 2, 4, 8)},
                    **rank_rows_ceiling(full, st, dp_loop, rest, world),
                    "ceiling_note": "speed-up over one rank if the sharded stages scaled perfectly, from the per-stage / per-unit seconds "
                                    "measured in THIS run (collectives and the smaller per-rank kernels' efficiency not priced: an upper "
                                    "bound); no multi-GPU node was available to this build -- the N-rank path is exercised by the "
                                    "2-process tests only"}
            if rank == 0:
                calib_out["h1_contraction"] = time_h1_contraction(dev)
        except Exception as e:
            import traceback
            calib_out["error"] = repr(e) + " | " + traceback.format_exc()[-600:]
            if world > 1:
                sys.stderr.write("rank %d: calibration failed: %s\n" % (rank, calib_out["error"]))
                sys.stderr.flush()
                if rank == 0:
                    line["calibration"].update(calib_out)
                    emit(line)
                os._exit(13)                                   # no collective after a failure: the launcher ends the other ranks
    if rank == 0:
        line["calibration"].update(calib_out)
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
