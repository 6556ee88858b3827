"""Task harness for class-conditional LDM-4 (ImageNet 256x256) — the flow of the reference's
scripts/sample_diffusion_ldm_imagenet.py:142-249 (and its launcher scripts/for_imagenet.sh) over this build's API, split
into the two jobs the saved calibration state allows:

    calibrate   TDAC calibration set -> set_{weight,act}_quantize_params_Conditional -> recon_block_Qmodel
                -> quantiser state (edadm/state.py) + frozen W4-packed integer model written to --out
    sample      load the frozen integer model on every rank, shard the batches {i : i mod world = rank}
                (edadm/sample_driver.py: a batch is a function of (seed, batch index)), DDIM + CFG on the int8
                executor, first-stage decode, images / latents written per rank

    python -m scripts.sample_diffusion_ldm_imagenet calibrate --out calib/ [--calib_num_samples 1024 --iters 1000]
    python -m torch.distributed.run --nproc-per-node 8 -m scripts.sample_diffusion_ldm_imagenet sample --state calib/ --n_samples 50000

No checkpoint or dataset ships with the reference tree (`ckpt_util.py` downloads them): `--synthetic` (default when
--ckpt is absent) builds the cin256-v2 UNet with seeded random weights, a seeded class-embedding table and, for the
decoder, seeded VQ-f4 weights.  Argument names follow the reference script."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)

LDM4 = dict(image_size=64, in_channels=3, out_channels=3, model_channels=192, attention_resolutions=[8, 4, 2],
            num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_heads=1, use_spatial_transformer=True,
            transformer_depth=1, context_dim=512)
VQF4 = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
            resolution=256, z_channels=3)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("job", choices=["calibrate", "sample"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--unet", type=json.loads, default=None, help="UNetModel kwargs as JSON (default: cin256-v2 LDM-4)")
    ap.add_argument("--latent", type=int, nargs=3, default=[3, 64, 64])
    ap.add_argument("--custom_steps", type=int, default=20)
    ap.add_argument("--ddim_eta", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=3.0)
    ap.add_argument("--weight_bit", type=int, default=4)
    ap.add_argument("--act_bit", type=int, default=8)
    ap.add_argument("--sm_abit", type=int, default=8)
    ap.add_argument("--split", action="store_true", default=True)
    # calibrate
    ap.add_argument("--out", default="calib_state")
    ap.add_argument("--calib_num_samples", type=int, default=1024)
    ap.add_argument("--batch_samples", type=int, default=32)
    ap.add_argument("--lamda", type=float, default=1.2)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr_a", type=float, default=1e-4)
    ap.add_argument("--lr_w", type=float, default=5e-1)
    ap.add_argument("--add_loss", type=float, default=0.8)
    ap.add_argument("--no_recon", action="store_true")
    # sample
    ap.add_argument("--state", default="calib_state")
    ap.add_argument("--n_samples", type=int, default=50000)
    ap.add_argument("--n_batch", type=int, default=50)
    ap.add_argument("--max_batches", type=int, default=None, help="per rank (smoke runs)")
    ap.add_argument("--no_decode", action="store_true")
    ap.add_argument("--inflight", type=int, default=2, help="sample batches in flight per GPU (edadm.sampling.InFlightSampler)")
    ap.add_argument("--save", default=None, help="directory for per-rank .npy batches (default: count only)")
    return ap


def build_models(args, dev):
    """(LatentDiffusionLite around the FP UNet, class embedder).  Synthetic weights: see module docstring."""
    from edadm.nets.ldm_unet import UNetModel
    from edadm.latent import LatentDiffusionLite, ClassEmbedder
    from qdiff.utils import seed_everything
    seed_everything(args.seed)
    kw = args.unet or LDM4
    unet = UNetModel(**kw)
    g = torch.Generator().manual_seed(args.seed)
    for prm in unet.parameters():                           # zero_module convolutions: give them weights
        if float(prm.detach().abs().max()) == 0.0:
            with torch.no_grad():
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    ce = ClassEmbedder(kw["context_dim"], n_classes=1001)
    ld = LatentDiffusionLite(unet, timesteps=1000, linear_start=0.0015, linear_end=0.0195, conditioning_key="crossattn",
                             cond_stage_model=ce, cond_stage_key="class_label")
    return ld.to(dev).eval()


def quantise(ld, args):
    from qdiff import QuantModel
    wq = {'n_bits': args.weight_bit, 'symmetric': True, 'channel_wise': True, 'scale_method': 'mse'}
    aq = {'n_bits': args.act_bit, 'symmetric': True, 'channel_wise': False, 'scale_method': 'mse', 'leaf_param': True, 'prob': 0.5}
    qnn = QuantModel(model=ld.model.diffusion_model, weight_quant_params=wq, act_quant_params=aq, act_quant_mode="qdiff",
                     sm_abit=args.sm_abit)
    qnn.cuda().eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)
    ld.model.diffusion_model = qnn
    return qnn


def calibration_flow(args, dev, walk=None):
    """The calibration job of sample_diffusion_ldm_imagenet.py:142-199 stage by stage: FP model -> QuantModel -> TDAC calibration
    set -> set_{weight,act}_quantize_params_Conditional over ALL calibration samples -> conditional reconstruction walk.
    `walk(qnn, cali, kwargs)` replaces the plain walk (bench.py passes its instrumented one).  Returns (ld, qnn, stage seconds)."""
    from scripts.calibration import TDAC_imagenet_calib_data_generator
    from qdiff_control import (set_weight_quantize_params_Conditional, set_act_quantize_params_Conditional,
                               recon_block_Qmodel)

    def now():
        torch.cuda.synchronize()
        return time.time()

    ld = build_models(args, dev)
    qnn = quantise(ld, args)
    args.latent_shape = list(args.latent)
    args.data = torch.randint(0, 1000, (args.calib_num_samples,), generator=torch.Generator().manual_seed(args.seed)).to(dev)
    t0 = now()
    cali = TDAC_imagenet_calib_data_generator(ld, args, args.calib_num_samples, args.batch_samples, dev, args.custom_steps)
    t1 = now()
    if args.split:
        qnn.model.split_shortcut = True
    set_weight_quantize_params_Conditional(ld, cali, args)
    t2 = now()
    set_act_quantize_params_Conditional(ld, cali, args)
    t3 = now()
    stages = {"tdac_s": t1 - t0, "weight_scale_init_s": t2 - t1, "act_scale_init_s": t3 - t2, "reconstruction_s": 0.0}
    if not args.no_recon:
        kwargs = dict(cali_data=cali, iters=args.iters, act_quant=True, asym=True, opt_mode='mse', lr_a=args.lr_a, lr_w=args.lr_w,
                      p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=args.add_loss,
                      recon_w=True, recon_a=True, keep_gpu=False)
        qnn.set_quant_state(True, True)
        if walk is not None:
            walk(qnn, cali, kwargs)
        else:
            ld.model.diffusion_model = recon_block_Qmodel(args, qnn, cali, kwargs).recon()
    qnn.set_quant_state(True, True)
    stages["reconstruction_s"] = now() - t3
    return ld, qnn, stages


def calibrate(args):
    from edadm import state
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.distributed.init_process_group("nccl")
    ld, qnn, st = calibration_flow(args, dev)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        np.savez(os.path.join(args.out, "quant_state.npz"), **state.quant_state_dict(qnn))
        nbytes = state.save_frozen(qnn, os.path.join(args.out, "frozen.npz"))
        torch.save(ld.cond_stage_model.state_dict(), os.path.join(args.out, "class_embedder.pt"))
        print(json.dumps({"job": "calibrate", "units": qnn.block_count, "tdac_s": st["tdac_s"],
                          "scale_init_s": st["weight_scale_init_s"] + st["act_scale_init_s"],
                          "reconstruction_s": st["reconstruction_s"], "frozen_bytes": nbytes, "out": args.out}))


def sample(args):
    from edadm import state, dist as edist
    from edadm.sampling import DDIMLoop
    from edadm.sample_driver import ShardedSampler
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        torch.distributed.init_process_group("nccl")
    ld = build_models(args, dev)
    qnn = quantise(ld, args)
    if args.split:
        qnn.model.split_shortcut = True
    with torch.no_grad():                                    # one FP pass creates the split quantizers
        C, H, W = args.latent
        ctx_dim = (args.unet or LDM4)["context_dim"]
        qnn(torch.zeros(2, C, H, W, device=dev), torch.zeros(2, dtype=torch.long, device=dev), torch.zeros(2, 1, ctx_dim, device=dev))
    with np.load(os.path.join(args.state, "quant_state.npz"), allow_pickle=False) as z:
        state.load_quant_state(qnn, {k: z[k] for k in z.files})
    qnn.set_quant_state(True, True)
    eng = state.load_frozen(qnn, os.path.join(args.state, "frozen.npz"))
    ld.cond_stage_model.load_state_dict(torch.load(os.path.join(args.state, "class_embedder.pt"), map_location=dev))
    B = args.n_batch
    mk = lambda cs=None: DDIMLoop(eng, tuple(args.latent), B, steps=args.custom_steps, eta=args.ddim_eta, scale=args.scale,
                                  context_shape=(1, ctx_dim), device=dev, capture_stream=cs)
    if args.inflight > 1:
        # independent batches in flight on alternating streams (edadm.sampling.InFlightSampler): +13..17 % images / s on one MI355X
        from edadm.sampling import InFlightSampler
        loop = InFlightSampler(mk, n=args.inflight, device=dev)
    else:
        loop = mk()
    dec = None
    if not args.no_decode:
        from edadm.nets.vq_decoder import Decoder
        from edadm.decoder import DecoderEngine
        torch.manual_seed(args.seed + 1)
        d = Decoder(**VQF4).to(dev).eval()
        dec = DecoderEngine(d, torch.nn.Conv2d(3, 3, 1).to(dev), codebook=torch.randn(8192, 3, device=dev))
    drv = ShardedSampler(loop, args.seed, args.n_samples, B, tuple(args.latent), n_classes=1000, device=dev)
    uncond_label = torch.full((B,), 1000, device=dev)

    def cond_fn(i, labels):
        with torch.no_grad():
            return (ld.get_learned_conditioning({ld.cond_stage_key: labels}).contiguous(),
                    ld.get_learned_conditioning({ld.cond_stage_key: uncond_label}).contiguous())

    done = {"images": 0}
    if args.save:
        os.makedirs(args.save, exist_ok=True)

    def sink(i, latents):
        img = latents
        if dec is not None:
            img = torch.clamp((dec(latents) + 1.0) / 2.0, min=0.0, max=1.0)
        done["images"] += img.shape[0]
        if args.save:
            np.save(os.path.join(args.save, "batch_%06d.npy" % i), img.cpu().numpy())

    torch.cuda.synchronize()
    t0 = time.time()
    n = drv.run(cond_fn, sink, limit=args.max_batches)
    torch.cuda.synchronize()
    dt = time.time() - t0
    tot = torch.tensor([float(done["images"]), dt], device=dev)
    if world > 1:
        cnt = tot[:1].clone()
        torch.distributed.all_reduce(cnt)                               # the only collective: a counter
        mx = tot[1:].clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        tot = torch.cat([cnt, mx])
    if rank == 0:
        print(json.dumps({"job": "sample", "ranks": world, "images": int(tot[0].item()), "seconds": float(tot[1].item()),
                          "images_per_sec": float(tot[0].item() / max(tot[1].item(), 1e-9)), "batches_this_rank": n,
                          "decode": dec is not None}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main(argv=None):
    args = parser().parse_args(argv)
    (calibrate if args.job == "calibrate" else sample)(args)


if __name__ == "__main__":
    main()
