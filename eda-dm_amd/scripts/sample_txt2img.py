"""Task harness for Stable Diffusion v1 (512x512 text-to-image, BASELINE config 5) -- the flow of the reference's
scripts/sample_txt2img.py:154-283 (launcher scripts/for_coco.sh) over this build's API, as two jobs:

    calibrate   TDAC_coco_calib_data_generator (PLMS trajectories, classifier-free guidance) -> set_{weight,act}_quantize_params_Stable
                -> conditional recon_block_Qmodel (qdiff_control, batch 2) -> quantiser state + frozen model in --out
    sample      frozen model on every rank, prompt batches {i : i mod world = rank}, `--custom_steps` PLMS steps x CFG `--scale`

    python -m scripts.sample_txt2img calibrate --out calib_sd/ --plms
    python -m torch.distributed.run --nproc-per-node 8 -m scripts.sample_txt2img sample --state calib_sd/ --n_samples 10000 --plms

Synthetic weights (no checkpoint in the tree).  The CLIP text encoder is third-party and out of scope: prompts map
to seeded [77, 768] embeddings through a lookup (`PromptTable`), the empty prompt to its own row.  The KL-f8 first stage is not
attached: latents are the output."""
import argparse
import json
import os
import sys
import zlib

import torch

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

SD = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
          channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
          use_checkpoint=True, legacy=False)


class PromptTable(torch.nn.Module):
    """stand-in for FrozenCLIPEmbedder (ldm/modules/encoders/modules.py:137-167): prompt -> [tokens, dim], a pure function of the text"""

    def __init__(self, tokens, dim, seed):
        super().__init__()
        self.tokens, self.dim, self.seed = tokens, dim, seed
        self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, prompts):
        rows = []
        for p in prompts:
            g = torch.Generator().manual_seed((self.seed * 1000003 + zlib.crc32(p.encode())) % (2 ** 31))
            rows.append(torch.randn(self.tokens, self.dim, generator=g))
        return torch.stack(rows).to(self.anchor.device)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("job", choices=["calibrate", "sample"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--unet", type=json.loads, default=None)
    ap.add_argument("--prompt", default="a puppy wearing a hat")
    ap.add_argument("--plms", action="store_true", default=True)
    ap.add_argument("--custom_steps", type=int, default=50)
    ap.add_argument("--ddim_eta", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=7.5)
    ap.add_argument("--C", type=int, default=4)
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--f", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--weight_bit", type=int, default=4)
    ap.add_argument("--act_bit", type=int, default=8)
    ap.add_argument("--sm_abit", type=int, default=8)
    ap.add_argument("--no_grad_ckpt", action="store_true", default=True)
    ap.add_argument("--split", action="store_true", default=True)
    ap.add_argument("--out", default="calib_sd")
    ap.add_argument("--calib_num_samples", type=int, default=256)
    ap.add_argument("--batch_samples", type=int, default=8)
    ap.add_argument("--lamda", type=float, default=5.0)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr_a", type=float, default=1e-4)
    ap.add_argument("--lr_w", type=float, default=3e-2)
    ap.add_argument("--add_loss", type=float, default=0.8)
    ap.add_argument("--no_recon", action="store_true")
    ap.add_argument("--state", default="calib_sd")
    ap.add_argument("--n_samples", type=int, default=10000)
    ap.add_argument("--n_batch", type=int, default=4)
    ap.add_argument("--max_batches", type=int, default=None)
    ap.add_argument("--save", default=None)
    return ap


def build(args, dev):
    from edadm.nets.ldm_unet import UNetModel
    from edadm.latent import LatentDiffusionLite
    from edadm import harness as H
    from qdiff import QuantModel
    from qdiff.utils import seed_everything
    seed_everything(args.seed)
    kw = args.unet or SD
    unet = UNetModel(**kw)
    H.reinit_zero_modules(unet, args.seed)
    ld = LatentDiffusionLite(unet, timesteps=1000, linear_start=0.00085, linear_end=0.012, conditioning_key="crossattn",
                             cond_stage_model=PromptTable(args.tokens, kw["context_dim"], args.seed)).to(dev).eval()
    wq = {'n_bits': args.weight_bit, 'symmetric': True, 'channel_wise': True, 'scale_method': 'mse'}
    aq = {'n_bits': args.act_bit, 'symmetric': True, 'channel_wise': False, 'scale_method': 'mse', 'leaf_param': True, 'prob': 0.5}
    qnn = QuantModel(model=ld.model.diffusion_model, weight_quant_params=wq, act_quant_params=aq, act_quant_mode="qdiff",
                     sm_abit=args.sm_abit).to(dev).eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    if args.no_grad_ckpt:
        qnn.set_grad_ckpt(False)
    ld.model.diffusion_model = qnn
    return ld, qnn, kw


def calibrate(args):
    from edadm import harness as H
    from scripts.calibration import TDAC_coco_calib_data_generator
    from qdiff_control import set_weight_quantize_params_Stable, set_act_quantize_params_Stable, recon_block_Qmodel
    world, rank, dev = H.init_dist()
    ld, qnn, kw = build(args, dev)
    args.list_prompts = ["%s, variation %d" % (args.prompt, i) for i in range(args.calib_num_samples)]   # stands in for the COCO captions
    t0 = H.now()
    cali = TDAC_coco_calib_data_generator(ld, args, args.calib_num_samples, args.batch_samples, dev, args.custom_steps)
    t1 = H.now()
    if args.split:
        setattr(qnn, "split", True)                         # sample_txt2img.py:183-184: an attribute nothing reads
    set_weight_quantize_params_Stable(ld, cali, args)
    set_act_quantize_params_Stable(ld, cali, args)
    t2 = H.now()
    if not args.no_recon:
        kwargs = dict(cali_data=cali, iters=args.iters, act_quant=True, asym=True, opt_mode='mse', lr_a=args.lr_a, lr_w=args.lr_w, p=2.0,
                      weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=2, input_prob=0.5, add_loss=args.add_loss, recon_w=True,
                      recon_a=True, keep_gpu=False)
        qnn.set_quant_state(True, True)
        ld.model.diffusion_model = recon_block_Qmodel(args, qnn, cali, kwargs).recon()
    qnn.set_quant_state(True, True)
    t3 = H.now()
    if rank == 0:
        H.save_calibrated(qnn, args.out, {"tdac_s": t1 - t0, "scale_init_s": t2 - t1, "reconstruction_s": t3 - t2})


def sample(args):
    from edadm import harness as H
    from edadm.sampling import PLMSLoop, DDIMLoop
    world, rank, dev = H.init_dist()
    ld, qnn, kw = build(args, dev)
    C, Hh, Ww = args.C, args.H // args.f, args.W // args.f
    eng = H.load_calibrated(qnn, args.state, lambda: qnn(torch.zeros(2, C, Hh, Ww, device=dev), torch.zeros(2, dtype=torch.long, device=dev),
                                                         torch.zeros(2, args.tokens, kw["context_dim"], device=dev)))
    B = args.n_batch
    mk = PLMSLoop if args.plms else DDIMLoop
    loop = mk(eng, (C, Hh, Ww), B, steps=args.custom_steps, scale=args.scale, linear_start=0.00085, linear_end=0.012,
              context_shape=(args.tokens, kw["context_dim"]), device=dev)
    uc = ld.get_learned_conditioning(B * [""]).contiguous()

    def batch(i, gen):
        c = ld.get_learned_conditioning(["%s #%d" % (args.prompt, i * B + j) for j in range(B)]).contiguous()
        return loop.sample(torch.randn(B, C, Hh, Ww, generator=gen, device=dev), c, uc)

    H.run_sharded(batch, args.n_samples, B, args.seed, save=args.save, max_batches=args.max_batches,
                  extra={"steps": args.custom_steps, "sampler": "plms" if args.plms else "ddim", "scale": args.scale})


def main(argv=None):
    args = parser().parse_args(argv)
    (calibrate if args.job == "calibrate" else sample)(args)


if __name__ == "__main__":
    main()
