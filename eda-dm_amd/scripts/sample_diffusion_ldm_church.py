"""Task harness for the unconditional LDM-8 (LSUN-Church 256x256, BASELINE config 3) -- the flow of the reference's
scripts/sample_diffusion_ldm_church.py:256-311 (launcher scripts/for_church.sh) over this build's API, as two jobs:

    calibrate   TDAC_church_calib_data_generator -> set_{weight,act}_quantize_params_LDM (through DDIMSampler.sample(quant_unet=True))
                -> Change_LDM_model_attnblock -> recon_block_Qmodel (unconditional walk) -> quantiser state + frozen model in --out
    sample      frozen model on every rank, batches {i : i mod world = rank}, `--custom_steps` DDIM steps (eta) on the int8 executor

    python -m scripts.sample_diffusion_ldm_church calibrate --out calib_church/
    python -m torch.distributed.run --nproc-per-node 8 -m scripts.sample_diffusion_ldm_church sample --state calib_church/ --n_samples 50000

Synthetic weights (no checkpoint in the tree): the UNet of models/ldm/lsun_churches256/config.yaml (or --unet JSON).  The KL-f8
first stage is not attached (SURVEY 8f-3 covers the VQ-f4 decoder of the headline config): latents are the output."""
import argparse
import json
import os
import sys

import torch

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

CHURCH = dict(image_size=32, in_channels=4, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8], num_res_blocks=2,
              channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True, resblock_updown=True)
# what distinguishes the unconditional LDM tasks (scripts/sample_diffusion_ldm_bedroom.py reuses this module with its own table)
TASK = dict(unet=CHURCH, linear_start=0.0015, linear_end=0.0155, tdac="TDAC_church_calib_data_generator")


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("job", choices=["calibrate", "sample"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--unet", type=json.loads, default=None)
    ap.add_argument("--custom_steps", type=int, default=500)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--weight_bit", type=int, default=4)
    ap.add_argument("--act_bit", type=int, default=8)
    ap.add_argument("--sm_abit", type=int, default=8)
    ap.add_argument("--split", action="store_true", default=True)
    ap.add_argument("--out", default="calib_church")
    ap.add_argument("--calib_num_samples", type=int, default=1024)
    ap.add_argument("--batch_samples", type=int, default=64)
    ap.add_argument("--lamda", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--lr_a", type=float, default=1e-4)
    ap.add_argument("--lr_w", type=float, default=5e-2)
    ap.add_argument("--add_loss", type=float, default=1.0)
    ap.add_argument("--no_recon", action="store_true")
    ap.add_argument("--state", default="calib_church")
    ap.add_argument("--n_samples", type=int, default=50000)
    ap.add_argument("--batch_size", type=int, default=100)
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
    kw = args.unet or TASK["unet"]
    unet = UNetModel(**kw)
    H.reinit_zero_modules(unet, args.seed)
    ld = LatentDiffusionLite(unet, timesteps=1000, linear_start=TASK["linear_start"], linear_end=TASK["linear_end"],
                             conditioning_key=None).to(dev).eval()
    wq = {'n_bits': args.weight_bit, 'symmetric': True, 'channel_wise': True, 'scale_method': 'mse'}
    aq = {'n_bits': args.act_bit, 'symmetric': True, 'channel_wise': False, 'scale_method': 'mse', 'leaf_param': True, 'prob': 0.5}
    qnn = QuantModel(model=ld.model.diffusion_model, weight_quant_params=wq, act_quant_params=aq, sm_abit=args.sm_abit).to(dev).eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(False, False)
    ld.model.diffusion_model = qnn
    return ld, qnn, kw, aq


def calibrate(args):
    from edadm import harness as H
    from scripts import calibration as tdac_mod
    from qdiff import set_weight_quantize_params_LDM, set_act_quantize_params_LDM, Change_LDM_model_attnblock, recon_block_Qmodel
    world, rank, dev = H.init_dist()
    ld, qnn, kw, aq = build(args, dev)
    t0 = H.now()
    cali = getattr(tdac_mod, TASK["tdac"])(ld, args, args.calib_num_samples, args.batch_samples, dev, args.custom_steps)
    t1 = H.now()
    if args.split:
        qnn.model.split_shortcut = True
    set_weight_quantize_params_LDM(ld, cali, args)
    set_act_quantize_params_LDM(ld, cali, args)
    t2 = H.now()
    if not args.no_recon:
        Change_LDM_model_attnblock(qnn, aq)
        kwargs = dict(cali_data=cali[:-1], iters=args.iters, act_quant=True, asym=True, opt_mode='mse', lr_a=args.lr_a, lr_w=args.lr_w,
                      p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=args.add_loss,
                      recon_w=True, recon_a=True, keep_gpu=False)
        qnn.set_quant_state(True, True)
        ld.model.diffusion_model = recon_block_Qmodel(args, qnn, cali, kwargs).recon()
    qnn.set_quant_state(True, True)
    t3 = H.now()
    if rank == 0:
        H.save_calibrated(qnn, args.out, {"tdac_s": t1 - t0, "scale_init_s": t2 - t1, "reconstruction_s": t3 - t2,
                                         "attention_blocks_wrapped": not args.no_recon})


def sample(args):
    from edadm import harness as H
    from edadm.sampling import DDIMLoop
    from qdiff import Change_LDM_model_attnblock
    world, rank, dev = H.init_dist()
    ld, qnn, kw, aq = build(args, dev)
    if args.split:
        qnn.model.split_shortcut = True
    C, S = kw["in_channels"], kw["image_size"]
    with open(os.path.join(args.state, "meta.json")) as fh:
        if json.load(fh).get("attention_blocks_wrapped"):
            Change_LDM_model_attnblock(qnn, aq)                 # module paths of the saved state are those after the wrap
    eng = H.load_calibrated(qnn, args.state, lambda: qnn(torch.zeros(2, C, S, S, device=dev), torch.zeros(2, dtype=torch.long, device=dev)))
    B = args.batch_size
    loop = DDIMLoop(eng, (C, S, S), B, steps=args.custom_steps, eta=args.eta, scale=1.0, linear_start=TASK["linear_start"],
                    linear_end=TASK["linear_end"], context_shape=None, device=dev)

    def batch(i, gen):
        return loop.sample(torch.randn(B, C, S, S, generator=gen, device=dev))

    H.run_sharded(batch, args.n_samples, B, args.seed, save=args.save, max_batches=args.max_batches, extra={"steps": args.custom_steps})


def main(argv=None, task=None, make_parser=None):
    global TASK
    saved = TASK
    if task is not None:
        TASK = task
    try:
        _main((make_parser or parser)().parse_args(argv))
    finally:
        TASK = saved


def _main(args):
    if args.job == "calibrate":
        calibrate(args)
        if int(os.environ.get("RANK", "0")) == 0:
            with open(os.path.join(args.out, "meta.json"), "w") as fh:
                json.dump({"attention_blocks_wrapped": not args.no_recon}, fh)
    else:
        sample(args)


if __name__ == "__main__":
    main()
