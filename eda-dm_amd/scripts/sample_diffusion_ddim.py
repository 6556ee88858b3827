"""Task harness for the DDPM UNet (CIFAR-10 32x32, BASELINE configs 1 and 2) -- the flow of the reference's
scripts/sample_diffusion_ddim.py:265-323 (launcher scripts/for_cifar.sh) over this build's API, as two jobs:

    calibrate   TDAC_cifar_calib_data_generator -> set_{weight,act}_quantize_params -> recon_block_Qmodel (--block_recon) or
                recon_layer_Qmodel (--layer_recon) -> quantiser state + frozen integer model in --out
    sample      frozen model on every rank, batches {i : i mod world = rank}, `--timesteps` DDIM steps (quad / uniform skip, eta)
                on the int8 executor replayed from a HIP graph

    python -m scripts.sample_diffusion_ddim calibrate --out calib_cifar/ [--weight_bit 4 --calib_num_samples 1024 --batch_samples 1024]
    python -m scripts.sample_diffusion_ddim sample --state calib_cifar/ --max_images 50000

No checkpoint ships with the reference tree: the UNet of configs/cifar10.yml (or --model JSON) gets seeded random weights."""
import argparse
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

CIFAR10 = dict(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               resamp_with_conv=True, image_size=32)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("job", choices=["calibrate", "sample"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--model", type=json.loads, default=None, help="model section of the config as JSON (default: configs/cifar10.yml)")
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--skip_type", default="quad", choices=["quad", "uniform"])
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--weight_bit", type=int, default=4)
    ap.add_argument("--act_bit", type=int, default=8)
    ap.add_argument("--split", action="store_true", default=True)
    ap.add_argument("--out", default="calib_cifar")
    ap.add_argument("--calib_num_samples", type=int, default=1024)
    ap.add_argument("--batch_samples", type=int, default=1024)
    ap.add_argument("--lamda", type=float, default=1.2)
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--lr_a", type=float, default=5e-4)
    ap.add_argument("--lr_w", type=float, default=5e-1)
    ap.add_argument("--add_loss", type=float, default=0.8)
    ap.add_argument("--layer_recon", action="store_true", help="recon_layer_Qmodel instead of recon_block_Qmodel")
    ap.add_argument("--no_recon", action="store_true")
    ap.add_argument("--state", default="calib_cifar")
    ap.add_argument("--max_images", type=int, default=50000)
    ap.add_argument("--n_batch", type=int, default=500)
    ap.add_argument("--max_batches", type=int, default=None)
    ap.add_argument("--save", default=None)
    return ap


def seq_of(args):
    if args.skip_type == "uniform":
        return list(range(0, 1000, 1000 // args.timesteps))
    return [int(s) for s in (np.linspace(0, np.sqrt(1000 * 0.8), args.timesteps) ** 2)]      # sample_diffusion_ddim.py:125-133


def build(args, dev):
    from edadm.nets.ddpm_unet import Model
    from qdiff import QuantModel
    from qdiff.utils import seed_everything
    seed_everything(args.seed)
    m = dict(args.model or CIFAR10)
    size = m.pop("image_size", 32)
    cfg = SimpleNamespace(model=SimpleNamespace(**m), data=SimpleNamespace(image_size=size),
                          diffusion=SimpleNamespace(num_diffusion_timesteps=1000))
    model = Model(cfg).to(dev).eval()
    wq = {'n_bits': args.weight_bit, 'symmetric': True, 'channel_wise': True, 'scale_method': 'mse'}
    aq = {'n_bits': args.act_bit, 'symmetric': True, 'channel_wise': False, 'scale_method': 'mse', 'leaf_param': True, 'prob': 0.5}
    qnn = QuantModel(model, wq, aq, sm_abit=8).to(dev).eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(False, False)
    return qnn, cfg, size


def calibrate(args):
    from edadm import harness as H
    from scripts.calibration import TDAC_cifar_calib_data_generator
    from qdiff import set_weight_quantize_params, set_act_quantize_params, recon_block_Qmodel, recon_layer_Qmodel
    world, rank, dev = H.init_dist()
    qnn, cfg, size = build(args, dev)
    diffusion = SimpleNamespace(seq=seq_of(args), betas=torch.linspace(1e-4, 2e-2, 1000).to(dev), args=SimpleNamespace(eta=args.eta))
    t0 = H.now()
    cali = TDAC_cifar_calib_data_generator(qnn.model, cfg, args.lamda, args.calib_num_samples, args.batch_samples, dev, diffusion, False)
    t1 = H.now()
    if args.split:
        qnn.model.config.split_shortcut = True
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali)
    t2 = H.now()
    if not args.no_recon:
        kwargs = dict(cali_data=cali, iters=args.iters, act_quant=True, asym=True, opt_mode='mse', lr_a=args.lr_a, lr_w=args.lr_w, p=2.0,
                      weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=args.add_loss, recon_w=True,
                      recon_a=True)
        qnn.set_quant_state(True, True)
        qnn = (recon_layer_Qmodel if args.layer_recon else recon_block_Qmodel)(args, qnn, cali, kwargs).recon()
    qnn.set_quant_state(True, True)
    t3 = H.now()
    if rank == 0:
        H.save_calibrated(qnn, args.out, {"tdac_s": t1 - t0, "scale_init_s": t2 - t1, "reconstruction_s": t3 - t2})


def sample(args):
    from edadm import harness as H
    from edadm.sampling import GraphedUNet
    from ddim.functions.denoising import generalized_steps
    world, rank, dev = H.init_dist()
    qnn, cfg, size = build(args, dev)
    if args.split:
        qnn.model.config.split_shortcut = True
    C = cfg.model.in_channels
    eng = H.load_calibrated(qnn, args.state, lambda: qnn(torch.zeros(2, C, size, size, device=dev), torch.zeros(2, device=dev)))
    B, seq = args.n_batch, seq_of(args)
    betas = torch.linspace(1e-4, 2e-2, 1000).to(dev)
    unet = GraphedUNet(eng, torch.zeros(B, C, size, size, device=dev), torch.zeros(B, device=dev), None)

    def batch(i, gen):
        x = torch.randn(B, C, size, size, generator=gen, device=dev)
        with torch.no_grad():
            xs, _ = generalized_steps(x, seq, lambda xt, t: unet(xt, t), betas, eta=args.eta)
        return torch.clamp((xs[-1] + 1.0) / 2.0, 0.0, 1.0)                    # inverse_data_transform (rescaled)

    H.run_sharded(batch, args.max_images, B, args.seed, save=args.save, max_batches=args.max_batches, extra={"steps": len(seq)})


def main(argv=None):
    args = parser().parse_args(argv)
    (calibrate if args.job == "calibrate" else sample)(args)


if __name__ == "__main__":
    main()
