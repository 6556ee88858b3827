"""Task harness for the unconditional LDM-4 on LSUN-Bedroom 256x256 -- the flow of the reference's
scripts/sample_diffusion_ldm_bedroom.py:257-316 (launcher scripts/for_bedroom.sh, parser scripts/task_config.py:41-75), which is the
Church script with TDAC_bedroom_calib_data_generator (scripts/calibration.py:156-262: the `> 0` fix-up of the rounding remainder),
the UNet of models/ldm/lsun_beds256/config.yaml (224 channels, multipliers 1-2-3-4, 32-channel heads at ds 2 / 4 / 8, 64 x 64 x 3
latents, linear_end 0.0195), 200 DDIM steps at eta 1 and the W4A8 launcher's lr_w 1e-2 / lr_a 5e-3 / add_loss 1.0:

    python -m scripts.sample_diffusion_ldm_bedroom calibrate --out calib_bedroom/
    python -m torch.distributed.run --nproc-per-node 8 -m scripts.sample_diffusion_ldm_bedroom sample --state calib_bedroom/ --n_samples 50000

Everything else -- scale initialisation through DDIMSampler.sample(quant_unet=True), Change_LDM_model_attnblock, the unconditional
recon_block_Qmodel walk, state + frozen model, rank-sharded sampling on the int8 executor -- is scripts/sample_diffusion_ldm_church.py."""
from scripts import sample_diffusion_ldm_church as _church

BEDROOM = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=[8, 4, 2], num_res_blocks=2,
               channel_mult=[1, 2, 3, 4], num_head_channels=32)
TASK = dict(unet=BEDROOM, linear_start=0.0015, linear_end=0.0195, tdac="TDAC_bedroom_calib_data_generator")


def parser():
    ap = _church.parser()
    ap.set_defaults(custom_steps=200, eta=1.0, out="calib_bedroom", state="calib_bedroom", lr_w=1e-2, lr_a=5e-3, add_loss=1.0,
                    batch_size=50, lamda=1.0)
    return ap


def main(argv=None):
    _church.main(argv, task=TASK, make_parser=parser)


if __name__ == "__main__":
    main()
