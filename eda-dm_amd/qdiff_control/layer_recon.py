"""layer_reconstruction for conditional models — qdiff_control/layer_recon.py:13-109 (no split handling)."""
from edadm.recon import reconstruct, LossFunction, LinearTempDecay  # noqa: F401
from qdiff_control.data_utils import save_inp_oup_data


def layer_reconstruction(model, layer, cali_data, batch_size: int = 32, iters: int = 20000, weight: float = 0.001,
                         opt_mode: str = 'mse', asym: bool = False, b_range: tuple = (20, 2), warmup: float = 0.0,
                         act_quant: bool = False, lr_a: float = 4e-5, lr_w=1e-2, p: float = 2.0,
                         input_prob: float = 1.0, keep_gpu: bool = True, recon_w: bool = False, recon_a: bool = False,
                         add_loss: float = 0.0):
    reconstruct(model, layer, cali_data, is_block=False, batch_size=batch_size, iters=iters, weight=weight,
                opt_mode=opt_mode, asym=asym, b_range=b_range, warmup=warmup, act_quant=act_quant, lr_a=lr_a,
                lr_w=lr_w, p=p, input_prob=input_prob, keep_gpu=keep_gpu, recon_w=recon_w, recon_a=recon_a,
                add_loss=add_loss, cache_batch=batch_size, control=True, save_fn=save_inp_oup_data)
