"""block_reconstruction — signature of the reference's qdiff/block_recon.py:13-18; the loop itself
is edadm/recon.py (HIP kernels K1, K2, K7, K8, K10)."""
from edadm.recon import reconstruct, LossFunction, LinearTempDecay  # noqa: F401


def block_reconstruction(model, block, cali_data, batch_size: int = 32, iters: int = 20000, weight: float = 0.01,
                         opt_mode: str = 'mse', asym: bool = False, b_range: tuple = (20, 2), warmup: float = 0.0,
                         act_quant: bool = False, lr_a: float = 4e-5, lr_w=1e-2, p: float = 2.0,
                         input_prob: float = 1.0, keep_gpu: bool = True, recon_w: bool = False, recon_a: bool = False,
                         add_loss: float = 0.0):
    # the reference caches with a hard-coded batch of 32 (block_recon.py:126)
    reconstruct(model, block, cali_data, is_block=True, batch_size=batch_size, iters=iters, weight=weight,
                opt_mode=opt_mode, asym=asym, b_range=b_range, warmup=warmup, act_quant=act_quant, lr_a=lr_a,
                lr_w=lr_w, p=p, input_prob=input_prob, keep_gpu=keep_gpu, recon_w=recon_w, recon_a=recon_a,
                add_loss=add_loss, cache_batch=32)
