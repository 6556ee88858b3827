"""AttnBlock_layer_reconstruction — qdiff/attn_layer_recon.py:13-133 of the reference: tune only
the q/k/v/w activation step sizes of a QuantAttnBlock against the block's FP output."""
import random

import torch

from edadm import ops
from edadm.recon import FusedAdam, LossFunction, _as_param
from qdiff.quant_block import QuantAttnBlock
from qdiff.quant_layer import _mask_rng
from qdiff.data_utils import save_inp_oup_data


def AttnBlock_layer_reconstruction(model, block, cali_data, batch_size: int = 32, iters: int = 20000,
                                   weight: float = 0.01, opt_mode: str = 'mse', asym: bool = False,
                                   b_range: tuple = (20, 2), warmup: float = 0.0, act_quant: bool = False,
                                   lr_a: float = 4e-5, lr_w=1e-2, p: float = 2.0, input_prob: float = 1.0,
                                   keep_gpu: bool = True, recon_w: bool = False, recon_a: bool = False,
                                   add_loss: float = 0.0, layer_loss: bool = False):
    block.set_quant_state(True, act_quant)
    a_para, aqs = [], []
    if act_quant:
        for m in block.modules():                               # attn_layer_recon.py:47-63: QuantAttnBlock only
            if not isinstance(m, QuantAttnBlock):
                continue
            for q in (m.act_quantizer_q, m.act_quantizer_k, m.act_quantizer_v, m.act_quantizer_w):
                _as_param(q)
                if recon_a:
                    a_para.append(q.delta)
                    q.is_training = True
                    aqs.append(q)
    a_opt = FusedAdam(a_para, lr_a, iters) if a_para else None
    loss_func = LossFunction(block, round_loss='none', weight=weight, max_count=iters, rec_loss=opt_mode,
                             b_range=b_range, decay_start=0, warmup=warmup, p=p)
    _, cached_inps, cached_outs = save_inp_oup_data(model, block, cali_data, asym, act_quant, batch_size=32,
                                                    input_prob=True, keep_gpu=keep_gpu, final=False)
    sz = cached_outs.size(0)
    model.block_count = model.block_count + 1
    model.engine = None          # a frozen executor holds the old step sizes: freeze() again after calibration
    for _ in range(iters):
        idx = torch.tensor(random.sample(range(sz), batch_size), device=cached_outs.device)
        cur_out, cur_inp, cur_sym = cached_outs[idx], cached_inps[0][idx], cached_inps[1][idx]
        if input_prob < 1.0:
            cur_inp = ops.mix_where(cur_inp.contiguous(), cur_sym.contiguous(), input_prob,
                                    seed=_mask_rng.getrandbits(62))
        else:
            cur_inp = cur_sym                                   # attn_layer_recon.py:101-102
        if a_opt:
            a_opt.zero_grad()
        loss = loss_func(block(cur_inp), cur_out)
        if loss.requires_grad:
            loss.backward()
        if a_opt:
            a_opt.step()
    for q in aqs:
        q.is_training = False
