"""Conditional activation caching — qdiff_control/data_utils.py:7-80 of the reference: each calibration
batch (x, t, index, cond, uncond) is doubled for classifier-free guidance before the prefix forwards
(x_in = [x, x], t_in = [t, t], c_in = [uncond, cond], :28-31), so the cache holds 2 x N rows."""
import torch

from qdiff.data_utils import save_inp_oup_data as _save, GetLayerInpOut, DataSaverHook, StopForwardException  # noqa: F401


def cfg_double(batch):
    return [torch.cat([batch[0]] * 2), torch.cat([batch[1]] * 2), torch.cat([batch[4], batch[3]])]


def save_inp_oup_data(model, layer, cali_data, asym=False, act_quant=False, batch_size=32, input_prob=False,
                      keep_gpu=True):
    return _save(model, layer, cali_data, asym, act_quant, batch_size=batch_size, input_prob=input_prob,
                 keep_gpu=keep_gpu, batch_transform=cfg_double)
