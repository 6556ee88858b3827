"""Activation caching for reconstruction — qdiff/data_utils.py:7-171 of the reference
(`save_inp_oup_data`, `GetLayerInpOut`, `DataSaverHook`, `StopForwardException`).

Per calibration batch: an FP forward up to the unit (its input = `cur_sym`, its output = target)
and a quantised-prefix forward up to the unit (its input = `cur_inp`).  Everything stays in HBM
(288 GB per MI355X: `keep_gpu` is accepted for source compatibility and ignored), and with
torch.distributed initialised the calibration batches are sharded across ranks and the slabs
all-gathered (edadm/dist.py)."""
import torch

from qdiff.quant_layer import QuantModule
from qdiff.quant_model import QuantModel
from qdiff.quant_block import BaseQuantBlock
from edadm import dist as edist


class StopForwardException(Exception):
    pass


class DataSaverHook:
    def __init__(self, store_input=False, store_output=False, stop_forward=False):
        self.store_input, self.store_output, self.stop_forward = store_input, store_output, stop_forward
        self.input_store = None
        self.output_store = None

    def __call__(self, module, input_batch, output_batch):
        if self.store_input:
            self.input_store = input_batch
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


class GetLayerInpOut:
    def __init__(self, model, layer, device, input_prob=False, act_quant=False, asym=False):
        self.model, self.layer, self.device = model, layer, device
        self.asym, self.act_quant, self.input_prob = asym, act_quant, input_prob
        self.data_saver = DataSaverHook(store_input=True, store_output=True, stop_forward=True)

    def _run(self, model_input):
        try:
            self.model(*[t.to(self.device) for t in model_input])
        except StopForwardException:
            pass

    @staticmethod
    def _pack(store):
        if len(store) == 1:
            return store[0].detach()
        return (store[0].detach(), store[1].detach())

    def __call__(self, model_input):
        self.model.eval()
        self.model.set_quant_state(False, False)
        handle = self.layer.register_forward_hook(self.data_saver)
        with torch.no_grad():
            # the engine must not intercept: these forwards need the hooks of the module graph
            eng, self.model.engine = getattr(self.model, "engine", None), None
            try:
                self._run(model_input)
                input_sym = self._pack(self.data_saver.input_store) if self.input_prob else None
                if self.asym:
                    self.data_saver.store_output = False
                    self.model.set_quant_state(weight_quant=True, act_quant=self.act_quant)
                    self._run(model_input)
                self.data_saver.store_output = True
            finally:
                self.model.engine = eng
        handle.remove()
        resblock = len(self.data_saver.input_store) != 1
        input_store = self._pack(self.data_saver.input_store)
        if self.input_prob:
            return resblock, input_store, self.data_saver.output_store.detach(), input_sym
        return resblock, input_store, self.data_saver.output_store.detach()


def save_inp_oup_data(model, layer, cali_data, asym=False, act_quant=False, batch_size=32, input_prob=False,
                      keep_gpu=True, batch_transform=None):
    """Returns (Resblock, cached_inps, cached_outs) with the reference's tuple nesting (:67-75).
    `batch_transform` maps a raw calibration batch to the model inputs (the conditional variant
    doubles the batch for classifier-free guidance, qdiff_control/data_utils.py:28-31)."""
    device = next(model.parameters()).device
    get = GetLayerInpOut(model, layer, device=device, asym=asym, input_prob=input_prob, act_quant=act_quant)
    n_batches = int(cali_data[0].size(0) / batch_size)
    mine = edist.shard_batches(n_batches)
    local = {}
    resblock = False
    for i in mine:
        batch = [c[i * batch_size:(i + 1) * batch_size] for c in cali_data]
        if batch_transform is not None:
            batch = batch_transform(batch)
        res = get(batch)
        resblock = res[0]
        local[i] = res[1:]
    if n_batches == 0:
        raise ValueError("fewer calibration samples than the caching batch size (%d)" % batch_size)

    def gather(select):
        return torch.cat(edist.all_gather_batches({i: select(v) for i, v in local.items()}, n_batches))

    if resblock:
        inps = [gather(lambda v: v[0][0]), gather(lambda v: v[0][1])]
    else:
        inps = gather(lambda v: v[0])
    outs = gather(lambda v: v[1])
    if input_prob:
        if resblock:
            syms = [gather(lambda v: v[2][0]), gather(lambda v: v[2][1])]
            return resblock, (inps, syms), outs
        return resblock, (inps, gather(lambda v: v[2])), outs
    if resblock:
        return resblock, (inps), outs
    return resblock, (inps,), outs
