"""qdiff — the reference's PTQ engine API (qdiff/__init__.py:1-7) on the MI355X-native kernels."""
from qdiff.quant_block import BaseQuantBlock
from qdiff.quant_layer import QuantModule
from qdiff.quant_model import QuantModel
from qdiff.set_quantize_params import set_weight_quantize_params, set_act_quantize_params
from qdiff.recon_block_Qmodel import recon_block_Qmodel, Change_LDM_model_attnblock
from qdiff.recon_layer_Qmodel import recon_layer_Qmodel
from qdiff.set_quantize_params_LDM import set_weight_quantize_params_LDM, set_act_quantize_params_LDM
