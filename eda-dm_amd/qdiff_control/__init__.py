"""qdiff_control — the conditional-model twins of the PTQ engine (reference qdiff_control/__init__.py:1-4).
`get_prompts` / `center_resize_image` (COCO dataset preparation, pycocotools / skimage) are out of scope."""
from qdiff_control.set_quantize_params_Stable import set_weight_quantize_params_Stable, set_act_quantize_params_Stable
from qdiff_control.recon_block_Qmodel import recon_block_Qmodel
from qdiff_control.set_quantize_params_Conditional import (set_weight_quantize_params_Conditional,
                                                            set_act_quantize_params_Conditional)
