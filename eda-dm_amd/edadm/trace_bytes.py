"""ALGORITHMIC bytes of the HBM-bound entry points, accounted at the C-ABI boundary (edadm/lib.py::call) -- SURVEY.md section 8(d):
4 B x (elements read + written once) per kernel launch (1 B for an int8 operand, 2 B for an f16 one).  Diagnostic only: switched on by
the environment variable EDADM_TRACE_BYTES=<path> (tools/prof_elementwise.sh), it adds the bytes of every call of the listed entry
points over the life of the process and writes {entry point: {calls, bytes}} to <path> at exit; tools/elementwise_hbm.py joins that
with the rocprofv3 kernel durations and FETCH_SIZE / WRITE_SIZE counters of the same command.  Never imported by the product path."""
import atexit
import ctypes
import json


def _v(a):
    if isinstance(a, ctypes.c_void_p):
        return a.value or 0
    if a is None:
        return 0
    if hasattr(a, "value"):
        return a.value
    return a


def _has(a):
    return 1 if _v(a) else 0


# entry point -> bytes(args).  Argument positions follow include/edadm.h.
FORMULAS = {
    # K1 (quant_layer.py:266-276)
    "edadm_fake_quant_fwd": lambda a: 4 * _v(a[3]) * (2 + _has(a[2])),
    "edadm_fake_quant_bwd": lambda a: 4 * _v(a[4]) * (2 + _has(a[2])),
    # K2 (adaptive_rounding.py:49-61)
    "edadm_adaround_fwd": lambda a: 12 * _v(a[5]) * _v(a[6]),
    "edadm_adaround_bwd": lambda a: 16 * _v(a[6]) * _v(a[7]),
    # K5 (GroupNorm + SiLU / LayerNorm + the consumers' quantisers)
    "edadm_groupnorm_stats": lambda a: 4 * _v(a[3]) * _v(a[4]) * _v(a[5]),
    "edadm_groupnorm_stats_cat": lambda a: 4 * _v(a[6]) * _v(a[7]) * (_v(a[1]) + _v(a[3]) * _has(a[2])),
    # *_rep: x2 holds B2 images read periodically -- each of its bytes counted once
    "edadm_groupnorm_stats_cat_rep": lambda a: 4 * _v(a[7]) * (_v(a[6]) * _v(a[1]) + (_v(a[10]) or _v(a[6])) * _v(a[3]) * _has(a[2])),
    # (edadm_groupnorm_final_cat*: reduce the producers' partial sums -- kilobytes; their launches are listed with the statistics group)
    "edadm_groupnorm_apply": lambda a: _v(a[5]) * _v(a[6]) * _v(a[7]) * (4 + 4 * _has(a[10]) + _has(a[11]) + _has(a[12]) + _has(a[13])),
    "edadm_groupnorm_apply_cat": lambda a: _v(a[8]) * _v(a[9]) * (_v(a[1]) + _v(a[3])) * (4 + 4 * _has(a[12]) + _has(a[13]) + _has(a[14]) + _has(a[15])),
    # ... _raw: one more int8 output (the un-normalised input quantised for the skip convolution); x2 may hold B2 < B images
    "edadm_groupnorm_apply_cat_raw": lambda a: _v(a[8]) * _v(a[9]) * (_v(a[1]) + _v(a[3])) * (4 + 4 * _has(a[12]) + _has(a[13]) + _has(a[14]) + _has(a[15]) + _has(a[18])),
    "edadm_quant_i8_cat": lambda a: 5 * _v(a[5]) * (_v(a[1]) + _v(a[3]) * _has(a[2])),
    "edadm_quant_i8_cat_rep": lambda a: 5 * _v(a[5]) * (_v(a[1]) + _v(a[3]) * _has(a[2])),
    "edadm_layernorm_quant": lambda a: _v(a[3]) * _v(a[4]) * (4 + 4 * _has(a[6]) + _has(a[7]) + _has(a[8]) + _has(a[9])),
    "edadm_layernorm_quant_radd": lambda a: _v(a[8]) * (4 * _v(a[1]) + _v(a[7]) * (4 * _has(a[4]) + _has(a[10]) + _has(a[11]) + _has(a[12]))),
    "edadm_quant_i8": lambda a: 5 * _v(a[2]) * _v(a[3]),
    # K7 (quant_layer.py:26-33; block_recon.py:186-189)
    "edadm_lp_loss_fwd": lambda a: 8 * _v(a[2]),
    "edadm_lp_loss_bwd": lambda a: 12 * _v(a[2]),
    "edadm_lp_loss_inject": lambda a: 4 * _v(a[7]) * (2 * _v(a[4]) + 2 * _v(a[6])),
    # K8, K9, K10
    "edadm_adam_step": lambda a: 28 * _v(a[4]),
    "edadm_ddim_step": lambda a: 4 * _v(a[8]) * _v(a[9]) * (4 + _has(a[5]) + _has(a[7])),
    "edadm_mix_where": lambda a: 4 * _v(a[3]) * (3 + _has(a[4])),
    # K12: the non-contraction operators of a reconstruction iteration
    "edadm_gn_fwd_nhwc": lambda a: 8 * _v(a[6]) * _v(a[7]) * _v(a[8]),
    "edadm_gn_bwd_nhwc": lambda a: 12 * _v(a[7]) * _v(a[8]) * _v(a[9]),
    "edadm_ln_fwd": lambda a: 8 * _v(a[5]) * _v(a[6]),
    "edadm_ln_bwd": lambda a: 12 * _v(a[5]) * _v(a[6]),
    "edadm_geglu_fwd": lambda a: 12 * _v(a[2]) * _v(a[3]),
    "edadm_geglu_bwd": lambda a: 20 * _v(a[3]) * _v(a[4]),
    "edadm_silu_bwd": lambda a: 12 * _v(a[3]),
    "edadm_softmax_bwd": lambda a: 12 * _v(a[3]) * _v(a[4]),
    "edadm_softmax_fwd_any": lambda a: 8 * _v(a[2]) * _v(a[3]),
    # K11's re-formatting passes (operand expansion of the three-product contraction)
    "edadm_absmax_parts": lambda a: 4 * _v(a[1]),
    "edadm_split_f16": lambda a: 8 * _v(a[1]) * _v(a[2]) * _v(a[3]),
    # in [R][C] fp32 -> [C][R] as f16 (hi, lo) pairs; with a convolution geometry the [R][C] matrix is the im2col of an NHWC tensor gathered on
    # the fly (3x3 in every such layer of the path): the tensor is read once, the nine-fold matrix written
    "edadm_transpose_split_f16": lambda a: (4 * _v(a[1]) * _v(a[2]) // 9 + 4 * _v(a[1]) * _v(a[2])) if _has(a[5]) else 8 * _v(a[1]) * _v(a[2]),
}

TOTALS = {}


def hook(name, args):
    f = FORMULAS.get(name)
    if f is None:
        return
    t = TOTALS.setdefault(name, [0, 0])
    t[0] += 1
    t[1] += int(f(args))


def install(path):
    from . import lib
    lib.TRACE_HOOK = hook

    def dump():
        with open(path, "w") as fh:
            json.dump({k: {"calls": v[0], "bytes": v[1]} for k, v in sorted(TOTALS.items())}, fh, indent=1)
    atexit.register(dump)
