"""Torch-tensor front end of the C ABI (include/edadm.h).  Tensors only lend their device
pointers; every launch goes to torch's current HIP stream so it orders with the surrounding
torch work and can be captured into a HIP graph.  No CPU fallback: non-device tensors raise."""
import ctypes

import torch

from . import lib

_ws = {}
_ws_retired = []


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise lib.EdadmError("edadm ops need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise lib.EdadmError("edadm ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise lib.EdadmError("expected %s, got %s" % (dtype, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def _pf(t):
    return _p(t, torch.float32)


def is_cl(x):
    """a dense 4-D tensor whose memory order is NHWC (torch channels_last) and not also NCHW"""
    return x.dim() == 4 and (not x.is_contiguous()) and x.is_contiguous(memory_format=torch.channels_last)


def mem_view(x):
    """(x as a contiguous tensor in its own memory order, whether that order is NHWC): elementwise kernels run on memory order,
    so a channels_last tensor needs no conversion -- the calibration graph keeps convolutional units in NHWC end to end.  A
    CHANNEL SLICE of a channels_last tensor (the two halves of a skip concatenation, each with its own quantiser) is NHWC-ordered
    too: it is gathered into a dense NHWC tensor (a strided row copy, no transposition)."""
    if x.is_contiguous():
        return x, False
    if x.dim() == 4 and x.stride(1) == 1 and x.shape[1] > 1:
        return x.permute(0, 2, 3, 1).contiguous(), True          # no copy when x is dense channels_last
    return x.contiguous(), False


def mem_like(t, cl):
    """a logical-NCHW tensor `t` laid out like mem_view's first result (cl: NHWC order)"""
    return t.permute(0, 2, 3, 1).contiguous() if cl else t.contiguous()


def mem_restore(y, cl):
    """the logical NCHW view of a result computed in NHWC memory order"""
    return y.permute(0, 3, 1, 2) if cl else y


_inited_devices = set()


def init_device(device):
    """Per-device constant tables of the library, filled eagerly (edadm_init_device) before any stream capture can exist."""
    if device.type != "cuda":
        raise lib.EdadmError("edadm ops need a device (no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _inited_devices:
        with torch.cuda.device(idx):
            lib.call("edadm_init_device")
        _inited_devices.add(idx)


_ws_scope = [None]
_scope_tokens = [0]


def new_scope_token():
    """A workspace-scope tag no other object of this process has had (an id() can be recycled once its owner is gone: a new owner
    would silently inherit buffers of a stale size on the same stream)."""
    _scope_tokens[0] += 1
    return "scope%d" % _scope_tokens[0]


def release_scope(tag):
    """Drop the workspaces of a scope whose owner (and its captured graphs, the only launches that knew those addresses) is gone."""
    for key in [k for k in _ws if k[3] == tag]:
        del _ws[key]


class workspace_scope:
    """Launches issued inside the scope take workspaces of their own (key extended by `tag`): a HIP graph captured inside keeps
    buffers no other graph or eager launch shares, whatever streams they were captured / are replayed on (torch hands out streams
    from a pool of 32 per device round-robin, so two streams created far apart can be the same one).  edadm.sampling.GraphedUNet
    captures inside a scope of its own: several of them can be replayed concurrently (InFlightSampler)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        self.prev, _ws_scope[0] = _ws_scope[0], self.tag

    def __exit__(self, *a):
        _ws_scope[0] = self.prev


def workspace(device, floats=None):
    init_device(device)
    n = int(lib.load().edadm_reduce_ws_floats()) if floats is None else int(floats)
    # one buffer per (device, stream[, scope]): launches on different streams may run concurrently (a decoder on a side stream next
    # to the sampling graph), and a graph captured on its capture stream keeps the buffer of that stream to itself
    key = (device.index, "r" if floats is None else "x", torch.cuda.current_stream(device).cuda_stream, _ws_scope[0])
    w = _ws.get(key)
    if w is None or w.numel() < n:
        if w is not None:
            # a captured HIP graph may have this buffer's address baked into its launches (GroupNorm partials): the
            # outgrown buffer stays alive for the life of the process instead of going back to the allocator, where a
            # replay of such a graph would scribble over whatever tensor owned the memory next
            _ws_retired.append(w)
        w = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        _ws[key] = w
    return w


def qp_tensor(entries, device):
    """entries: list of (delta, zp, qmax) python floats / 0-d tensors -> device float[4*n]."""
    rows = []
    for d, z, qmax in entries:
        rows.append(torch.stack([torch.as_tensor(d, dtype=torch.float32, device=device).reshape(()),
                                 torch.as_tensor(z, dtype=torch.float32, device=device).reshape(()),
                                 torch.tensor(float(qmax), device=device), torch.zeros((), device=device)]))
    return torch.stack(rows).reshape(-1).contiguous()


# ------------------------------------------------------------------------------ K1
def fake_quant_fwd(x, delta, zp, qmax, inner=1, u=None, prob=1.0, seed=0, want_codes=False):
    out = torch.empty_like(x)
    codes = torch.empty_like(x) if want_codes else None
    nq = delta.numel()
    lib.call("edadm_fake_quant_fwd", _pf(x), _pf(out), _pf(codes), x.numel(), _pf(delta), _pf(zp), nq, int(inner),
             float(qmax), _pf(u), float(prob), int(seed), _stream())
    return (out, codes) if want_codes else out


def fake_quant_bwd(gy, x, delta, zp, qmax, u=None, prob=1.0, seed=0, need_gx=True):
    gx = torch.empty_like(x) if need_gx else None
    gd = torch.empty(1, dtype=torch.float32, device=x.device)
    lib.call("edadm_fake_quant_bwd", _pf(gy), _pf(x), _pf(gx), _pf(gd), x.numel(), _pf(delta), _pf(zp), float(qmax),
             _pf(u), float(prob), int(seed), _pf(workspace(x.device)), _stream())
    return gx, gd


# ------------------------------------------------------------------------------ K2
def _rows_cols(w):
    return w.shape[0], w[0].numel()


def adaround_init_alpha(w_view, delta):
    """w_view: weight or a dim-1 slice of it (rows keep the parent's leading dimension)."""
    rows, cols = _rows_cols(w_view)
    alpha = torch.empty(w_view.shape, dtype=torch.float32, device=w_view.device)
    lib.call("edadm_adaround_init_alpha", ctypes.c_void_p(w_view.data_ptr()), w_view.stride(0), _pf(alpha), rows, cols,
             _pf(delta.reshape(-1).contiguous()), _stream())
    return alpha


def adaround_fwd(w_view, alpha, out_view, delta, zp, qmax, soft):
    rows, cols = _rows_cols(w_view)
    lib.call("edadm_adaround_fwd", ctypes.c_void_p(w_view.data_ptr()), w_view.stride(0), _pf(alpha),
             ctypes.c_void_p(out_view.data_ptr()), out_view.stride(0), rows, cols, _pf(delta), _pf(zp), float(qmax),
             1 if soft else 0, _stream())
    return out_view


def adaround_bwd(gy_view, w_view, alpha, delta, zp, qmax):
    rows, cols = _rows_cols(w_view)
    ga = torch.empty_like(alpha)
    lib.call("edadm_adaround_bwd", ctypes.c_void_p(gy_view.data_ptr()), gy_view.stride(0),
             ctypes.c_void_p(w_view.data_ptr()), w_view.stride(0), _pf(alpha), _pf(ga), rows, cols, _pf(delta), _pf(zp),
             float(qmax), _stream())
    return ga


# ------------------------------------------------------------------------------ K3
def mse_scores_tensor(x, scale, zp, qmax):
    nc = scale.numel()
    score = torch.empty(nc, dtype=torch.float32, device=x.device)
    lib.call("edadm_mse_scores_tensor", _pf(x), x.numel(), _pf(scale), _pf(zp), nc, float(qmax), _pf(score),
             _pf(workspace(x.device)), _stream())
    return score


def mse_scores_channel(x2d, scale, zp, qmax):
    """x2d [rows][cols]; scale/zp [nc][rows] -> score [nc][rows]."""
    rows, cols = x2d.shape
    nc = scale.shape[0]
    score = torch.empty(nc, rows, dtype=torch.float32, device=x2d.device)
    lib.call("edadm_mse_scores_channel", _pf(x2d), rows, cols, _pf(scale), _pf(zp), nc, float(qmax), _pf(score),
             _stream())
    return score


def mse_candidates(xmin, xmax, mode, one_side, n_bits, num, channel_clamp):
    rows = xmin.numel()
    nc = num if mode == 1 else num * (1 << n_bits)
    scale = torch.empty(nc, rows, dtype=torch.float32, device=xmin.device)
    zp = torch.empty(nc, rows, dtype=torch.float32, device=xmin.device)
    lib.call("edadm_mse_candidates", _pf(xmin), _pf(xmax), rows, int(mode), int(one_side), int(n_bits), int(num),
             1 if channel_clamp else 0, _pf(scale), _pf(zp), _stream())
    return scale, zp


def mse_select(score, xmin, xmax, mode, one_side, n_bits, num, channel_clamp, run_min=None, run_max=None, first=True):
    nc, rows = score.shape
    delta = torch.empty(rows, dtype=torch.float32, device=score.device)
    zp = torch.empty(rows, dtype=torch.float32, device=score.device)
    lib.call("edadm_mse_select", _pf(score), nc, rows, _pf(xmin), _pf(xmax), int(mode), int(one_side), int(n_bits),
             int(num), 1 if channel_clamp else 0, _pf(run_min), _pf(run_max), 1 if first else 0, _pf(delta), _pf(zp),
             _stream())
    return delta, zp


def minmax(x):
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    lib.call("edadm_minmax", _pf(x), x.numel(), _pf(out), _pf(workspace(x.device)), _stream())
    return out


# ------------------------------------------------------------------------------ K7 / K8 / K9 / K10
def lp_loss_fwd(pred, tgt, C=None):
    """C: size of the summed (channel) dimension when it is not dimension 1 (an NHWC memory-order view)"""
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    inv = 1.0 / (pred.numel() / (pred.shape[1] if C is None else C))
    lib.call("edadm_lp_loss_fwd", _pf(pred), _pf(tgt), pred.numel(), inv, _pf(loss), _pf(workspace(pred.device)),
             _stream())
    return loss


def lp_loss_bwd(pred, tgt, gscale, C=None):
    g = torch.empty_like(pred)
    inv = 1.0 / (pred.numel() / (pred.shape[1] if C is None else C))
    lib.call("edadm_lp_loss_bwd", _pf(pred), _pf(tgt), pred.numel(), inv, _pf(gscale), _pf(g), _stream())
    return g


def lp_loss_inject(gout, pred, tgt_rows, idx, row0, nrows, inv_denom, gscale):
    """gout, pred: [rows][...] contiguous in the same memory order; tgt_rows: [*][row elems] contiguous (gathered through idx [nrows],
    int64, or taken row for row when idx is None) -> gout + the loss term's gradient on rows [row0, row0 + nrows)"""
    rows = pred.shape[0]
    re = pred.numel() // rows
    gin = torch.empty_like(gout)
    lib.call("edadm_lp_loss_inject", _pf(gout), _pf(pred), _pf(tgt_rows), None if idx is None else _p(idx, torch.int64), rows, int(row0),
             int(nrows), re, float(inv_denom), _pf(gscale), _pf(gin), _stream())
    return gin


def tdac_pair_scores(feature_map, eps=1e-6):
    """feature_map: list of T device tensors [B, C, ...] -> (mse [T][T], cosine distance sums [T][T]) (edadm_tdac_pair_scores)"""
    f = torch.stack([t.detach().float().contiguous() for t in feature_map]).contiguous()
    T, B, C = f.shape[0], f.shape[1], f.shape[2]
    P = f[0, 0, 0].numel()
    mse = torch.empty(T, T, dtype=torch.float32, device=f.device)
    cd = torch.empty(T, T, dtype=torch.float32, device=f.device)
    lib.call("edadm_tdac_pair_scores", _pf(f), T, B, C, P, float(eps), _pf(mse), _pf(cd), _stream())
    return mse, cd


def adam_step(p, g, m, v, hyper):
    lib.call("edadm_adam_step", _pf(p), _pf(g), _pf(m), _pf(v), p.numel(), _pf(hyper), _stream())


def rng_epoch(value, add=False):
    """Set (add=False) or increment the device-side mask-RNG epoch that edadm_fake_quant_* / edadm_mix_where fold into their
    seed arguments (edadm.h): one launch on the current stream, capturable."""
    lib.call("edadm_rng_epoch", ctypes.c_uint64(int(value)), 1 if add else 0, _stream())


def mix_where(a, b, prob, u=None, seed=0):
    out = torch.empty_like(a)
    lib.call("edadm_mix_where", _pf(a), _pf(b), _pf(out), a.numel(), _pf(u), float(prob), int(seed), _stream())
    return out


def ddim_step(x, e_cond, e_uncond, cfg_scale, coef, noise=None, want_x0=False):
    xp = torch.empty_like(x)
    p0 = torch.empty_like(x) if want_x0 else None
    B = x.shape[0]
    lib.call("edadm_ddim_step", _pf(x), _pf(e_cond), _pf(e_uncond), float(cfg_scale), _pf(coef), _pf(noise), _pf(xp),
             _pf(p0), B, x.numel() // B, _stream())
    return (xp, p0) if want_x0 else xp


def plms_step(x, e_cond, e_uncond, cfg_scale, olds, order, coef, want_x0=False):
    """One PLMS update (K9b).  olds: previous e_t tensors, newest first.  Returns (x_prev, e_t[, pred_x0])."""
    xp, et = torch.empty_like(x), torch.empty_like(x)
    p0 = torch.empty_like(x) if want_x0 else None
    o = list(olds) + [None] * (3 - len(olds))
    B = x.shape[0]
    lib.call("edadm_plms_step", _pf(x), _pf(e_cond), _pf(e_uncond), float(cfg_scale), _pf(o[0]), _pf(o[1]), _pf(o[2]),
             int(order), _pf(coef), _pf(et), _pf(xp), _pf(p0), B, x.numel() // B, _stream())
    return (xp, et, p0) if want_x0 else (xp, et)


# ------------------------------------------------------------------------------ operand producers
class Cat:
    """Channel concatenation [a | b] of two NHWC tensors that is never materialised: GroupNorm and the activation
    quantiser read the two halves in place (edadm_*_cat)."""

    def __init__(self, a, b):
        # b may hold a whole fraction of a's leading dimension: it is then read periodically (image i of the
        # concatenation takes b's image i % len(b)) -- the skip tensors shared by a classifier-free-guidance pair
        assert a.shape[1:-1] == b.shape[1:-1] and a.shape[0] % b.shape[0] == 0
        self.a, self.b = a, b
        self.rep = a.shape[0] // b.shape[0]
        self.shape = tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],)
        self.device = a.device

    def rows2d(self):
        return Cat(self.a.reshape(-1, self.a.shape[-1]), self.b.reshape(-1, self.b.shape[-1]))

    def full_b(self):
        """b at a's batch (materialised only by the rare consumers that cannot read it periodically)"""
        return self.b if self.rep == 1 else torch.cat([self.b] * self.rep)


def quant_i8(x2d, qp, split=0, out=None):
    if isinstance(x2d, Cat):
        a, b = x2d.a.reshape(-1, x2d.a.shape[-1]), x2d.b.reshape(-1, x2d.b.shape[-1])
        rows, C = a.shape[0], a.shape[1] + b.shape[1]
        if out is None:
            out = torch.empty(rows, C, dtype=torch.int8, device=a.device)
        lib.call("edadm_quant_i8_cat_rep", _pf(a), a.shape[1], _pf(b), b.shape[1], _p(out, torch.int8), rows, _pf(qp),
                 int(split), b.shape[0] if b.shape[0] != rows else 0, _stream())
        return out
    rows, C = x2d.shape
    if out is None:
        out = torch.empty(rows, C, dtype=torch.int8, device=x2d.device)
    lib.call("edadm_quant_i8", _pf(x2d), _p(out, torch.int8), rows, C, _pf(qp), int(split), _stream())
    return out


def quant_f16(x2d, qp, premul=1.0, out=None):
    rows, C = x2d.shape
    if out is None:
        out = torch.empty(rows, C, dtype=torch.float16, device=x2d.device)
    lib.call("edadm_quant_f16", ctypes.c_void_p(x2d.data_ptr()), x2d.stride(0), ctypes.c_void_p(out.data_ptr()),
             out.stride(0), rows, C, _pf(qp), float(premul), _stream())
    return out


def quant_f16_qkv(x2d, d, qp3, premuls):
    """legacy qkv tensor [rows][heads x (q|k|v) x d] -> f16 codes, same layout, one launch (edadm_quant_f16_qkv)"""
    rows, C = x2d.shape
    out = torch.empty(rows, C, dtype=torch.float16, device=x2d.device)
    lib.call("edadm_quant_f16_qkv", _pf(x2d), x2d.stride(0), ctypes.c_void_p(out.data_ptr()), C, rows, C, int(d), _pf(qp3),
             float(premuls[0]), float(premuls[1]), float(premuls[2]), _stream())
    return out


def attention_fused_ok(heads, d, Nq, Nk):
    return bool(lib.load().edadm_attention_fused_ok(int(heads), int(d), int(Nq), int(Nk)))


def attention_fused(q, k, v, B, heads, Nq, Nk, d, alpha_qk, pqp, alpha_pv, q_off=0, k_off=0, v_off=0, head_stride=None,
                    out_qp=None):
    """K6f: q [B*Nq][ldq], k / v [B*Nk][ld] f16 codes (head h at column *_off + h * head_stride) -> [B*Nq][heads*d] fp32, or the
    consumer's int8 operand with out_qp.  No score matrix in memory."""
    hs = int(head_stride or d)
    hd = heads * d
    out = torch.empty(B * Nq, hd, dtype=torch.int8 if out_qp is not None else torch.float32, device=q.device)
    esz = 2
    lib.call("edadm_attention_fused_f16", ctypes.c_void_p(q.data_ptr() + q_off * esz), q.stride(0), Nq * q.stride(0), hs,
             ctypes.c_void_p(k.data_ptr() + k_off * esz), k.stride(0), Nk * k.stride(0), hs,
             ctypes.c_void_p(v.data_ptr() + v_off * esz), v.stride(0), Nk * v.stride(0), hs,
             ctypes.c_void_p(out.data_ptr()), hd, Nq * hd, int(B), int(heads), int(Nq), int(Nk), int(d), float(alpha_qk), _pf(pqp),
             float(alpha_pv), 2 if out_qp is not None else 0, _pf(out_qp), _stream())
    return out


def attention_i8qk_ok(heads, d, Nq, Nk):
    return bool(lib.load().edadm_attention_fused_i8qk_ok(int(heads), int(d), int(Nq), int(Nk)))


def attention_fused_i8qk(q8, k8, v, B, heads, Nq, Nk, d, alpha_qk, zq, pqp, alpha_pv, out_qp=None):
    """K6w on the int8 MFMA for the scores: q8 / k8 int8 operands (code - 128) [B*N][heads*d], v f16 codes; zq: zero point of the q
    quantiser.  -> [B*Nq][heads*d] fp32, or the consumer's int8 operand with out_qp."""
    hd = heads * d
    out = torch.empty(B * Nq, hd, dtype=torch.int8 if out_qp is not None else torch.float32, device=v.device)
    lib.call("edadm_attention_fused_i8qk", ctypes.c_void_p(q8.data_ptr()), q8.stride(0), Nq * q8.stride(0), d,
             ctypes.c_void_p(k8.data_ptr()), k8.stride(0), Nk * k8.stride(0), d, ctypes.c_void_p(v.data_ptr()), v.stride(0),
             Nk * v.stride(0), d, ctypes.c_void_p(out.data_ptr()), hd, Nq * hd, int(B), int(heads), int(Nq), int(Nk), int(d), float(alpha_qk),
             float(zq), _pf(pqp), float(alpha_pv), 2 if out_qp is not None else 0, _pf(out_qp), _stream())
    return out


def nchw_to_nhwc(x):
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    out = torch.empty((B,) + tuple(x.shape[2:]) + (C,), dtype=torch.float32, device=x.device)
    lib.call("edadm_nchw_to_nhwc", _pf(x), _pf(out), B, C, HW, _stream())
    return out


def nhwc_to_nchw(x):
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    out = torch.empty((B, C) + tuple(x.shape[1:-1]), dtype=torch.float32, device=x.device)
    lib.call("edadm_nhwc_to_nchw", _pf(x), _pf(out), B, C, HW, _stream())
    return out


def im2col_quant_i8(x_nhwc, Kpad, qp):
    B, H, W, C = x_nhwc.shape
    out = torch.empty(B * H * W, Kpad, dtype=torch.int8, device=x_nhwc.device)
    lib.call("edadm_im2col_quant_i8", _pf(x_nhwc), _p(out, torch.int8), B, H, W, C, Kpad, _pf(qp), _stream())
    return out


def groupnorm_stats(x_nhwc, G, eps):
    if isinstance(x_nhwc, Cat):
        a, b = x_nhwc.a, x_nhwc.b
        B, C = a.shape[0], x_nhwc.shape[-1]
        HW = a.numel() // (B * a.shape[-1])
        stats = torch.empty(B, G, 2, dtype=torch.float32, device=a.device)
        ws = workspace(a.device, lib.load().edadm_gn_ws_floats(B, HW, C))
        lib.call("edadm_groupnorm_stats_cat_rep", _pf(a), a.shape[-1], _pf(b), b.shape[-1], _pf(stats), _pf(ws), B, HW, G,
                 float(eps), b.shape[0] if x_nhwc.rep > 1 else 0, _stream())
        return stats
    B, C = x_nhwc.shape[0], x_nhwc.shape[-1]
    HW = x_nhwc.numel() // (B * C)
    stats = torch.empty(B, G, 2, dtype=torch.float32, device=x_nhwc.device)
    ws = workspace(x_nhwc.device, lib.load().edadm_gn_ws_floats(B, HW, C))
    lib.call("edadm_groupnorm_stats", _pf(x_nhwc), _pf(stats), _pf(ws), B, HW, C, G, float(eps), _stream())
    return stats


def groupnorm_apply(x_nhwc, stats, gamma, beta, G, silu, qp=None, nq=0, want_f32=False, scale_shift=None, raw_qp=None,
                    raw_split=0):
    """-> (fp32 or None, [int8 operands of the normalised value]) -- and, with raw_qp, a third element: the int8 operand of
    the un-normalised input under raw_qp (split quantisers at column raw_split), for the tensor's second consumer."""
    cat = isinstance(x_nhwc, Cat)
    a, b = (x_nhwc.a, x_nhwc.b) if cat else (x_nhwc, None)
    B, C = a.shape[0], x_nhwc.shape[-1]
    HW = a.numel() // (B * a.shape[-1])
    dev = a.device
    out = torch.empty(tuple(x_nhwc.shape), dtype=torch.float32, device=dev) if want_f32 else None
    qs = [torch.empty(tuple(x_nhwc.shape), dtype=torch.int8, device=dev) for _ in range(nq)]
    qq = qs + [None] * (3 - nq)
    if raw_qp is not None or (cat and x_nhwc.rep > 1):
        qraw = torch.empty(tuple(x_nhwc.shape), dtype=torch.int8, device=dev) if raw_qp is not None else None
        lib.call("edadm_groupnorm_apply_cat_raw", _pf(a), a.shape[-1], _pf(b), b.shape[-1] if cat else 0, _pf(stats),
                 _pf(gamma), _pf(beta), _pf(scale_shift), B, HW, G, 1 if silu else 0, _pf(out), _p(qq[0]), _p(qq[1]),
                 _p(qq[2]), _pf(qp), nq, _p(qraw, torch.int8) if qraw is not None else None, _pf(raw_qp), int(raw_split),
                 b.shape[0] if (cat and x_nhwc.rep > 1) else 0, _stream())
        return (out, qs, qraw) if raw_qp is not None else (out, qs)
    if cat:
        lib.call("edadm_groupnorm_apply_cat", _pf(a), a.shape[-1], _pf(b), b.shape[-1], _pf(stats), _pf(gamma), _pf(beta),
                 _pf(scale_shift), B, HW, G, 1 if silu else 0, _pf(out), _p(qq[0]), _p(qq[1]), _p(qq[2]), _pf(qp), nq,
                 _stream())
        return out, qs
    lib.call("edadm_groupnorm_apply", _pf(x_nhwc), _pf(stats), _pf(gamma), _pf(beta), _pf(scale_shift), B, HW, C, G,
             1 if silu else 0, _pf(out), _p(qq[0]), _p(qq[1]), _p(qq[2]), _pf(qp), nq, _stream())
    return out, qs


def layernorm_quant_radd(x2d, radd, rows_per_batch, gamma, beta, eps, qp, nq, rows=None):
    """LayerNorm(x2d[m % len(x2d)] + radd[m // rows_per_batch]) -> (the sum [rows][C], [int8 operands])."""
    xr, C = x2d.shape
    rows = xr if rows is None else int(rows)
    summ = torch.empty(rows, C, dtype=torch.float32, device=x2d.device)
    qs = [torch.empty(rows, C, dtype=torch.int8, device=x2d.device) for _ in range(nq)]
    qq = qs + [None] * (3 - nq)
    lib.call("edadm_layernorm_quant_radd", _pf(x2d), xr, _pf(radd), int(rows_per_batch), _pf(summ), _pf(gamma), _pf(beta),
             rows, C, float(eps), _p(qq[0]), _p(qq[1]), _p(qq[2]), _pf(qp), nq, _stream())
    return summ, qs


def layernorm_quant(x2d, gamma, beta, eps, qp=None, nq=0, want_f32=False):
    rows, C = x2d.shape
    out = torch.empty_like(x2d) if want_f32 else None
    qs = [torch.empty(rows, C, dtype=torch.int8, device=x2d.device) for _ in range(nq)]
    qq = qs + [None] * (3 - nq)
    lib.call("edadm_layernorm_quant", _pf(x2d), _pf(gamma), _pf(beta), rows, C, float(eps), _pf(out), _p(qq[0]),
             _p(qq[1]), _p(qq[2]), _pf(qp), nq, _stream())
    return out, qs


def silu_quant_i8(x, qp):
    out = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    lib.call("edadm_silu_quant_i8", _pf(x), _p(out, torch.int8), x.numel(), _pf(qp), _stream())
    return out


def geglu_quant_i8(x2d, qp):
    rows, two = x2d.shape
    out = torch.empty(rows, two // 2, dtype=torch.int8, device=x2d.device)
    lib.call("edadm_geglu_quant_i8", _pf(x2d), _p(out, torch.int8), rows, two // 2, _pf(qp), _stream())
    return out


def silu(x):
    out = torch.empty_like(x)
    lib.call("edadm_silu", _pf(x), _pf(out), x.numel(), _stream())
    return out


def add(a, b):
    out = torch.empty_like(a)
    lib.call("edadm_add", _pf(a), _pf(b), _pf(out), a.numel(), _stream())
    return out


def add_rowbcast(x2d, r, rows_per_batch, out=None, rows=None):
    """x2d[m] + r[m // rows_per_batch] (a per-image vector added to every row of the image).  rows > len(x2d): x2d is
    read periodically and the result has `rows` rows (the shared half of a guidance pair fanning out)."""
    xr, C = x2d.shape
    rows = xr if rows is None else int(rows)
    out = torch.empty(rows, C, dtype=torch.float32, device=x2d.device) if out is None else out
    lib.call("edadm_add_rowbcast_rep", _pf(x2d), _pf(r), _pf(out), rows, C, int(rows_per_batch), xr if rows != xr else 0,
             _stream())
    return out


def concat_c(a, b):
    Ca, Cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // Ca
    out = torch.empty(tuple(a.shape[:-1]) + (Ca + Cb,), dtype=torch.float32, device=a.device)
    lib.call("edadm_concat_c", _pf(a), Ca, _pf(b), Cb, _pf(out), rows, _stream())
    return out


def avgpool2_nhwc(x):
    B, H, W, C = x.shape
    out = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    lib.call("edadm_avgpool2_nhwc", _pf(x), _pf(out), B, H, W, C, _stream())
    return out


def upsample2_nhwc(x):
    B, H, W, C = x.shape
    out = torch.empty(B, H * 2, W * 2, C, dtype=torch.float32, device=x.device)
    lib.call("edadm_upsample2_nhwc", _pf(x), _pf(out), B, H, W, C, _stream())
    return out


# ------------------------------------------------------------------------------ K4 / K6
def make_geom(B, H, W, Cin, Ho, Wo, KH, KW, stride, pad0, ups, padval):
    return (ctypes.c_int32 * 16)(1, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad0, 1 if ups else 0, int(padval), 0, 0, 0)


def groupnorm_final(ws1, C1, ws2, C2, B, HW, G, eps, B2=0, rows1=64, rows2=64):
    """stats[B][G][2] from producer-written partials [B][HW/rows][C][2] (ws2: second half of a concatenation, B2 > 0: of
    B2 images read periodically; rows1 / rows2: rows per slab of the two producers)."""
    stats = torch.empty(B, G, 2, dtype=torch.float32, device=ws1.device)
    lib.call("edadm_groupnorm_final_cat_rep2", _pf(ws1), C1, _pf(ws2), C2 if ws2 is not None else 0, _pf(stats), B, HW, G,
             HW // int(rows1), HW // int(rows2) if ws2 is not None else 0, float(eps), int(B2), _stream())
    return stats


def device_status(clear=True):
    """Synchronise the current stream and raise if a kernel recorded a deferred error (edadm_device_status)."""
    rc = lib.load().edadm_device_status(1 if clear else 0, _stream())
    if rc != 0:
        raise RuntimeError("libedadm: deferred device error %d (a persistent-GEMM hand-off timed out: outputs of that launch "
                           "are invalid)" % rc)


def qgemm_i8_gn_ok(M, N, hw):
    return bool(lib.load().edadm_qgemm_i8_gn_ok(int(M), int(N), int(hw)))


def qgemm_i8(A, Wt, M, N, K, scale, bias, out, geom=None, lda=None, ldw=None, rowadd=None, rows_per_batch=1,
             residual=None, gn_ws=None, gn_hw=0):
    """out[M][N] (fp32, contiguous rows of length N) = scale[n] * (A . Wt^T) + bias[n] [+rowadd] [+residual]; gn_ws [M / 64][N][2]:
    the per-channel (sum, sum of squares) of every 64-row slab of out, from the epilogue's registers (edadm_qgemm_i8_gn)."""
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    gptr = ctypes.cast(geom, ctypes.c_void_p) if geom is not None else None
    if gn_ws is not None:
        lib.call("edadm_qgemm_i8_gn", ctypes.c_void_p(A.data_ptr()), int(lda), ctypes.c_void_p(Wt.data_ptr()), int(ldw),
                 int(M), int(N), int(K), gptr, _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual),
                 int(N), _pf(out), int(N), _pf(gn_ws), int(gn_hw), _stream())
        return out
    lib.call("edadm_qgemm_i8", ctypes.c_void_p(A.data_ptr()), int(lda), ctypes.c_void_p(Wt.data_ptr()), int(ldw),
             int(M), int(N), int(K), gptr, _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual),
             int(N), _pf(out), int(N), _stream())
    return out


def qgemm_i8_split2_ok(M, N, K1, K2):
    return bool(lib.load().edadm_qgemm_i8_split2_ok(int(M), int(N), int(K1), int(K2)))


def qgemm_i8_split2(A, W1, W2, M, N, K1, K2, scale1, scale2, bias, out):
    """edadm_qgemm_i8_split2: a dense layer with split quantisers in one launch; A [M][K1 + K2] int8 (the two ranges side by side)"""
    lib.call("edadm_qgemm_i8_split2", ctypes.c_void_p(A.data_ptr()), int(A.stride(0)), int(K1), ctypes.c_void_p(W1.data_ptr()), int(K1),
             ctypes.c_void_p(W2.data_ptr()), int(K2), int(M), int(N), int(K1), int(K2), _pf(scale1), _pf(scale2), _pf(bias), _pf(out), int(N),
             _stream())
    return out


def qgemm_w4(A, W4, zp4, M, N, K, scale, bias, out, lda=None, rowadd=None, rows_per_batch=1, residual=None):
    """edadm_qgemm_w4: int8 activations x packed 4-bit weights (nibbles expanded in registers), fp32 out [M][N]"""
    lib.call("edadm_qgemm_w4", ctypes.c_void_p(A.data_ptr()), int(K if lda is None else lda), ctypes.c_void_p(W4.data_ptr()), _pf(zp4),
             int(M), int(N), int(K), _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual), int(N), _pf(out), int(N),
             _stream())
    return out


def conv3_direct_ok(B, H, W, Cin, N):
    return bool(lib.load().edadm_conv3_direct_ok(int(B), int(H), int(W), int(Cin), int(N)))


def conv3_direct_tile(B, H, W, Cin, N):
    """output pixels per workgroup tile of edadm_qconv3_i8_direct for this shape (256 / 128; 0: not taken); its GroupNorm
    partials come in slabs of tile / 4 rows"""
    return int(lib.load().edadm_conv3_direct_tile(int(B), int(H), int(W), int(Cin), int(N)))


def conv3_pack_w(w_i8, N, Cin):
    """[N][3][3][Cin] int8 filter -> the layout of edadm_qconv3_i8_direct ([N/192][Cin/64][3][3][192][64], chunks swizzled)."""
    rows = int(lib.load().edadm_conv3_packed_rows(int(N)))
    out = torch.empty(rows * 9 * Cin, dtype=torch.int8, device=w_i8.device)
    lib.call("edadm_conv3_pack_w", _p(w_i8, torch.int8), _p(out, torch.int8), int(N), int(Cin), _stream())
    return out


def qconv3_i8_direct(a_nhwc, wdc, B, H, W, Cin, N, padval, scale, bias, out, rowadd=None, rows_per_batch=1, residual=None,
                     ups=False, gn_ws=None):
    """3x3 / stride 1 / pad 1 convolution of the int8 NHWC operand with the input patch resident in LDS (edadm.h)."""
    lib.call("edadm_qconv3_i8_direct", _p(a_nhwc, torch.int8), _p(wdc, torch.int8), int(B), int(H), int(W), int(Cin), int(N),
             int(padval), 1 if ups else 0, _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual), int(N), _pf(out),
             int(N), _pf(gn_ws), _stream())
    return out


def gemm_f16_nt(A, lda, strideA, Bm, ldb, strideB, batch, M, N, K, alpha, out=None, inner=1, strideA_i=0,
                strideB_i=0, ldc=None, strideC=None, strideC_i=0):
    """C[z] = alpha * A[z] . B[z]^T, z = outer*inner + head (two-level strides, in elements)."""
    ldc = N if ldc is None else ldc
    strideC = M * N * inner if strideC is None else strideC
    if strideC_i == 0 and inner > 1:
        strideC_i = M * N
    if out is None:
        out = torch.empty(batch * inner, M, N, dtype=torch.float32, device=A.device)
    lib.call("edadm_gemm_f16_nt", ctypes.c_void_p(A.data_ptr()), int(lda), int(strideA), int(strideA_i),
             ctypes.c_void_p(Bm.data_ptr()), int(ldb), int(strideB), int(strideB_i), _pf(out), int(ldc), int(strideC),
             int(strideC_i), int(batch), int(inner), int(M), int(N), int(K), float(alpha), _stream())
    return out


def qgemm_f16(A, Wt, M, N, K, scale, bias, out, geom=None, lda=None, ldw=None, rowadd=None, rows_per_batch=1,
              residual=None):
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    gptr = ctypes.cast(geom, ctypes.c_void_p) if geom is not None else None
    lib.call("edadm_qgemm_f16", ctypes.c_void_p(A.data_ptr()), int(lda), ctypes.c_void_p(Wt.data_ptr()), int(ldw),
             int(M), int(N), int(K), gptr, _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual),
             int(N), _pf(out), int(N), _stream())
    return out


_OUT_DT = {1: torch.float16, 2: torch.int8, 3: torch.int8, 4: torch.float16}


def vt_mode_ok(M, N, rows_per_batch):
    """can the v projection write its f16 operand transposed per image (edadm_qgemm_i8_q out_mode 4)?"""
    return rows_per_batch % 32 == 0 and M % rows_per_batch == 0 and M % 128 == 0 and N % (192 if N % 192 == 0 else 128) == 0


def qgemm_i8_q(A, Wt, M, N, K, scale, bias, out_mode, oqp, geom=None, lda=None, ldw=None, rowadd=None,
               rows_per_batch=1, residual=None):
    """Quantised-output form of qgemm_i8: returns the consumer's MFMA operand directly
    (1: f16 code-zp [M][N], 2: int8 code-128 [M][N], 3: GEGLU + int8 [M][N/2])."""
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    ncol = N // 2 if out_mode == 3 else N
    if out_mode == 4:                                      # [B][N][rows_per_batch]
        out = torch.empty(M // rows_per_batch, N, rows_per_batch, dtype=torch.float16, device=A.device)
        ncol = rows_per_batch
    else:
        out = torch.empty(M, ncol, dtype=_OUT_DT[out_mode], device=A.device)
    gptr = ctypes.cast(geom, ctypes.c_void_p) if geom is not None else None
    lib.call("edadm_qgemm_i8_q", ctypes.c_void_p(A.data_ptr()), int(lda), ctypes.c_void_p(Wt.data_ptr()), int(ldw),
             int(M), int(N), int(K), gptr, _pf(scale), _pf(bias), _pf(rowadd), int(rows_per_batch), _pf(residual),
             int(N), ctypes.c_void_p(out.data_ptr()), int(ncol), int(out_mode), _pf(oqp), _stream())
    return out


class _GemmProblem(ctypes.Structure):
    """struct edadm_gemm_problem (include/edadm.h)"""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
                ("scale", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64),
                ("oqp", ctypes.c_void_p), ("N", ctypes.c_int64), ("rows_per_batch", ctypes.c_int64), ("out_mode", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


def qgemm_i8_grouped_q_ok(M, N, K):
    """N: the output columns of all problems of the launch together"""
    return bool(lib.load().edadm_qgemm_i8_grouped_q_ok(int(M), int(N), int(K)))


def qgemm_i8_grouped_q(problems, M, K):
    """edadm_qgemm_i8_grouped_q: up to four quantised-output dense layers of the same M and K in ONE launch (the q / k / v projections
    of a self-attention, or one GEGLU projection).  problems: dicts with A (int8 [M][lda]), W (int8 [N][K]), N, scale, bias, out_mode,
    oqp and rows_per_batch (mode 4).  Returns the outputs in order -- the bits of one qgemm_i8_q call per problem."""
    arr = (_GemmProblem * len(problems))()
    outs = []
    for q, pr in zip(arr, problems):
        A, N, mode = pr["A"], int(pr["N"]), int(pr["out_mode"])
        rpb = int(pr.get("rows_per_batch") or 0)
        if mode == 4:
            out = torch.empty(M // rpb, N, rpb, dtype=torch.float16, device=A.device)
            ldo = rpb
        else:
            ldo = N // 2 if mode == 3 else N
            out = torch.empty(M, ldo, dtype=_OUT_DT[mode], device=A.device)
        outs.append(out)
        q.A, q.lda, q.W, q.ldw = A.data_ptr(), int(pr.get("lda") or K), pr["W"].data_ptr(), int(pr.get("ldw") or K)
        q.scale, q.bias = pr["scale"].data_ptr(), (pr["bias"].data_ptr() if pr.get("bias") is not None else None)
        q.out, q.ldo, q.oqp, q.N, q.rows_per_batch, q.out_mode, q.reserved = out.data_ptr(), ldo, pr["oqp"].data_ptr(), N, rpb, mode, 0
        assert A.dtype == torch.int8 and pr["W"].dtype == torch.int8 and pr["scale"].dtype == torch.float32 and pr["oqp"].dtype == torch.float32
    lib.call("edadm_qgemm_i8_grouped_q", ctypes.cast(arr, ctypes.c_void_p), len(problems), int(M), int(K), _stream())
    return outs


def gemm_f16_nt_q(A, lda, strideA, Bm, ldb, strideB, batch, M, N, K, alpha, out, out_mode, oqp, inner=1,
                  strideA_i=0, strideB_i=0, ldc=None, strideC=None, strideC_i=0):
    ldc = N if ldc is None else ldc
    strideC = M * N * inner if strideC is None else strideC
    lib.call("edadm_gemm_f16_nt_q", ctypes.c_void_p(A.data_ptr()), int(lda), int(strideA), int(strideA_i),
             ctypes.c_void_p(Bm.data_ptr()), int(ldb), int(strideB), int(strideB_i), ctypes.c_void_p(out.data_ptr()),
             int(ldc), int(strideC), int(strideC_i), int(batch), int(inner), int(M), int(N), int(K), float(alpha),
             int(out_mode), _pf(oqp), _stream())
    return out


def vq_nearest(z2d, codebook):
    """rows of z2d [R][D] -> their nearest codebook rows (edadm_vq_nearest; D <= 8)"""
    out = torch.empty_like(z2d)
    lib.call("edadm_vq_nearest", _pf(z2d), _pf(codebook), _pf(out), None, z2d.shape[0], z2d.shape[1], codebook.shape[0], _stream())
    return out


def conv3x3_f32_smalln(x_nhwc, w, bias):
    B, H, W, C = x_nhwc.shape
    N = w.shape[0]
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=x_nhwc.device)
    lib.call("edadm_conv3x3_f32_smalln", _pf(x_nhwc), _pf(w), _pf(bias), _pf(out), B, H, W, C, N, _stream())
    return out


def softmax_quant_f16(s2d, qp, ldo=None):
    rows, cols = s2d.shape
    ldo = cols if ldo is None else ldo
    out = torch.empty(rows, ldo, dtype=torch.float16, device=s2d.device)
    lib.call("edadm_softmax_quant_f16", _pf(s2d), ctypes.c_void_p(out.data_ptr()), rows, cols, ldo, _pf(qp), _stream())
    return out


def transpose_f16(x, ldx, strideX, batch, n, d, ldo, out=None, strideO=None):
    """[b][n][d] (row stride ldx, batch stride strideX) -> [b][d][ldo] with zero padding past n."""
    if out is None:
        out = torch.empty(batch, d, ldo, dtype=torch.float16, device=x.device)
    strideO = d * ldo if strideO is None else strideO
    lib.call("edadm_transpose_f16", ctypes.c_void_p(x.data_ptr()), int(ldx), int(strideX),
             ctypes.c_void_p(out.data_ptr()), int(ldo), int(strideO), int(batch), int(n), int(d), _stream())
    return out


def unpack_w4(packed, zp, rows, cols):
    out = torch.empty(rows, cols, dtype=torch.int8, device=packed.device)
    lib.call("edadm_unpack_w4", _p(packed, torch.uint8), _pf(zp), _p(out, torch.int8), rows, cols, _stream())
    return out


def pack_w4(w_i8, zp):
    """int8 operand rows (code - zp[row], codes in [0, 15]) -> packed nibbles [rows*cols/2] (uint8)."""
    rows, cols = w_i8.shape
    out = torch.empty(rows * cols // 2, dtype=torch.uint8, device=w_i8.device)
    lib.call("edadm_pack_w4", _p(w_i8, torch.int8), _pf(zp), _p(out, torch.uint8), rows, cols, _stream())
    return out


# ------------------------------------------------------------------------------ fp32 contraction (H1)
def gemm_f32_nt(A, Bm, M, N, K, alpha=1.0, bias=None, residual=None, out=None, lda=None, ldb=None, batch=1,
                strideA=0, strideB=0, strideC=None):
    """C[z][M][N] = alpha * A[z] . B[z]^T (+bias[n]) (+residual) on the fp32 MFMA."""
    lda = K if lda is None else lda
    ldb = K if ldb is None else ldb
    strideC = M * N if strideC is None else strideC
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=torch.float32, device=A.device)
    lib.call("edadm_gemm_f32_nt", ctypes.c_void_p(A.data_ptr()), int(lda), int(strideA), ctypes.c_void_p(Bm.data_ptr()),
             int(ldb), int(strideB), _pf(out), int(N), int(strideC), int(batch), int(M), int(N), int(K), float(alpha),
             _pf(bias), _pf(residual), int(N), _stream())
    return out


def split_f16(x, R, T, C, order, per_row, other=None, N=0, amax=None):
    """fp32 [R][T][C] -> f16 [R][T][3][C] two-term expansion (edadm_split_f16).  Returns (planes, inv, comb)."""
    out = torch.empty(R, T * (2 if order == 2 else 3) * C, dtype=torch.float16, device=x.device)
    inv = torch.empty(R if per_row else 1, dtype=torch.float32, device=x.device)
    comb = torch.empty(N, dtype=torch.float32, device=x.device) if other is not None else None
    lib.call("edadm_split_f16", _pf(x), int(R), int(T), int(C), int(order), 1 if per_row else 0, _pf(amax),
             ctypes.c_void_p(out.data_ptr()), _pf(inv), _pf(other), int(other.numel()) if other is not None else 0,
             _pf(comb), int(N), _pf(workspace(x.device)), _stream())
    return out, inv, comb


def absmax_parts(x):
    """1024 partial maxima of |x| (edadm_absmax_parts): the scan an operand needs before its f16 expansion, kept so that
    a tensor expanded twice (forward, weight gradient) is scanned once."""
    parts = torch.empty(1024, dtype=torch.float32, device=x.device)
    lib.call("edadm_absmax_parts", _pf(x), x.numel(), _pf(parts), _stream())
    return parts


def transpose_split_f16(x2d, L, order, amax=None, conv=None):
    """[R][C] fp32 -> ([C][R/L][...] f16 with row stride out.shape[1], inv) (edadm_transpose_split_f16).  conv = (KH, KW, stride, pad, Ho, Wo): x2d is
    the NHWC activation [B][H][W][C] and the matrix its im2col, gathered on the fly."""
    if conv is not None:
        B, H, W, Cc = x2d.shape
        KH, KW, stride, pad, Ho, Wo = conv
        R, C = B * Ho * Wo, KH * KW * Cc
        geom = (ctypes.c_int32 * 10)(B, H, W, Cc, Ho, Wo, KH, KW, stride, pad)
        gptr = ctypes.cast(geom, ctypes.c_void_p)
    else:
        (R, C), gptr = x2d.shape, None
    ld = (2 if order == 2 else 3) * R + 128          # rows a power of two apart would all sit on the same HBM channels
    out = torch.empty(C, ld, dtype=torch.float16, device=x2d.device)
    inv = torch.empty(1, dtype=torch.float32, device=x2d.device)
    lib.call("edadm_transpose_split_f16", _pf(x2d), int(R), int(C), int(L), int(order), gptr, _pf(amax),
             ctypes.c_void_p(out.data_ptr()), int(ld), _pf(inv), _pf(workspace(x2d.device)), _stream())
    return out, inv


def f16x3_conv_ok(x, w, ups=False):
    """shapes the three-product f16 path takes: 16-byte f16 channel groups and 32-bit gather offsets"""
    B, H, W, C = x.shape
    return C % 16 == 0 and B * H * W * C * 4 < (1 << 31) and w.shape[0] >= 16


F16X3_DIRECT = True      # 3x3 / stride 1 / pad 1 three-product convolutions with the input patch resident in LDS (edadm_qconv3_f16x3_direct)


def conv3_f16x3_direct_ok(B, H, W, C, N):
    """shapes edadm_qconv3_f16x3_direct takes (H, W: the dimensions the convolution runs over)"""
    return F16X3_DIRECT and bool(lib.load().edadm_conv3_f16x3_direct_ok(int(B), int(H), int(W), int(C), int(N)))


def conv3_f16x3_pack_w(wb, N, C):
    """order-2 expansion of a 3x3 filter [N][9 * 2 C] f16 -> the direct kernel's layout (edadm_conv3_pack_w on 4 C bytes per tap)"""
    rows = int(lib.load().edadm_conv3_packed_rows(int(N)))
    out = torch.empty(rows * 9 * 4 * C, dtype=torch.int8, device=wb.device)
    lib.call("edadm_conv3_pack_w", ctypes.c_void_p(wb.data_ptr()), _p(out, torch.int8), int(N), int(4 * C), _stream())
    return out


def conv3_f16x3_direct(xa, B, H, W, C, wdc, N, comb, bias=None, residual=None, ups=False):
    """xa: order-2 expansion of the NHWC activation (stored at [H/2][W/2] with ups); -> fp32 [B][H][W][N]"""
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=xa.device)
    lib.call("edadm_qconv3_f16x3_direct", ctypes.c_void_p(xa.data_ptr()), _p(wdc, torch.int8), int(B), int(H), int(W), int(C), int(N),
             1 if ups else 0, _pf(comb), _pf(bias), _pf(residual), int(N), _pf(out), int(N), _stream())
    return out


def conv2d_f16x3_nhwc(x, w, bias=None, residual=None, stride=1, pad=1, ups=False, presplit=None, amax=None):
    """conv2d_f32_nhwc's contract on the f16 MFMA: both operands as two-term f16 expansions ([hi x16 | lo x16] per 16
    channels), three products per K-slice in one implicit GEMM, fp32 accumulation (fp32-grade result, see csrc/elem.hip
    and operand type 3 of csrc/gemm.hip).  C % 16 == 0."""
    B, H, W, C = x.shape
    N, KH, KW, _ = w.shape
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hl + 2 * pad - KH) // stride + 1, (Wl + 2 * pad - KW) // stride + 1
    wb, inv_b = (presplit[0], presplit[1]) if presplit is not None else split_f16(w, N, KH * KW, C, 2, True)[:2]   # static weights: split once
    xa, _, comb = split_f16(x, B * H * W, 1, C, 2, False, other=inv_b, N=N, amax=amax)
    if KH == 3 and KW == 3 and stride == 1 and pad == 1 and conv3_f16x3_direct_ok(B, Hl, Wl, C, N):
        wdc = presplit[2] if presplit is not None and len(presplit) > 2 else conv3_f16x3_pack_w(wb, N, C)
        return conv3_f16x3_direct(xa, B, Hl, Wl, C, wdc, N, comb, bias, residual, ups)
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32, device=x.device)
    geom = (ctypes.c_int32 * 12)(1, B, H, W, 2 * C, Ho, Wo, KH, KW, stride, pad, 1 if ups else 0)
    K2 = KH * KW * 2 * C
    lib.call("edadm_qgemm_f16x3", ctypes.c_void_p(xa.data_ptr()), int(K2), ctypes.c_void_p(wb.data_ptr()), int(K2),
             int(B * Ho * Wo), int(N), int(K2), ctypes.cast(geom, ctypes.c_void_p), _pf(comb), _pf(bias), _pf(residual),
             int(N), _pf(out), int(N), _stream())
    return out


def gn_split_f16(x, gamma, beta, G, eps, silu, other, N):
    """GroupNorm (+ swish) of x [B][H][W][C] written straight as the order-2 expansion a three-product convolution reads
    (edadm_gn_split_f16): (planes [B H W][2 C] f16, comb [N]) with comb = this operand's inverse scale x other (the filter's)."""
    B, H, W, C = x.shape
    xa = torch.empty(B * H * W, 2 * C, dtype=torch.float16, device=x.device)
    inv = torch.empty(1, dtype=torch.float32, device=x.device)
    comb = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = workspace(x.device, lib.load().edadm_gn_split_ws_floats(B, H * W, C, int(G)))
    lib.call("edadm_gn_split_f16", _pf(x), B, H * W, C, int(G), float(eps), _pf(gamma), _pf(beta), 1 if silu else 0,
             ctypes.c_void_p(xa.data_ptr()), _pf(inv), _pf(other), int(other.numel()), _pf(comb), int(N), _pf(ws), _stream())
    return xa, comb


def gn_split_ok(x, G):
    B, H, W, C = x.shape
    return C % 16 == 0 and C <= 1024 and C % G == 0 and B * G <= 1024 and B <= 65535


def conv2d_f16x3_pre(xa, comb, shape, w_shape, wb, bias=None, residual=None, stride=1, pad=1, ups=False, wdc=None):
    """conv2d_f16x3_nhwc on an activation that is already expanded (gn_split_f16): shape = (B, H, W, C) of the fp32 tensor.
    wdc: the filter packed for the direct kernel (conv3_f16x3_pack_w), used when the shape qualifies."""
    B, H, W, C = shape
    N, KH, KW, _ = w_shape
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hl + 2 * pad - KH) // stride + 1, (Wl + 2 * pad - KW) // stride + 1
    if wdc is not None and KH == 3 and KW == 3 and stride == 1 and pad == 1 and conv3_f16x3_direct_ok(B, Hl, Wl, C, N):
        return conv3_f16x3_direct(xa, B, Hl, Wl, C, wdc, N, comb, bias, residual, ups)
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32, device=xa.device)
    geom = (ctypes.c_int32 * 12)(1, B, H, W, 2 * C, Ho, Wo, KH, KW, stride, pad, 1 if ups else 0)
    K2 = KH * KW * 2 * C
    lib.call("edadm_qgemm_f16x3", ctypes.c_void_p(xa.data_ptr()), int(K2), ctypes.c_void_p(wb.data_ptr()), int(K2),
             int(B * Ho * Wo), int(N), int(K2), ctypes.cast(geom, ctypes.c_void_p), _pf(comb), _pf(bias), _pf(residual),
             int(N), _pf(out), int(N), _stream())
    return out


def matmul_f16x3_nt(a2d, w2d, bias=None, residual=None, amax=None):
    """[M][K] . [N][K]^T (+bias) (+residual) through the same expansion (K % 16 == 0)."""
    M, K = a2d.shape
    N = w2d.shape[0]
    wb, inv_b, _ = split_f16(w2d, N, 1, K, 2, True)
    xa, _, comb = split_f16(a2d, M, 1, K, 2, False, other=inv_b, N=N, amax=amax)
    out = torch.empty(M, N, dtype=torch.float32, device=a2d.device)
    lib.call("edadm_qgemm_f16x3", ctypes.c_void_p(xa.data_ptr()), int(2 * K), ctypes.c_void_p(wb.data_ptr()), int(2 * K),
             int(M), int(N), int(2 * K), None, _pf(comb), _pf(bias), _pf(residual), int(N), _pf(out), int(N), _stream())
    return out


def qgemm_f16x3_pre(xa, lda, wb, ldw, M, N, K2, comb, out, residual=None):
    """edadm_qgemm_f16x3 on operands that are ALREADY two-term expansions (edadm_split_f16 order 2): out[M][N] = comb[n] * xa . wb^T
    (+ residual), three f16 products per K-slice.  xa / wb may be row views of larger expansions (pointer offsets)."""
    lib.call("edadm_qgemm_f16x3", ctypes.c_void_p(xa.data_ptr()), int(lda), ctypes.c_void_p(wb.data_ptr()), int(ldw), int(M), int(N),
             int(K2), None, _pf(comb), None, _pf(residual), int(N), _pf(out), int(N), _stream())
    return out


def gemm_f16x3_nt(A, lda, strideA, Bm, ldb, strideB, batch, M, N, K2):
    """C[z] = A[z] . B[z]^T over order-2 expansions (edadm_gemm_f16x3_nt): the weight gradient's split-K slabs."""
    out = torch.empty(batch, M, N, dtype=torch.float32, device=A.device)
    lib.call("edadm_gemm_f16x3_nt", ctypes.c_void_p(A.data_ptr()), int(lda), int(strideA), ctypes.c_void_p(Bm.data_ptr()),
             int(ldb), int(strideB), _pf(out), int(N), int(M * N), int(batch), int(M), int(N), int(K2), 1.0, _stream())
    return out


def im2col_f32(x_nhwc, KH, KW, stride, pad, Ho, Wo):
    B, H, W, C = x_nhwc.shape
    cols = torch.empty(B * Ho * Wo, KH * KW * C, dtype=torch.float32, device=x_nhwc.device)
    lib.call("edadm_im2col_f32", _pf(x_nhwc), _pf(cols), B, H, W, C, Ho, Wo, int(KH), int(KW), int(stride), int(pad),
             _stream())
    return cols


def col2im_f32(dcols, B, H, W, C, KH, KW, stride, pad, Ho, Wo):
    dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dcols.device)
    lib.call("edadm_col2im_f32", _pf(dcols), _pf(dx), B, H, W, C, Ho, Wo, int(KH), int(KW), int(stride), int(pad),
             _stream())
    return dx


def transpose_f32(x2d):
    """[R][C] -> [C][R]."""
    R, C = x2d.shape
    out = torch.empty(C, R, dtype=torch.float32, device=x2d.device)
    lib.call("edadm_nchw_to_nhwc", _pf(x2d), _pf(out), 1, R, C, _stream())
    return out


def sum_slabs(slabs):
    S = slabs.shape[0]
    n = slabs[0].numel()
    out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    lib.call("edadm_sum_slabs", _pf(slabs), _pf(out), n, S, _stream())
    return out


def conv2d_f32_nhwc(x, w, bias=None, residual=None, stride=1, pad=1, ups=False):
    """x [B,H,W,C] fp32, w [N,KH,KW,C] -> [B,Ho,Wo,N] (implicit GEMM on the fp32 MFMA; `ups`: conv of the nearest-2x
    upsampled input without materialising it)."""
    B, H, W, C = x.shape
    N, KH, KW, _ = w.shape
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hl + 2 * pad - KH) // stride + 1, (Wl + 2 * pad - KW) // stride + 1
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32, device=x.device)
    lib.call("edadm_conv2d_f32_nhwc", _pf(x), _pf(w), _pf(bias), _pf(residual), _pf(out), B, H, W, C, Ho, Wo, N, int(KH), int(KW),
             int(stride), int(pad), 1 if ups else 0, _stream())
    return out


def softmax_f32(s2d):
    out = torch.empty_like(s2d)
    lib.call("edadm_softmax_f32", _pf(s2d), _pf(out), s2d.shape[0], s2d.shape[1], _stream())
    return out
