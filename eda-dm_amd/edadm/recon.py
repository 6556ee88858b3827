"""The reconstruction inner loop (H1) shared by block_reconstruction / layer_reconstruction and
their qdiff_control twins (block_recon.py:13-232, layer_recon.py:13-129 of the reference).

Per iteration (block): quantised forward #1, FP forward #2 (per-module targets), quantised
forward #3 (per-module outputs, fresh masks), loss = L_block + add_loss * sum_j L_module_j,
backward, Adam on all AdaRound alphas (lr_w) and all activation deltas (lr_a), cosine to 0.

MI355X-native pieces: fake-quant fwd/bwd (K1), AdaRound fwd/bwd (K2), loss (K7), one fused Adam
launch over a flat parameter slab per group (K8), stochastic input mixing (K10) are HIP kernels;
cached activations stay in HBM; the fp32 contraction inside the block graph is edadm/contract.py (f16 three-product
or exact-fp32 MFMA GEMMs); the FP forward #2 is replaced by per-sample feature maps computed once (fp_features).
"""
import math
import os
import random
import time

import torch

from . import ops, dist as edist
from qdiff.quant_layer import QuantModule, lp_loss, _mask_rng
from qdiff.adaptive_rounding import AdaRoundQuantizer
from qdiff.utils import AttentionMap


class LinearTempDecay:
    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: int = 10, end_b: int = 2):
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b, self.end_b = start_b, end_b

    def __call__(self, t):
        if t < self.start_decay:
            return self.start_b
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, (1 - rel_t))


class LossFunction:
    """rec loss (+ the AdaRound rounding regulariser, which the reference switches off with
    round_loss='none', block_recon.py:119, but keeps in the API)."""

    def __init__(self, block, round_loss='relaxation', weight=1., rec_loss='mse', max_count=2000, b_range=(10, 2),
                 decay_start=0.0, warmup=0.0, p=2.):
        self.block = self.layer = block
        self.round_loss, self.weight, self.rec_loss = round_loss, weight, rec_loss
        self.loss_start = max_count * warmup
        self.p, self.iters = p, max_count
        self.temp_decay = LinearTempDecay(max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0

    def __call__(self, pred, tgt, grad=None):
        self.count += 1
        if self.rec_loss == 'mse':
            rec = lp_loss(pred, tgt, p=self.p)
        elif self.rec_loss == 'fisher_diag':
            rec = ((pred - tgt).pow(2) * grad.pow(2)).sum(1).mean()
        elif self.rec_loss == 'fisher_full':
            a, g = (pred - tgt).abs(), grad.abs()
            rec = (torch.sum(a * g, (1, 2, 3)).view(-1, 1, 1, 1) * a * g).mean() / 100
        else:
            raise ValueError('Not supported reconstruction loss function: {}'.format(self.rec_loss))
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round_loss == 'none':
            rnd = 0
        elif self.round_loss == 'relaxation':
            rnd = 0
            mods = [self.block] if isinstance(self.block, QuantModule) else \
                [m for m in self.block.modules() if isinstance(m, QuantModule)]
            for m in mods:
                rv = m.weight_quantizer.get_soft_targets()
                rnd = rnd + self.weight * (1 - ((rv - .5).abs() * 2).pow(b)).sum()
        else:
            raise NotImplementedError
        return rec + rnd


class FusedAdam:
    """torch.optim.Adam + CosineAnnealingLR(T_max, eta_min=0) for a parameter group, as ONE HIP
    launch per step: the parameters are re-homed as views of a flat slab (so are their grads)."""

    def __init__(self, params, lr, t_max, betas=(0.9, 0.999)):
        self.params = list(params)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        self.grad_views = []
        for p in self.params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)
            self.grad_views.append(self.grad[o:o + k].view(p.shape))
            p.grad = None
            o += k
        self.lr0, self.t_max, self.betas = lr, t_max, betas
        self.t = 0
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)

    def zero_grad(self):
        # .grad = None: autograd then TAKES each incoming gradient tensor instead of adding it into a zeroed one -- an add
        # kernel per parameter and iteration otherwise (87 launches for a transformer block, ~1 ms of its 19 ms iteration);
        # launch() collects them into the slab with one multi-tensor copy.  A parameter reached by two paths still
        # accumulates (the second contribution is added to the first).
        for p in self.params:
            p.grad = None

    def prepare(self):
        """host half of a step: the step's learning rate and bias corrections into the device-side hyper vector"""
        lr = self.lr0 * (1 + math.cos(math.pi * self.t / self.t_max)) / 2
        self.t += 1
        b1, b2 = self.betas
        self.hyper.copy_(torch.tensor([lr / (1 - b1 ** self.t), math.sqrt(1 - b2 ** self.t), b1, b2]))

    def schedule(self, iters):
        """the hyper vectors of the next `iters` steps as one device table [iters][4] (one host-to-device copy per unit): the loop
        then feeds a step with an asynchronous device-to-device row copy and never blocks on the stream"""
        rows = []
        b1, b2 = self.betas
        for k in range(iters):
            t = self.t + k
            lr = self.lr0 * (1 + math.cos(math.pi * t / self.t_max)) / 2
            rows.append([lr / (1 - b1 ** (t + 1)), math.sqrt(1 - b2 ** (t + 1)), b1, b2])
        host = torch.tensor(rows, dtype=torch.float32)
        if self.hyper.is_cuda:                               # pinned staging: the copy is asynchronous, the unit's set-up does not block
            host = host.pin_memory()
        self.table, self.t0 = host.to(self.hyper.device, non_blocking=True), self.t
        self._table_host = host                              # keeps the pinned buffer alive until the copy has run
        return self.table

    def prepare_row(self):
        self.hyper.copy_(self.table[self.t - self.t0], non_blocking=True)
        self.t += 1

    def launch(self):
        """device half: gradients into the slab (one multi-tensor copy), one kernel over the flat slab (capturable: reads
        the hyper vector from device memory)"""
        self.collect()
        self.apply()

    def collect(self):
        """gradients into the slab (one multi-tensor copy)"""
        have = [(v, p.grad) for v, p in zip(self.grad_views, self.params) if p.grad is not None]
        if len(have) != len(self.params):
            self.grad.zero_()                                # a parameter without gradient this iteration: a zero one
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def apply(self):
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.hyper)

    # ---- data-parallel iterations (reconstruct(): DP_LOOP): every rank holds the partial gradient of its rows in `grad`
    def dp_gather(self, rank, world):
        """partials of all ranks into dp_buf [world][n] (RCCL: one all_gather_into_tensor over the direct xGMI links; gloo: staged
        through the host, CPU tests and the two-processes-on-one-GPU test)"""
        import torch.distributed as dist
        if getattr(self, "dp_buf", None) is None:
            self.dp_buf = torch.empty(world, self.grad.numel(), dtype=torch.float32, device=self.grad.device)
        if dist.get_backend() == "gloo":
            parts = [torch.empty(self.grad.numel(), dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, self.grad.detach().cpu())
            for r in range(world):
                self.dp_buf[r].copy_(parts[r])
        else:
            dist.all_gather_into_tensor(self.dp_buf.reshape(-1), self.grad)
        DP_STATS["gather_bytes"] += self.dp_buf.numel() * 4

    def dp_sum(self):
        """the full-batch gradient = the partials added in RANK ORDER: one fixed order of fp32 sums for a given world size, the same on
        every rank (they hold the same dp_buf), so the replicas stay bit-identical without waiting for the end-of-unit broadcast"""
        self.grad.copy_(self.dp_buf[0])
        for r in range(1, self.dp_buf.shape[0]):
            self.grad.add_(self.dp_buf[r])

    def step(self):
        self.prepare()
        self.launch()


# Data-parallel reconstruction iterations (block_recon.py:133-217): with several ranks, a unit whose rows are large
# (DP_MIN_POSITIONS feature-map positions / tokens per row: the 64 x 64 and 32 x 32 levels of LDM-4) splits every minibatch over the
# ranks -- rank r forwards / backwards rows r, r + N, ... of the SAME drawn minibatch with the loss scaled by 1 / N, the partial
# d loss / d alpha and d loss / d delta slabs are all-gathered and added in rank order, and every rank takes the same Adam step.
# Smaller units stay replicated (their iterations are bound by launch latency, not by work).  One rank: nothing changes.
DP_LOOP = True
DP_MIN_POSITIONS = 1024
DP_STATS = {"units": 0, "iters": 0, "gather_bytes": 0}


def _as_param(q):
    q.delta = torch.nn.Parameter(q.delta.detach().clone())


def _attention_quantizers(module, control=True):
    """q, k, v, w activation quantizers owned by an attention wrapper, in the reference's order.  The transformer
    block's eight are trainables of the conditional walk only (qdiff_control/block_recon.py:82-110); qdiff/block_recon.py:66-93
    knows QuantAttentionBlock and QuantAttnBlock, qdiff_control/block_recon.py:68-110 QuantAttnBlock and the transformer block."""
    from qdiff.quant_block import QuantAttnBlock, QuantAttentionBlock, QuantBasicTransformerBlock
    names = ("act_quantizer_q", "act_quantizer_k", "act_quantizer_v", "act_quantizer_w")
    if isinstance(module, QuantAttentionBlock) and not control:
        qk, smv = module.attention.qkv_matmul, module.attention.smv_matmul
        return [qk.act_quantizer_q, qk.act_quantizer_k, smv.act_quantizer_v, smv.act_quantizer_w]
    if isinstance(module, QuantAttnBlock):
        return [getattr(module, n) for n in names]
    if isinstance(module, QuantBasicTransformerBlock) and control:
        return [getattr(a, n) for a in (module.attn1, module.attn2) for n in names]
    return []


# bench.py sets TIMING = {"iter_s": 0.0, "iters": 0}: wall time of every iteration after the first of each unit
TIMING = None
# iterations from which a unit's loop replays a HIP graph of one iteration (0 / huge = always eager), and the eager iterations
# in front of the capture (lazy initialisation, allocator warm-up)
GRAPH_MIN_ITERS = 32
GRAPH_WARMUP = 2
# HBM budget of the per-sample FP feature maps of the fine-grained loss (fp_features), and a force flag for short measuring
# runs (bench.py) in which a sample is drawn less than twice
FP_FEAT_GB = 56.0
FP_FEAT_FORCE = False
# The two quantised forwards of a block iteration (block_recon.py:154 for the block-output loss, :167 for the per-module outputs,
# each with fresh quantizer masks) take the SAME inputs: they run as ONE forward over the batch [x | x] -- the counter RNG gives
# the two halves independent masks, rows are independent everywhere in these graphs -- so every weight-side pass (AdaRound, the
# filters' f16 expansion, the weight-gradient GEMM) runs once instead of twice and the small levels launch half as many, twice
# as large kernels.  Units whose wrapper checkpoints (QuantAttentionBlock; ResBlocks / transformer blocks with the flag on) keep
# the two separate forwards: the reference recomputes only the first of them in backward (see edadm/nets/ldm_unet.py).
BATCH_FORWARDS = True
BATCH_FORWARDS_BELOW_PIXELS = 4096
STATE = {"batched": False}           # whether the unit being reconstructed runs the batched form (read by the parity tests' mask replay)
# parity tests: a callable (cur_inp) -> uniforms that replace the in-kernel RNG of the input mix (block_recon.py:141-145), the
# counterpart of UniformAffineQuantizer.injected_uniform; None = the counter RNG keyed by (seed, element)
INJECT_MIX_UNIFORM = None


# The fine-grained loss terms of a batched block iteration as gradient injections (csrc/elem.hip k_lp_inject): a hooked module's
# output passes through an identity whose backward adds that module's loss gradient (rows [nb, 2 nb) of the batched [x | x] tensor
# against the cached FP feature rows of the drawn samples) to the gradient arriving from the next layer -- one pass instead of
# autograd's gather, loss backward, zero-padded slice gradient and accumulation add.  Same arithmetic per element, same bits; the
# loss VALUE of these terms is never formed (nothing reads it).  Units that run the two quantised forwards separately keep the plain form.
INJECT_MODULE_LOSS = True


class _InjectLp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, spec):
        ctx.spec = spec
        ctx.save_for_backward(y)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        sp = ctx.spec
        ym, cl = ops.mem_view(y)
        store, idx = sp["store"], sp["idx"]
        if isinstance(store, RowsNHWC):
            rows = store.t if cl else None
        else:
            rows = store if (not cl and store.is_contiguous()) else None
        if rows is None:                                     # layouts differ (not the production path): gather, then bring to y's order
            rows, idx = ops.mem_like(store[sp["idx"]], cl).contiguous(), None
        nrows = sp["nrows"]
        inv = 1.0 / (nrows * (ym.numel() // ym.shape[0]) / y.shape[1])
        g = ops.lp_loss_inject(ops.mem_like(gy, cl), ym, rows, idx, sp["row0"], nrows, inv, sp["gscale"])
        return ops.mem_restore(g, cl), None


class _LossHook(AttentionMap):
    """AttentionMap (utils.py:12-24) that can also route the module's output through _InjectLp (armed per iteration by reconstruct)"""
    inject = None

    def hook_fn(self, module, input, output):
        if self.inject is not None:
            self.out = self.feature = None
            return _InjectLp.apply(output, self.inject)
        self.out = output
        self.feature = input


class RowsNHWC:
    """Calibration rows of a feature map [N, C, H, W] kept in NHWC memory order (edadm/contract.py CHANNELS_LAST): a minibatch
    gather `rows[idx]` comes out as a channels_last tensor, the layout the unit's convolutions, GroupNorms and element-wise
    kernels run in -- no layout pass between the cache and the loss."""

    def __init__(self, t=None, store=None):
        self.t = store if store is not None else t.permute(0, 2, 3, 1).contiguous()     # a view when t already is channels_last

    def __getitem__(self, idx):
        return self.t[idx].permute(0, 3, 1, 2)

    @property
    def shape(self):
        n, h, w, c = self.t.shape
        return torch.Size((n, c, h, w))

    def size(self, d):
        return self.shape[d]

    def dim(self):
        return 4

    @property
    def is_cuda(self):
        return self.t.is_cuda

    @property
    def device(self):
        return self.t.device


def _rows(t):
    from . import contract
    if contract.CHANNELS_LAST and torch.is_tensor(t) and t.dim() == 4 and t.is_cuda and t.dtype == torch.float32 \
            and t.shape[1] % 4 == 0:
        return RowsNHWC(t)
    return t


def fp_features(unit, hooks, cached_inps, resblock, sz, chunk, budget_bytes):
    """The FP feature maps of the fine-grained loss (block_recon.py:170-178: an FP forward of the block on `cur_sym`
    every iteration) are a pure function of the calibration sample, and each of the `sz` cached samples is drawn
    iters * batch / sz times (16x at the shipped setting): one FP forward per sample, kept in HBM, replaces the
    per-iteration FP forward when the maps of all samples fit `budget_bytes`.  Returns a list (one [sz, ...] tensor per
    hooked module but the last, which the loss skips) or None when they do not fit."""
    if budget_bytes <= 0 or len(hooks) < 2:
        return None
    sym = cached_inps[1]
    feats = None
    unit.set_quant_state(False, False)
    with torch.no_grad():
        for lo in range(0, sz, chunk):
            hi = min(sz, lo + chunk)
            unit(*((sym[0][lo:hi], sym[1][lo:hi]) if resblock else (sym[lo:hi],)))
            outs = [h.out for h in hooks[:-1]]
            if feats is None:
                per_row = sum(o[0].numel() * o.element_size() for o in outs)
                if per_row * sz > budget_bytes:
                    return None
                from . import contract
                nhwc = [contract.CHANNELS_LAST and o.dim() == 4 and o.shape[1] % 4 == 0 for o in outs]
                feats = [torch.empty((sz, o.shape[2], o.shape[3], o.shape[1]) if cl else (sz,) + tuple(o.shape[1:]), dtype=o.dtype,
                                     device=o.device) for o, cl in zip(outs, nhwc)]
            for f, o, cl in zip(feats, outs, nhwc):
                (f.permute(0, 3, 1, 2) if cl else f)[lo:hi] = o
    return [RowsNHWC(store=f) if cl else f for f, cl in zip(feats, nhwc)]


def reconstruct(model, unit, cali_data, *, is_block, batch_size=32, iters=20000, weight=0.01, opt_mode='mse',
                asym=False, b_range=(20, 2), warmup=0.0, act_quant=False, lr_a=4e-5, lr_w=1e-2, p=2.0,
                input_prob=1.0, keep_gpu=True, recon_w=False, recon_a=False, add_loss=0.0, cache_batch=32,
                batch_transform=None, control=False, save_fn=None):
    from qdiff.quant_block import BaseQuantBlock
    from qdiff.data_utils import save_inp_oup_data
    save_fn = save_fn or save_inp_oup_data
    # the device-side mask epoch counts graph replays; every unit starts from 0 so that a calibration is a function of
    # (random.seed, seed_mask_rng) alone, however many replays earlier units ran (edadm.h: epoch 0 leaves seeds as passed)
    rank, world = edist.world()
    ops.rng_epoch(0)
    unit.set_quant_state(True, act_quant)
    round_mode = 'learned_hard_sigmoid'
    hooks, w_para, a_para, trained_aq = [], [], [], []
    modules = list(unit.modules()) if is_block else [unit]
    for module in modules:
        if isinstance(module, QuantModule):
            if is_block:
                hooks.append(_LossHook(module))
            if module.split == 0 or (control and not is_block):
                module.weight_quantizer = AdaRoundQuantizer(uaq=module.weight_quantizer, round_mode=round_mode,
                                                            weight_tensor=module.org_weight.data)
                wqs = [module.weight_quantizer]
            else:
                module.weight_quantizer = AdaRoundQuantizer(
                    uaq=module.weight_quantizer, round_mode=round_mode,
                    weight_tensor=module.org_weight.data[:, :module.split, ...])
                module.weight_quantizer_0 = AdaRoundQuantizer(
                    uaq=module.weight_quantizer_0, round_mode=round_mode,
                    weight_tensor=module.org_weight.data[:, module.split:, ...])
                wqs = [module.weight_quantizer, module.weight_quantizer_0]
            if recon_w:
                for q in wqs:
                    q.soft_targets = True
                    w_para.append(q.alpha)
        if isinstance(module, (QuantModule, BaseQuantBlock)):
            aqs = _attention_quantizers(module, control) if act_quant else []
            if act_quant and module.act_quantizer.delta is not None:
                aqs = aqs + [module.act_quantizer]
                if module.split != 0 and not (control and not is_block):
                    aqs.append(module.act_quantizer_0)
            for q in aqs:
                _as_param(q)
                if recon_a:
                    a_para.append(q.delta)
                    q.is_training = True
                    trained_aq.append(q)
    w_opt = FusedAdam(w_para, lr_w, iters) if w_para else None
    a_opt = FusedAdam(a_para, lr_a, iters) if a_para else None
    loss_func = LossFunction(unit, round_loss='none', weight=weight, max_count=iters, rec_loss=opt_mode,
                             b_range=b_range, decay_start=0, warmup=warmup, p=p)

    resblock, cached_inps, cached_outs = save_fn(model, unit, cali_data, asym, act_quant, batch_size=cache_batch,
                                                 input_prob=True, keep_gpu=keep_gpu)
    # feature-map caches go to NHWC memory order once; minibatch gathers then feed the unit channels_last tensors
    cached_outs = _rows(cached_outs)
    if resblock:
        cached_inps = tuple((_rows(pair[0]), pair[1]) for pair in cached_inps)
    else:
        cached_inps = tuple(_rows(c) for c in cached_inps)
    sz = cached_outs.size(0)
    model.block_count = model.block_count + 1
    # a frozen int8 executor was compiled from the pre-reconstruction parameters: drop it (freeze() again after the walk)
    model.engine = None
    feats = None
    # worth it when every sample is drawn more than once (bench.py forces it on its short run and scales the time)
    if is_block and hooks and (iters * batch_size >= 2 * sz or FP_FEAT_FORCE):
        if TIMING is not None:
            torch.cuda.synchronize()
            _t_feat = time.time()
        tr = getattr(model, "_fp_trace", None)               # the walk's HBM scale (qdiff.data_utils.hbm_scale), 1 without a trace
        feats = fp_features(unit, hooks, cached_inps, resblock, sz, batch_size,
                            int(FP_FEAT_GB * (tr.scale if tr is not None else 1.0) * (1 << 30)))
        unit.set_quant_state(True, act_quant)
        if TIMING is not None:
            torch.cuda.synchronize()
            TIMING["feat_s"] = TIMING.get("feat_s", 0.0) + time.time() - _t_feat
            TIMING["feat_units"] = TIMING.get("feat_units", 0) + (feats is not None)
    # One iteration is ~400-2000 kernel launches; at the 16x16 and 8x8 levels they are so short that the loop is bound by
    # the host's launch rate (5 us per launch through Python + autograd), not by the GPU.  From the third iteration on the
    # whole iteration -- gathers of the drawn minibatch, the three forwards, autograd's backward, both Adam launches -- is
    # therefore ONE HIP-graph replay: the minibatch indices go through a static device buffer, the learning-rate vector is
    # written before the replay, and the stochastic masks (kernel seed ARGUMENTS are frozen in a graph) come from a device-side
    # epoch word bumped at the head of every replay (edadm_rng_epoch).  Same kernels, same order, same bits as eager.
    use_graph = iters >= GRAPH_MIN_ITERS and cached_outs.is_cuda
    idx_buf = torch.zeros(batch_size, dtype=torch.long, device=cached_outs.device)
    graph = None
    t_first = GRAPH_WARMUP + 1 if use_graph else 1

    batched = BATCH_FORWARDS and is_block and bool(hooks) and not any(
        (type(m).__name__ == "QuantAttentionBlock") or (type(m).__name__ == "QuantResBlock" and m.use_checkpoint)
        or (type(m).__name__ == "QuantBasicTransformerBlock" and m.checkpoint) for m in unit.modules())

    # ... and only below 64 x 64: there the doubled activations (and their concatenation) cost more than the halved weight-side
    # work and launches save (measured on LDM-4: 8x8 units 7.5 -> 5.2 ms per iteration, 16x16 and 32x32 8.9 -> 7.4, but 64x64 7.5 -> 9.2)
    if batched and cached_outs.dim() == 4 and cached_outs.shape[-1] * cached_outs.shape[-2] >= BATCH_FORWARDS_BELOW_PIXELS:
        batched = False
    STATE["batched"] = batched                                # mirror of the LAST unit for the parity tests' mask replay ...
    unit.recon_batched = batched                              # ... the unit carries its own flag (interleaved reconstructions)
    positions = (cached_outs.shape[-1] * cached_outs.shape[-2]) if cached_outs.dim() == 4 else (cached_outs.shape[1] if cached_outs.dim() == 3 else 1)
    dp = bool(DP_LOOP and world > 1 and batch_size % world == 0 and positions >= DP_MIN_POSITIONS and (w_opt or a_opt) and cached_outs.is_cuda)
    unit.recon_dp = dp
    if dp:
        # every rank draws its own masks: the device-side epoch (mixed into every kernel seed) starts from a rank-specific value
        ops.rng_epoch(rank << 20)
        DP_STATS["units"] += 1
        DP_STATS["iters"] += iters

    # gradient injection of the per-module loss terms: batched units whose FP feature rows are cached
    inject = bool(INJECT_MODULE_LOSS and batched and feats is not None and len(hooks) >= 2)
    gscale = (torch.full((1,), float(add_loss), dtype=torch.float32, device=idx_buf.device) / (world if dp else 1)) if inject else None
    unit.recon_inject = inject

    def body(apply=True):
        idx_t = idx_buf[rank::world].contiguous() if dp else idx_buf     # data parallel: rows r, r + N, ... of the drawn minibatch
        cur_out = cached_outs[idx_t]
        if resblock:
            cur_inp, cur_sym = cached_inps[0][0][idx_t], cached_inps[1][0][idx_t]
            temb_inp, temb_sym = cached_inps[0][1][idx_t], cached_inps[1][1][idx_t]
        else:
            cur_inp, cur_sym = cached_inps[0][idx_t], cached_inps[1][idx_t]
        if input_prob < 1.0:
            inp_m, cl = ops.mem_view(cur_inp)                # element-wise: in the rows' memory order
            if INJECT_MIX_UNIFORM is not None:
                mixed = ops.mix_where(inp_m, ops.mem_like(cur_sym, cl), input_prob,
                                      u=ops.mem_like(INJECT_MIX_UNIFORM(cur_inp), cl))
            else:
                mixed = ops.mix_where(inp_m, ops.mem_like(cur_sym, cl), input_prob, seed=_mask_rng.getrandbits(62))
            cur_inp = ops.mem_restore(mixed, cl)
        elif is_block:
            cur_inp = cur_sym                 # block_recon.py:144-145 (the layer loop keeps cur_inp)
        for o in (w_opt, a_opt):
            if o:
                o.zero_grad()
        args_q = (cur_inp, temb_inp) if resblock else (cur_inp,)
        nb = cur_out.shape[0]
        if inject:
            for h, f in zip(hooks[:-1], feats):
                h.inject = {"store": f, "idx": idx_t, "row0": nb, "nrows": nb, "gscale": gscale}
            try:
                out_quant = unit(*(torch.cat([a, a]) for a in args_q))[:nb]
            finally:
                for h in hooks:
                    h.inject = None
        elif batched:
            out_quant = unit(*(torch.cat([a, a]) for a in args_q))[:nb]
            module_q = [h.out[nb:] for h in hooks]
        else:
            out_quant = unit(*args_q)
        m_loss = 0.0
        if is_block and hooks and not inject:
            if feats is not None:
                module_r = [f[idx_t] for f in feats] + [None]
            else:
                args_fp = (cur_sym, temb_sym) if resblock else (cur_sym,)
                unit.set_quant_state(False, False)
                with torch.no_grad():
                    unit(*args_fp)
                module_r = [h.out for h in hooks]
                unit.set_quant_state(True, act_quant)
            if not batched:
                unit(*args_q)
                module_q = [h.out for h in hooks]
            for j in range(len(module_r) - 1):
                m_loss = m_loss + lp_loss(module_q[j], module_r[j], p=2)
        loss = loss_func(out_quant, cur_out) + add_loss * m_loss
        if dp:
            loss = loss / world                                  # both terms are means over the rows: the partials add up to the minibatch's
        loss.backward()
        for o in (w_opt, a_opt):
            if o:
                if apply:
                    o.launch()
                else:
                    o.collect()
        for h in hooks:                       # tensors of this iteration must not outlive it (they would pin the autograd
            h.out = h.feature = None          # graph of a captured iteration past the end of the capture)

    # Everything the host contributes to an iteration -- the minibatch draw, the step's learning rate and bias corrections -- is
    # drawn / computed for ALL iterations up front (same `random.sample` sequence) and sits in device tables: an iteration is two
    # or three asynchronous device-to-device row copies + one graph replay, the host runs ahead and the stream never drains
    # between iterations (a pageable host-to-device copy per iteration blocked on the previous replay).
    # every minibatch draw of the unit up front (the reference's random.sample stream, block_recon.py:137), in chunks of 1024
    # iterations through pinned memory: bounded host memory for the reference's default of 20000 iterations, no blocking copy
    IDX_CHUNK = 1024
    idx_host, idx_dev = [], []
    for c0 in range(0, iters, IDX_CHUNK):
        h = torch.tensor([random.sample(range(sz), batch_size) for _ in range(min(IDX_CHUNK, iters - c0))], dtype=torch.long).reshape(-1, batch_size)
        if idx_buf.is_cuda:
            h = h.pin_memory()
        idx_host.append(h)
        idx_dev.append(h.to(idx_buf.device, non_blocking=True))

    class _Idx:
        def __getitem__(self, it):
            return idx_dev[it // IDX_CHUNK][it % IDX_CHUNK]
    idx_all = _Idx()
    for o in (w_opt, a_opt):
        if o:
            o.schedule(iters)
    # The loop trains alphas and step sizes only (block_recon.py:44-117), but the FP model's biases (and any other parameter of the unit)
    # still ask autograd for gradients nobody reads: a full-tensor reduction per biased layer and iteration (4 x 67 us in a transformer
    # block at 32 x 32).  They are switched off for the duration of the loop; the trained parameters' gradients do not depend on them.
    trained = {id(p) for p in w_para + a_para}
    unread = [p for p in unit.parameters() if p.requires_grad and id(p) not in trained]
    for p in unread:
        p.requires_grad_(False)
    try:
        for it in range(iters):
            if TIMING is not None and it == t_first:           # steady-state iterations only (bench.py): the first ones carry
                torch.cuda.synchronize()                       # allocator warm-up, lazy initialisation and the graph capture
                _t_steady = time.time()
            idx_buf.copy_(idx_all[it], non_blocking=True)
            for o in (w_opt, a_opt):
                if o:
                    o.prepare_row()
            if dp:
                # forward / backward of this rank's rows (graph A once captured) -> all-gather of the partial slabs (eager: a collective)
                # -> rank-ordered sum + Adam (graph B)
                if graph is not None:
                    graph[0].replay()
                elif use_graph and it >= GRAPH_WARMUP:
                    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    from . import contract
                    f0 = contract.FLOPS[0]
                    with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                        ops.rng_epoch(1, add=True)
                        body(apply=False)
                    if TIMING is not None:
                        TIMING["flops"] = TIMING.get("flops", 0.0) + (contract.FLOPS[0] - f0) * (iters - t_first)
                    with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                        for o in (w_opt, a_opt):
                            if o:
                                o.dp_sum()
                                o.apply()
                    graph = (ga, gb)
                    graph[0].replay()
                else:
                    body(apply=False)
                for o in (w_opt, a_opt):
                    if o:
                        o.dp_gather(rank, world)
                if graph is not None:
                    graph[1].replay()
                else:
                    for o in (w_opt, a_opt):
                        if o:
                            o.dp_sum()
                            o.apply()
                continue
            if graph is not None:
                graph.replay()
                continue
            if use_graph and it >= GRAPH_WARMUP:
                graph = torch.cuda.CUDAGraph()
                # thread_local: with several ranks the RCCL watchdog thread polls events while this thread captures; in the default
                # (global) mode any such call from another thread invalidates the capture
                from . import contract
                f0 = contract.FLOPS[0]
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    ops.rng_epoch(1, add=True)
                    body()
                if TIMING is not None:                         # executed fp32-equivalent flops of one iteration of this unit
                    TIMING["flops"] = TIMING.get("flops", 0.0) + (contract.FLOPS[0] - f0) * (iters - t_first)
                graph.replay()                                 # capture does not execute: this runs iteration `it`
                continue
            body()
    finally:
        for p in unread:
            p.requires_grad_(True)
    del graph
    if TIMING is not None and iters > t_first:
        torch.cuda.synchronize()
        TIMING["iter_s"] += time.time() - _t_steady
        TIMING["iters"] += iters - t_first
        TIMING["graphed_units"] = TIMING.get("graphed_units", 0) + int(use_graph)
        TIMING.setdefault("per_unit", []).append((type(unit).__name__, sum(p.numel() for p in w_para),
                                                  1e3 * (time.time() - _t_steady) / (iters - t_first)))
        TIMING.setdefault("per_unit_positions", []).append(int(positions))
    for module in modules:
        if isinstance(module, QuantModule):
            module.weight_quantizer.soft_targets = False
            module.act_quantizer.is_training = False
            if module.split != 0 and hasattr(module, "weight_quantizer_0"):
                if isinstance(module.weight_quantizer_0, AdaRoundQuantizer):
                    module.weight_quantizer_0.soft_targets = False
                module.act_quantizer_0.is_training = False
    for q in trained_aq:
        q.is_training = False
    for h in hooks:
        h.remove()
    # replicas stay bit-identical: rank 0's learned parameters win (no-op on one GPU)
    edist.broadcast_params([p for p in w_para + a_para])
