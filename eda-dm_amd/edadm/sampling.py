"""Quantised DDIM sampling loop (H2 stepping): classifier-free-guided DDIM over the int8 executor
(ldm/models/diffusion/ddim_control.py:167-254 of the reference), with the UNet forward captured
once into a HIP graph and replayed per step, and the CFG combine + x_{t-1} update as one kernel
(edadm_ddim_step, K9)."""
import weakref

import numpy as np
import torch

from . import ops
from .schedule import make_beta_schedule, make_ddim_timesteps, make_ddim_sampling_parameters, ddim_coef_table


class GraphedUNet:
    """Static-shape HIP-graph replay of `engine(x, t, ctx)`.  With a one-token context the cross-attention branches
    are a function of the context alone (Engine.context_branches): they get their own graph, replayed when the
    context changes -- once per sample batch -- and the per-step graph only adds their vectors."""

    def __init__(self, engine, x, t, ctx, warmup=2, timesteps=None, cfg_pair=False, capture_stream=None):
        """timesteps: the schedule of the run (one int per step, in call order).  With it the time-embedding path of all
        steps is one graph replayed by begin() at the start of a run (Engine.emb_tables), and a step -- called with
        its index -- copies its row of that table instead of recomputing 2 + 2 x 22 tiny launches."""
        # cfg_pair: the caller always passes x = [img, img] (a classifier-free-guidance pair): the context-independent
        # prefix of the network runs once for both halves (Engine.cfg_pair)
        # capture_stream: the stream the three graphs are captured on (torch's default is one shared capture stream for every graph
        # of the process).  The library's reduction workspaces are per (device, stream, scope), and the captures below run inside a
        # workspace scope of this object: the graphs of two GraphedUNets never share a workspace, so they can be REPLAYED
        # concurrently (several sample batches in flight, InFlightSampler below) whatever streams were involved
        gkw = {} if capture_stream is None else {"stream": capture_stream}
        # the scope's workspaces live exactly as long as this object (and with it the graphs that have their addresses baked in)
        self._scope = ops.new_scope_token()
        weakref.finalize(self, ops.release_scope, self._scope)
        self.engine = engine
        self.x, self.t = x.clone(), t.clone()
        self.ctx = None if ctx is None else ctx.clone()
        self._ctx_src, self._ctx_ver = None, -1
        rows = x.shape[0]
        pair = bool(cfg_pair) and hasattr(engine, "cfg_pair")
        if pair:
            engine.cfg_pair = True
        self.emb_graph = self.emb_tab = self.emb_stage = None
        ts_all = None
        if timesteps is not None and hasattr(engine, "emb_tables"):
            ts_all = torch.tensor(np.repeat(np.asarray(timesteps, dtype=np.int64), rows), device=x.device)
        self.ts_all = ts_all                          # an input of the table graph: must outlive this constructor
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                engine.ctx_r = engine.context_branches(self.ctx) if hasattr(engine, "context_branches") else None
                tabs = engine.emb_tables(ts_all, len(timesteps)) if ts_all is not None else None
                self.out = engine(self.x, self.t, self.ctx)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.ctx_graph, self.ctx_r = None, None
        if getattr(engine, "ctx_r", None) is not None:
            self.ctx_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.ctx_graph, **gkw), ops.workspace_scope(self._scope):
                self.ctx_r = engine.context_branches(self.ctx)
        engine.ctx_r = self.ctx_r
        emb_r = None
        if ts_all is not None and tabs is not None:
            self.emb_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.emb_graph, **gkw), ops.workspace_scope(self._scope):
                self.emb_tab, layout = engine.emb_tables(ts_all, len(timesteps))
            self.emb_stage = torch.empty_like(self.emb_tab[0])
            emb_r = {k: self.emb_stage[off:off + rows * n].view(rows, n) for k, (off, n) in layout.items()}
        engine.emb_r = emb_r
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, **gkw), ops.workspace_scope(self._scope):
                self.out = engine(self.x, self.t, self.ctx)
        finally:
            engine.ctx_r = None                       # eager calls of the engine keep evaluating both per call
            engine.emb_r = None
            if pair:
                engine.cfg_pair = False
        if self.ctx_graph is not None:
            self.ctx_graph.replay()                   # capture does not execute: make the vectors match self.ctx now
        self.begin()

    def begin(self):
        """Start of a sampling run: the time-embedding table of the schedule (a no-op without a schedule)."""
        if self.emb_graph is not None:
            self.emb_graph.replay()

    def __call__(self, x, t, ctx=None, step=None):
        self.x.copy_(x)
        self.t.copy_(t)
        if self.emb_graph is not None:
            if step is None:
                raise ValueError("this graph was built for a fixed schedule: pass the step index")
            self.emb_stage.copy_(self.emb_tab[step])
        if ctx is not None and (ctx is not self._ctx_src or ctx._version != self._ctx_ver):
            self.ctx.copy_(ctx)
            self._ctx_src, self._ctx_ver = ctx, ctx._version
            if self.ctx_graph is not None:
                self.ctx_graph.replay()
        self.graph.replay()
        return self.out


class DDIMLoop:
    """S-step DDIM with classifier-free guidance on a frozen QuantModel."""

    def __init__(self, engine, shape, batch, steps=20, eta=0.0, scale=3.0, linear_start=0.0015, linear_end=0.0195,
                 n_timesteps=1000, context_shape=None, use_graph=True, device="cuda", capture_stream=None):
        self.engine, self.shape, self.batch, self.scale, self.eta = engine, tuple(shape), batch, scale, eta
        betas = make_beta_schedule("linear", n_timesteps, linear_start, linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.ddim_timesteps = make_ddim_timesteps("uniform", steps, n_timesteps)
        sig, al, alp = make_ddim_sampling_parameters(ac.astype(np.float32), self.ddim_timesteps, eta)
        self.coef = torch.tensor(ddim_coef_table(al, alp, sig), device=device)
        self.cfg = scale != 1.0
        rows = batch * (2 if self.cfg else 1)
        self.unet = engine
        if use_graph:
            x0 = torch.zeros((rows,) + self.shape, device=device)
            t0 = torch.zeros(rows, dtype=torch.long, device=device)
            c0 = None if context_shape is None else torch.zeros((rows,) + tuple(context_shape), device=device)
            self.unet = GraphedUNet(engine, x0, t0, c0, timesteps=self._schedule(), cfg_pair=self.cfg, capture_stream=capture_stream)

    def _schedule(self):
        """timesteps in call order, or None when the loop does not walk a fixed list (PLMS: extra evaluations)"""
        return [int(v) for v in np.flip(self.ddim_timesteps)]

    @torch.no_grad()
    def sample(self, x_T, cond=None, uncond=None):
        """x_T [B,C,H,W]; cond / uncond [B,L,D].  Returns x_0 latents."""
        B = x_T.shape[0]
        img = x_T
        ctx = None
        if cond is not None:
            ctx = torch.cat([uncond, cond]) if self.cfg else cond
        total = self.ddim_timesteps.shape[0]
        graphed = isinstance(self.unet, GraphedUNet)
        if graphed:
            self.unet.begin()
        for i, step in enumerate(np.flip(self.ddim_timesteps)):
            index = total - i - 1
            ts = torch.full((B * (2 if self.cfg else 1),), int(step), device=img.device, dtype=torch.long)
            x_in = torch.cat([img, img]) if self.cfg else img
            e = self.unet(x_in, ts, ctx, step=i) if graphed else self.unet(x_in, ts, ctx)
            coef = self.coef[index:index + 1].expand(B, 5).contiguous()
            if self.cfg:
                img = ops.ddim_step(img.contiguous(), e[B:], e[:B], self.scale, coef)
            else:
                img = ops.ddim_step(img.contiguous(), e, None, 1.0, coef)
        return img


class InFlightSampler:
    """Several independent sample batches IN FLIGHT on one GPU: n loops over the same frozen engine (each with its own captured
    graphs and static buffers, captured on its own stream), batch i sampled on stream i mod n.  A UNet call is ~500 dependent
    launches, a third of them at the 16 x 16 / 8 x 8 levels where a launch does not fill 256 CUs, and every dependent hand-off leaves
    the chip idle for a few microseconds: a second batch's launches fill those holes.  The batches are independent (the loop of
    sample_diffusion_ldm_imagenet.py:215-249 has no carried state), so each batch's bits are those of the serial loop."""

    def __init__(self, make_loop, n=2, device="cuda"):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.loops = [make_loop(torch.cuda.Stream(device=device)) for _ in range(n)]       # argument: the capture stream
        self.k = 0

    def submit(self, x_T, cond=None, uncond=None):
        """Enqueue one batch on the next stream; returns (latents, stream) -- the latents are ready when the stream reaches here."""
        i = self.k % len(self.loops)
        self.k += 1
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream())
        for t in (x_T, cond, uncond):
            # the caller may drop its inputs as soon as this returns: tell the allocator they are in use on the batch's stream (else the
            # block goes back to the caller's stream and can be handed out again while the sampling kernels still read it)
            if t is not None and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            lat = self.loops[i].sample(x_T, cond, uncond)
        return lat, st

    def drain(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)


class PLMSLoop(DDIMLoop):
    """S-step PLMS (ldm/models/diffusion/plms.py:136-279; eta must be 0) on a frozen QuantModel or any callable
    `unet(x, t, ctx)`: the first step is the pseudo improved Euler pair of evaluations, later steps the
    Adams-Bashforth combinations of the last 1..3 guided predictions (K9b, edadm_plms_step)."""

    def __init__(self, engine, shape, batch, steps=50, scale=7.5, linear_start=0.00085, linear_end=0.012, **kw):
        super().__init__(engine, shape, batch, steps=steps, eta=0.0, scale=scale, linear_start=linear_start,
                         linear_end=linear_end, **kw)

    def _schedule(self):
        return None                      # the second evaluation of the first step visits a timestep out of order

    def _eps(self, img, step, ctx):
        B = img.shape[0]
        ts = torch.full((B * (2 if self.cfg else 1),), int(step), device=img.device, dtype=torch.long)
        e = self.unet(torch.cat([img, img]) if self.cfg else img, ts, ctx).contiguous()
        return (e[B:], e[:B]) if self.cfg else (e, None)

    @torch.no_grad()
    def sample(self, x_T, cond=None, uncond=None, intermediates=None):
        B = x_T.shape[0]
        img = x_T.contiguous()
        ctx = None
        if cond is not None:
            ctx = torch.cat([uncond, cond]) if self.cfg else cond
        time_range = np.flip(self.ddim_timesteps)
        total = time_range.shape[0]
        scale = self.scale if self.cfg else 1.0
        olds = []
        for i, step in enumerate(time_range):
            index = total - i - 1
            coef = self.coef[index:index + 1].expand(B, 5).contiguous()
            ec, eu = self._eps(img, step, ctx)
            if not olds:
                x_tmp, e_t = ops.plms_step(img, ec, eu, scale, [], 0, coef)
                ec2, eu2 = self._eps(x_tmp, time_range[min(i + 1, total - 1)], ctx)
                nxt, _, p0 = ops.plms_step(img, ec2, eu2, scale, [e_t], -1, coef, want_x0=True)
            else:
                nxt, e_t, p0 = ops.plms_step(img, ec, eu, scale, olds, len(olds), coef, want_x0=True)
            olds = [e_t] + olds[:2]
            img = nxt
            if intermediates is not None:
                intermediates.setdefault("x_inter", []).append(img)
                intermediates.setdefault("pred_x0", []).append(p0)
        return img
