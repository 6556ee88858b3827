"""Diagnostic (GPU box): gradients of ONE reconstruction iteration of a toy-model unit, product (HIP) vs oracle (CPU), both
forced to the reference's parameters after iteration k-1 of fixture G8c / G8b (teacher forcing: separates a per-iteration
error from the chaotic amplification of Adam's normalised steps).  python tools/recon_grad_check.py g8c_recon_caches rb 0 1 2"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import _uniforms  # noqa: E402
from helpers import build_toynet, WQ4, AQ8  # noqa: E402
from oracle import qdiff_oracle as O  # noqa: E402
from test_oracle_nets import ToyNet, sub_sd, T  # noqa: E402
from test_oracle_round3 import golden_caches, load_init_scales  # noqa: E402


def main(fixture, unit_name, ks):
    from qdiff import QuantModel
    from qdiff.quant_layer import UniformAffineQuantizer
    from qdiff.quant_block import QuantAttnBlock
    from edadm.state import load_quant_state
    import edadm.recon as recon
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    prob, input_prob = float(g["prob"]), float(g["input_prob"])
    kind = dict(conv_in="layer", temb_lin="layer", rb="block", at="block", conv_out="layer")[unit_name]
    cu = lambda a: torch.as_tensor(np.asarray(a)).cuda()
    results = []
    for k in ks:
        forced_w = g["traj/%s/w" % unit_name][k - 1] if k > 0 else None
        forced_a = g["traj/%s/a" % unit_name][k - 1] if k > 0 else None
        idx = [int(v) for v in g["idx/" + unit_name][k]]
        # ---------------- product
        aq = dict(AQ8)
        aq["prob"] = prob
        qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
        load_quant_state(qnn, {kk: g[kk] for kk in g.files if kk.startswith("init/qp/")}, prefix="init/qp/")
        mods = dict(qnn.named_modules())

        # draw c of iteration k of an owner: 1 draw per iteration for layers / the input mix, 2 for a block's quantizers
        counters = {}
        doubled = [False]                  # True while the PRODUCT runs a batched unit

        def draw(owner, shape, per_iter):
            shape = tuple(int(v) for v in shape)
            if per_iter == 2 and doubled[0]:              # the product's batched [x | x] forward = the reference's two calls
                doubled[0] = False
                try:
                    half = (shape[0] // 2,) + shape[1:]
                    return np.concatenate([draw(owner, half, 2), draw(owner, half, 2)])
                finally:
                    doubled[0] = True
            c = counters.get(owner, 0)
            counters[owner] = c + 1
            return _uniforms.uniform(owner, "iter", k * per_iter + c, shape)

        per_iter_q = 2 if kind == "block" else 1
        for name, m in qnn.named_modules():
            if isinstance(m, UniformAffineQuantizer) and m.leaf_param:
                tr = name.endswith(".act_quantizer_w") and isinstance(mods[name.rsplit(".", 1)[0]], QuantAttnBlock)
                m.injected_uniform = (lambda nm, tr: lambda xx: torch.from_numpy(
                    draw(nm, xx.shape, per_iter_q).transpose(0, 2, 1).copy() if tr else draw(nm, xx.shape, per_iter_q)).to(xx.device))(name, tr)
        recon.INJECT_MIX_UNIFORM = lambda xx: torch.from_numpy(draw("input_mix:" + unit_name, xx.shape, 1)).to(xx.device)

        def save_fn(*a, **kw):
            kk = "cache/%s/" % unit_name
            if bool(g[kk + "resblock"]):
                return True, ([cu(g[kk + "inp_q"]), cu(g[kk + "temb_q"])], [cu(g[kk + "inp_fp"]), cu(g[kk + "temb_fp"])]), cu(g[kk + "out_fp"])
            return False, (cu(g[kk + "inp_q"]), cu(g[kk + "inp_fp"])), cu(g[kk + "out_fp"])

        grads = {}
        oinit, olaunch = recon.FusedAdam.__init__, recon.FusedAdam.launch

        def init(self, params, lr, t_max, betas=(0.9, 0.999)):
            oinit(self, params, 0.0, t_max, betas)
            key = "a" if self.params[0].numel() == 1 else "w"
            f = forced_a if key == "a" else forced_w
            if f is not None:
                self.flat.copy_(torch.as_tensor(f).cuda())

        def launch(self):
            olaunch(self)
            grads["a" if self.params[0].numel() == 1 else "w"] = self.grad.detach().cpu().numpy().copy()

        recon.FusedAdam.__init__, recon.FusedAdam.launch = init, launch
        osample = random.sample
        random.sample = lambda pop, n: idx
        try:
            doubled[0] = kind == "block" and recon.BATCH_FORWARDS
            recon.reconstruct(qnn, getattr(qnn.model, unit_name), (cu(g["x"]), cu(g["t"])), is_block=(kind == "block"),
                              save_fn=save_fn, iters=1, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-3, lr_w=5e-2, p=2.0,
                              weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=input_prob, add_loss=0.8,
                              recon_w=True, recon_a=True)
        finally:
            doubled[0] = False
            recon.FusedAdam.__init__, recon.FusedAdam.launch = oinit, olaunch
            recon.INJECT_MIX_UNIFORM = None
        # ---------------- oracle
        net = ToyNet(sub_sd(g, "sd/"), WQ4, aq)
        x, t = T(g["x"]), T(g["t"])
        with torch.no_grad():
            net(x, t)
        load_init_scales(net, g)
        counters.clear()
        for q in net.all_quantizers():
            if isinstance(q, O.OQ):
                q.mask_fn = (lambda nm: lambda xx: torch.from_numpy(draw("model." + nm, xx.shape, per_iter_q)))(q.name)
        ograds = {}
        oa_init = O.OAdam.__init__

        def oinit2(self, params, lr, t_max, betas=(0.9, 0.999), eps=1e-8):
            oa_init(self, params, 0.0, t_max, betas, eps)
            f = forced_a if self.params[0].numel() == 1 else forced_w
            if f is not None:
                o = 0
                with torch.no_grad():
                    for p in self.params:
                        p.copy_(torch.as_tensor(f[o:o + p.numel()]).reshape(p.shape))
                        o += p.numel()

        O.OAdam.__init__ = oinit2
        try:
            O.reconstruct_unit(net, getattr(net, unit_name), kind, cali=(x, t), iters=1, act_quant=True, lr_a=1e-3, lr_w=5e-2, p=2.0,
                               batch_size=16, input_prob=input_prob, add_loss=0.8, recon_w=True, recon_a=True, cache_batch=32,
                               caches=golden_caches(g, unit_name),
                               rand_fn=lambda xx: torch.from_numpy(draw("input_mix:" + unit_name, xx.shape, 1)),
                               trace=lambda it, wp, ap, l: ograds.update(w=torch.cat([p.grad.flatten() for p in wp]).numpy().copy(),
                                                                         a=torch.cat([p.grad.flatten() for p in ap]).numpy().copy()))
        finally:
            O.OAdam.__init__ = oa_init
            random.sample = osample
        for key in ("w", "a"):
            a, b = grads[key], ograds[key]
            results.append((k, key, float(np.abs(a - b).max() / np.abs(b).max()),
                            float(np.median(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))))
            scale = np.abs(b).max()
            d = np.abs(a - b)
            rel = d / np.maximum(np.abs(b), 1e-30)
            print("k=%d %s: |g| max %.3e median %.3e | abs err max %.3e (%.2e of max) | rel err median %.2e, frac(rel>1e-2) %.5f, "
                  "frac(rel>1e-2 and |g|>1e-3 max) %.5f, sign flips %d, zeros prod %d oracle %d"
                  % (k, key, scale, np.median(np.abs(b)), d.max(), d.max() / scale, np.median(rel), (rel > 1e-2).mean(),
                     ((rel > 1e-2) & (np.abs(b) > 1e-3 * scale)).mean(), int((np.sign(a) != np.sign(b)).sum()),
                     int((a == 0).sum()), int((b == 0).sum())))
            if key == "w":
                worst = np.argsort(-d)[:8]
                print("   worst:", [(int(i), float(a[i]), float(b[i])) for i in worst])
            else:
                print("   a grads prod", a, "oracle", b)
    return results


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3:]])
