"""DDIM stepping for the CIFAR DDPM UNet — API of the reference's ddim/functions/denoising.py
(`compute_alpha` :4-7, `generalized_steps` :37-59, `cali_generalized_steps` :10-35).  The
per-step update runs as one HIP kernel (edadm_ddim_step, K9) and the trajectory stays on the device
(the reference copies every x_t to the host, a sync per step)."""
import torch

from edadm import ops


def compute_alpha(beta, t):
    beta = torch.cat([torch.zeros(1).to(beta.device), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def _coef(at, at_next, eta):
    """Per-sample rows of edadm_ddim_step: {sqrt(1-a_t), sqrt(a_t), sqrt(a_next), c2, c1}."""
    c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
    c2 = ((1 - at_next) - c1 ** 2).sqrt()
    return torch.stack([(1 - at).sqrt(), at.sqrt(), at_next.sqrt(), c2, c1], dim=-1).reshape(-1, 5).contiguous().float()


def _steps(x, seq, model, b, yield_each, **kwargs):
    with torch.no_grad():
        n = x.size(0)
        seq_next = [-1] + list(seq[:-1])
        x0_preds, xs = [], [x]
        if yield_each:
            yield xs
        eta = kwargs.get("eta", 0)
        for i, j in zip(reversed(seq), reversed(seq_next)):
            t = (torch.ones(n) * i).to(x.device)
            next_t = (torch.ones(n) * j).to(x.device)
            at, at_next = compute_alpha(b, t.long()), compute_alpha(b, next_t.long())
            xt = xs[-1]
            et = model(xt, t)
            noise = torch.randn_like(x) if eta else None
            xt_next, x0_t = ops.ddim_step(xt.contiguous(), et.contiguous(), None, 1.0, _coef(at, at_next, eta),
                                          noise=noise, want_x0=True)
            x0_preds.append(x0_t)
            xs.append(xt_next)
            if yield_each:
                yield xs
        if not yield_each:
            yield xs, x0_preds


def generalized_steps(x, seq, model, b, **kwargs):
    return next(_steps(x, seq, model, b, False, **kwargs))


def cali_generalized_steps(x, seq, model, b, **kwargs):
    """Generator yielding the list of x_t after every step (TDAC trajectory capture)."""
    return _steps(x, seq, model, b, True, **kwargs)
