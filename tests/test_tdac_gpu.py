"""-m gpu: the TDAC step scores (scripts/calibration.py:47-69 of the reference) as ONE HIP launch (edadm_tdac_pair_scores) against the
reference's own torch statements on the same device tensors: the density counts, the variety sums and the resulting allocation."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
pytestmark = pytest.mark.gpu


def _reference_statements(fm, r):
    T = len(fm)
    dense_num, cos_dis = torch.zeros(T, dtype=torch.int16), torch.zeros(T)
    mse = torch.zeros(T, T)
    for i in range(T):
        for j in range(T):
            if i != j:
                mse[i, j] = torch.mean((fm[i] - fm[j]) ** 2).cpu()
                if mse[i, j] <= r:
                    dense_num[i] = dense_num[i] + 1
                cos_dis[i] = cos_dis[i] + torch.sum(1 - F.cosine_similarity(fm[i], fm[j], dim=1, eps=1e-6)).cpu()
    return dense_num, cos_dis, mse


@pytest.mark.parametrize("shape,T", [((8, 32, 4, 4), 10), ((64, 960, 8, 8), 20), ((4, 24, 7), 6), ((5, 16, 3, 5), 3)])
def test_tdac_pair_scores_match_the_reference_statements(shape, T):
    from edadm import ops, lib
    from edadm.tdac import tdac_scores, tdac_allocate
    g = torch.Generator().manual_seed(sum(shape) + T)
    base = torch.randn(shape, generator=g)
    # a drifting sequence like a sampling trajectory: neighbours close, ends far apart, mean squared differences around the radius
    fm = [(base * (1.0 + 0.08 * t) + 0.35 * t * torch.randn(shape, generator=g)).cuda() for t in range(T)]
    if len(shape) == 4:
        fm[1] = fm[1].contiguous(memory_format=torch.channels_last)        # what a channels_last producer hands over
    r = 3.0 if shape[1] > 100 else float(np.median([float(torch.mean((fm[i] - fm[i + 2]) ** 2)) for i in range(T - 2)]))
    dn_ref, cd_ref, mse_ref = _reference_statements(fm, r)
    lib.CALLS = {}
    try:
        dn, cd = tdac_scores(fm, r)
        calls = dict(lib.CALLS)
    finally:
        lib.CALLS = None
    assert calls.get("edadm_tdac_pair_scores") == 1
    mse, cdm = ops.tdac_pair_scores(fm)
    off = ~torch.eye(T, dtype=torch.bool)
    assert torch.allclose(mse.cpu()[off], mse_ref[off], rtol=2e-6, atol=0)
    assert torch.equal(mse.cpu(), mse.cpu().T) and float(mse.cpu().diagonal().abs().max()) == 0.0
    # counts: equal unless a pair sits within rounding of the radius (excluded from the comparison if so)
    near = ((mse_ref - r).abs() <= 4e-6 * r) & off
    if not near.any():
        assert torch.equal(dn, dn_ref), (dn, dn_ref)
    assert torch.allclose(cd, cd_ref, rtol=2e-5, atol=1e-4 * float(cd_ref.abs().max()))
    if T >= 6 and not near.any():
        a = tdac_allocate(fm, 1.2, 256, r)
        # the same allocation as from the reference's statements
        dnn = (dn_ref - dn_ref.min()) / (dn_ref.max() - dn_ref.min())
        cnn = (cd_ref - cd_ref.min()) / (cd_ref.max() - cd_ref.min())
        w = dnn + 1.2 * cnn
        t_num = (w / w.sum() * 256).round().to(torch.int64)
        assert int(a[3].sum()) == 256
        assert int((a[3] - t_num).abs().max()) <= 1            # before the +-1 fix-up of the rounding error (calibration.py:72-90)


def test_tdac_pair_scores_beyond_256_steps():
    """LSUN-Church samples 500 DDIM steps by default (scripts/sample_diffusion_ldm_church.py:35): one feature map per step, T = 500.
    The launch has no limit on T (the diagonal is written by a strided loop); reference statements evaluated on host copies."""
    from edadm import ops
    from edadm.tdac import tdac_scores, tdac_allocate
    T, shape = 500, (2, 8, 2, 2)
    g = torch.Generator().manual_seed(500)
    base = torch.randn(shape, generator=g)
    fm_cpu = [base * (1.0 + 0.002 * t) + 0.01 * t * torch.randn(shape, generator=g) for t in range(T)]
    fm = [f.cuda() for f in fm_cpu]
    mse, cdm = ops.tdac_pair_scores(fm)
    mse, cdm = mse.cpu(), cdm.cpu()
    assert mse.shape == (T, T) and float(mse.diagonal().abs().max()) == 0.0 and float(cdm.diagonal().abs().max()) == 0.0
    assert torch.equal(mse, mse.T) and torch.equal(cdm, cdm.T)
    st = torch.stack(fm_cpu).double()
    rows = [0, 1, 255, 256, 257, 499]
    for i in rows:
        d = ((st[i][None] - st) ** 2).flatten(1).mean(1).float()
        d[i] = 0.0
        assert torch.allclose(mse[i], d, rtol=2e-6, atol=0), i
        cs = torch.stack([torch.sum(1 - F.cosine_similarity(fm_cpu[i], fm_cpu[j], dim=1, eps=1e-6)) for j in range(T)])
        cs[i] = 0.0
        assert torch.allclose(cdm[i], cs, rtol=2e-5, atol=2e-5), i
    r = float(mse[0, 40])
    dn, cd = tdac_scores(fm, r * 1.0000001)
    near = mse <= r * 1.0000001
    near.fill_diagonal_(False)
    assert torch.equal(dn.long(), near.sum(1))
    # the sequential fp32 sum over j != i in the reference's order
    for i in rows:
        s = torch.zeros(())
        for j in range(T):
            if j != i:
                s = s + cdm[i, j]
        assert float(s) == float(cd[i]), i
    a = tdac_allocate(fm, 1.2, 1024, r)
    assert int(a[3].sum()) == 1024 and a[3].numel() == T
