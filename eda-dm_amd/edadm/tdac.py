"""TDAC — Temporal Distribution Alignment Calibration (scripts/calibration.py of the reference): score
every sampling step by the density (`#{j : mean((F_i-F_j)^2) <= r}`, :45-52) and the variety
(`sum_j sum(1 - cos(F_i, F_j))`, :58-63) of the mid-block features, turn `w = D^ + lambda V^` into an
integer number of calibration samples per step summing to N (:66-92), and assemble the calibration set.

SURVEY.md §8f-1: on the device the O(T^2) scoring is ONE launch (edadm_tdac_pair_scores, csrc/small.hip: a workgroup per pair reads
the two maps once -- the reference's loop is 380 pairs x five torch passes); host tensors (the host-logic test of the allocation
rule, G9) take the reference's own torch statements."""
import numpy as np
import torch
import torch.nn.functional as F


def tdac_scores(feature_map, dense_r):
    T = len(feature_map)
    dense_num = torch.zeros(T, dtype=torch.int16)
    cos_dis = torch.zeros(T)
    if feature_map[0].is_cuda and T >= 2:
        from . import ops
        mse, cd = ops.tdac_pair_scores(feature_map)
        mse, cd = mse.cpu(), cd.cpu().numpy()
        # the reference's order of j: a count, and a sequential fp32 sum over j != i.  Both [T][T] tables come back with a zero
        # diagonal: `x + 0.0f == x`, so the running fp32 sum along a whole row (numpy's cumsum accumulates left to right in the
        # array's type) is the reference's sum over j != i, bit for bit; the count leaves the diagonal out explicitly
        near = mse <= dense_r
        near.fill_diagonal_(False)
        dense_num = near.sum(dim=1).to(torch.int16)
        cos_dis = torch.from_numpy(np.cumsum(cd, axis=1, dtype=np.float32)[:, -1].copy())
        return dense_num, cos_dis
    for i in range(T):
        for j in range(T):
            if i != j:
                if torch.mean((feature_map[i] - feature_map[j]) ** 2) <= dense_r:
                    dense_num[i] = dense_num[i] + 1
                cos_dis[i] = cos_dis[i] + torch.sum(1 - F.cosine_similarity(feature_map[i], feature_map[j], dim=1,
                                                                             eps=1e-6)).cpu()
    return dense_num, cos_dis


def tdac_allocate(feature_map, lamda, calib_num_samples, dense_r, fixup_ge=False):
    """-> (dense_num, cos_dis, w, t_num).  `fixup_ge`: the Church generator decrements entries that are
    already 0 (`>= 0`, calibration.py:332) where the others require `> 0` (:84,225,460,593)."""
    dense_num, cos_dis = tdac_scores(feature_map, dense_r)
    dn = (dense_num - dense_num.min()) / (dense_num.max() - dense_num.min())
    cn = (cos_dis - cos_dis.min()) / (cos_dis.max() - cos_dis.min())
    w = dn + lamda * cn
    prob = w / torch.sum(w)
    t_num = (prob * calib_num_samples).round().to(torch.int64)
    t_error = int(calib_num_samples - torch.sum(t_num))
    _, order = torch.sort(t_num, descending=True)
    if t_error >= 0:
        t_num[order[:t_error]] += 1
    else:
        for i in reversed(range(len(t_num))):
            if t_error == 0:
                break
            if (t_num[i] >= 0) if fixup_ge else (t_num[i] > 0):
                t_num[i] -= 1
                t_error += 1
    assert int(torch.sum(t_num)) == calib_num_samples
    return dense_num, cos_dis, w, t_num


def shuffled_step_list(t_num, device):
    t = torch.hstack([torch.full((int(num),), step) for step, num in enumerate(t_num)])
    return t[torch.randperm(t.size(0))].to(device)


def pick_by_step(all_samples, t):
    """all_samples[s] = x at sampling step s for every trajectory; row k takes its x from step t[k]."""
    calib = torch.zeros_like(all_samples[0])
    for s, x_s in enumerate(all_samples):
        mask = t == s
        if mask.any():
            calib += x_s * mask.float().view(-1, *([1] * (x_s.dim() - 1)))
    return calib
