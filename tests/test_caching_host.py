"""CPU: the host logic of the activation caches of `save_inp_oup_data` (qdiff/data_utils.py) -- look-ahead FP sweeps and
memoised reconstructed units -- on a toy model whose units are BaseQuantBlock subclasses with plain torch forwards:
same tensors as the reference's per-unit double pass (data_utils.py:112-150), in one process and sharded over two
gloo ranks (batches split across ranks, slabs all-gathered in batch order)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))


def _build():
    from qdiff.quant_block import BaseQuantBlock

    class Unit(BaseQuantBlock):
        def __init__(self, d, two_inputs=False):
            super().__init__({})
            self.lin = nn.Linear(d, d)
            self.gain = nn.Parameter(torch.zeros(()))
            self.two = two_inputs

        def forward(self, x, emb=None):
            y = self.lin(x)
            if emb is not None:
                y = y + emb
            if self.use_weight_quant:                      # the "quantised" behaviour, changed by `gain`
                y = torch.round(y * 8) / 8 * (1 + self.gain)
            return y

    class Toy(nn.Module):
        def __init__(self, d=8):
            super().__init__()
            self.emb = nn.Linear(1, d)
            self.u0, self.u1, self.u2, self.u3 = Unit(d), Unit(d, True), Unit(d), Unit(d)
            self.skip = Unit(d)
            self.skip.ignore_reconstruction = True
            self.block_count = 0

        def set_quant_state(self, weight_quant=False, act_quant=False):
            for m in self.modules():
                if isinstance(m, Unit):
                    m.set_quant_state(weight_quant, act_quant)

        def forward(self, x, t):
            e = self.emb(t[:, None].float())
            h0 = self.u0(x)
            h1 = self.u1(torch.relu(h0), e)
            h2 = self.u2(h1 + self.skip(h0))
            return self.u3(torch.cat([h2, h0], 0)[: x.shape[0]] + h2)

    torch.manual_seed(0)
    return Toy()


def _flat(r):
    out = []

    def walk(v):
        if torch.is_tensor(v):
            out.append(v)
        elif isinstance(v, (list, tuple)):
            for e in v:
                walk(e)
    walk(r[1:])
    return r[0], out


def _walk(model, cali, trace_gb, memo_gb):
    import qdiff.data_utils as du
    old = du.FP_TRACE_GB, du.Q_MEMO_GB
    du.FP_TRACE_GB, du.Q_MEMO_GB = float(trace_gb), float(memo_gb)
    try:
        du.clear_fp_trace(model)
        du.STATS.update(fp_passes=0, fp_captures=0, units_served=0, memo_hits=0)
        units = du.recon_units(model)
        res = []
        with torch.no_grad():
            for u in units:
                u.gain.zero_()
        for u in units:
            res.append(_flat(du.save_inp_oup_data(model, u, cali, True, True, batch_size=32, input_prob=True)))
            with torch.no_grad():
                u.gain.fill_(0.25)                      # "reconstruction": the unit's quantised behaviour changes
        stats = dict(du.STATS)
        du.clear_fp_trace(model)
    finally:
        du.FP_TRACE_GB, du.Q_MEMO_GB = old
    return units, res, stats


def _check(model, cali):
    units, ref, st0 = _walk(model, cali, "0", "0")
    assert len(units) == 4 and st0["fp_captures"] == 0             # the ignored unit is not a reconstruction unit
    assert ref[1][0] is True and ref[0][0] is False                # the two-input unit reports Resblock
    for trace_gb, memo_gb, sweeps in (("1", "0", 1), ("1", "1", 1), ("0.00001", "1", None), ("1", "0.000005", 1)):
        _, got, st = _walk(model, cali, trace_gb, memo_gb)
        for (rb, a), (rb2, b) in zip(ref, got):
            assert rb == rb2 and len(a) == len(b)
            for x, y in zip(a, b):
                assert x.shape == y.shape and torch.equal(x, y)
        assert st["units_served"] == 4
        assert st["fp_captures"] == sweeps if sweeps else 1 < st["fp_captures"] <= 4
        assert (st["memo_hits"] > 0) == (memo_gb != "0")
    return True


def test_caches_match_per_unit_passes_single_process():
    model = _build()
    g = torch.Generator().manual_seed(1)
    cali = (torch.randn(128, 8, generator=g), torch.randint(0, 1000, (128,), generator=g))
    assert _check(model, cali)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _build()
    g = torch.Generator().manual_seed(1)
    cali = (torch.randn(128, 8, generator=g), torch.randint(0, 1000, (128,), generator=g))
    ok = _check(model, cali)                                           # 4 batches: 2 per rank, gathered in order
    _, res, _ = _walk(model, cali, "1", "1")
    ret[rank] = (ok, [t.double().sum().item() for _, ts in res for t in ts])
    dist.destroy_process_group()


def test_caches_match_per_unit_passes_world2_gloo():
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == ret[1][1]                                      # both ranks hold the same gathered slabs
    # and they equal the single-process result
    model = _build()
    g = torch.Generator().manual_seed(1)
    cali = (torch.randn(128, 8, generator=g), torch.randint(0, 1000, (128,), generator=g))
    _, res, _ = _walk(model, cali, "0", "0")
    assert ret[0][1] == [t.double().sum().item() for _, ts in res for t in ts]
