"""-m gpu: the product's UniformAffineQuantizer (host class + HIP K1/K3) against the reference's own
initialisation results: per-channel weight scales (G1) and per-tensor activation scales with the
EMA over batches (G2) — bit-exact delta / zero_point, bit-exact fake-quant outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = lambda a: torch.as_tensor(np.asarray(a))


def test_weight_scale_init_bit_exact(golden):
    from qdiff.quant_layer import UniformAffineQuantizer
    g = golden("g1_weight_init")
    keys = sorted({k.rsplit("/", 1)[0] for k in g.files if k.endswith("/delta")})
    assert len(keys) >= 12
    for key in keys:
        cname, b, s = key.split("/")
        q = UniformAffineQuantizer(n_bits=int(b[1:]), symmetric=(s == "sym"), channel_wise=True, scale_method="mse")
        out = q(T(g["w/" + cname]).cuda())
        assert {"pos": 1, "neg": -1, "no": 0}[q.one_side_dist] == int(g[key + "/one_side"])
        np.testing.assert_array_equal(q.delta.cpu().numpy(), g[key + "/delta"], err_msg=key)
        np.testing.assert_array_equal(q.zero_point.cpu().numpy(), g[key + "/zero_point"], err_msg=key)
        np.testing.assert_array_equal(out.cpu().numpy(), g[key + "/out"], err_msg=key)


def test_act_scale_init_ema_bit_exact(golden):
    from qdiff.quant_layer import UniformAffineQuantizer
    g = golden("g2_act_init")
    for run in sorted({k.split("/step")[0] for k in g.files}):
        cname, b, s = run.split("/")
        q = UniformAffineQuantizer(n_bits=int(b[1:]), symmetric=(s == "sym"), channel_wise=False, scale_method="mse",
                                   leaf_param=True, prob=0.5)
        for k in range(4):
            key = "%s/step%d" % (run, k)
            out = q(T(g[key + "/x"]).cuda())
            np.testing.assert_array_equal(q.delta.detach().cpu().numpy(), g[key + "/delta"], err_msg=key)
            np.testing.assert_array_equal(q.zero_point.cpu().numpy(), g[key + "/zero_point"], err_msg=key)
            np.testing.assert_array_equal(q.running_min.cpu().numpy().reshape(()), g[key + "/running_min"], err_msg=key)
            np.testing.assert_array_equal(q.running_max.cpu().numpy().reshape(()), g[key + "/running_max"], err_msg=key)
            np.testing.assert_array_equal(out.detach().cpu().numpy(), g[key + "/out"], err_msg=key)


def test_no_cpu_fallback():
    from qdiff.quant_layer import UniformAffineQuantizer
    from edadm.lib import EdadmError
    q = UniformAffineQuantizer(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True)
    with pytest.raises(EdadmError):
        q(torch.randn(4, 4))
