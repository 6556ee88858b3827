"""Calibration-phase timings (BASELINE.md section 4 quantities ii-iv) on the full-size LDM-4 UNet with
synthetic weights: seconds per block_reconstruction iteration for named units at batch 32,
set_act_quantize_params per calibration batch, save_inp_oup_data per unit."""
import sys, os, time, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
import torch.nn.functional as F


def contract_blas(fwd_func, x, weight, bias, kw):
    if fwd_func is F.conv2d:
        kh, kw_ = weight.shape[2], weight.shape[3]
        if kh == 1 and kw_ == 1 and tuple(kw["stride"]) == (1, 1) and tuple(kw["padding"]) == (0, 0):
            out = torch.einsum("oc,bchw->bohw", weight[:, :, 0, 0], x)
        else:
            B, C, H, W = x.shape
            cols = F.unfold(x, (kh, kw_), padding=kw["padding"], stride=kw["stride"])      # [B, C*kh*kw, L]
            out = torch.matmul(weight.reshape(weight.shape[0], -1), cols)                 # [B, O, L]
            Ho = (H + 2 * kw["padding"][0] - kh) // kw["stride"][0] + 1
            out = out.reshape(B, weight.shape[0], Ho, -1)
        return out if bias is None else out + bias.view(1, -1, 1, 1)
    if fwd_func is F.conv1d:
        out = torch.einsum("oc,bcl->bol", weight[:, :, 0], x)
        return out if bias is None else out + bias.view(1, -1, 1)
    return fwd_func(x, weight, bias, **kw)



if os.environ.get("EDADM_CONTRACT", "") == "blas":
    # comparison leg: rocBLAS / torch contraction patched over the product's own kernels (never part of the product)
    import qdiff.quant_layer as _ql
    _ql._contract = contract_blas


torch.backends.cudnn.benchmark = os.environ.get("CUDNN_BENCH", "0") == "1"
dev = torch.device("cuda", 0)
t0 = time.time()
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=int(os.environ.get("CALIB_ROWS", "16")))
print("build+init", time.time() - t0, calib)
from qdiff.block_recon import block_reconstruction
from qdiff.layer_recon import layer_reconstruction
from qdiff.data_utils import save_inp_oup_data
g = torch.Generator().manual_seed(3)
N = 64
x = torch.randn(N, 3, 64, 64, generator=g).to(dev)
t = torch.randint(1, 1000, (N,), generator=g).to(dev)
c = torch.randn(N, 1, 512, generator=g).to(dev)
cali = (x, t, c)
qnn.set_quant_state(True, True)
res = {}
for name, unit, fn in (("input_blocks.1.0 (ResBlock 192->192 @64x64)", qnn.model.input_blocks[1][0], block_reconstruction),
                       ("input_blocks.4.1.transformer_blocks.0 (384 @32x32)", qnn.model.input_blocks[4][1].transformer_blocks[0], block_reconstruction),
                       ("middle_block.0 (ResBlock 960 @8x8)", qnn.model.middle_block[0], block_reconstruction)):
    torch.cuda.synchronize(); t1 = time.time()
    r = save_inp_oup_data(qnn, unit, cali, True, True, batch_size=32, input_prob=True, keep_gpu=True)
    torch.cuda.synchronize(); t_cache = time.time() - t1
    iters = 10
    kw = dict(cali_data=cali, iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0,
              weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=0.8, recon_w=True,
              recon_a=True, keep_gpu=True)
    torch.cuda.synchronize(); t1 = time.time()
    fn(qnn, unit, **kw)
    torch.cuda.synchronize(); t_all = time.time() - t1
    kw["iters"] = 3 * iters
    # second run reuses the AdaRound state? units are rebuilt each call: time difference isolates the loop
    res[name] = dict(cache_s_for_64_samples=t_cache, recon_call_s_10_iters=t_all)
    print(name, res[name])
print(json.dumps(res))
