"""Diagnostic: eager UNet calls of the frozen LDM-4 engine as a DDIM step issues them (for rocprofv3 --kernel-trace --stats / --pmc).
LAUNCH_LIST=<path>: also writes the engine's own list of int8 GEMM / convolution launches of ONE call -- layer, shape, algorithmic
bytes and the kernel structures the library picked (edadm_diag_launch_kernels) -- which tools/pmc_traffic.py uses as the
per-kernel denominator; each recorded launch is re-issued once for that, i.e. one more call's worth of GEMM launches."""
import sys, os, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch, bench
from edadm import lib
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev)
eng = qnn.freeze()
B = 50
x = torch.randn(2 * B, 3, 64, 64, device=dev); t = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
c = torch.randn(2 * B, 1, 512, device=dev)
want_list = bool(os.environ.get("LAUNCH_LIST"))
eng.prof = [] if want_list else None
eng.ctx_r = eng.context_branches(c)          # the per-step launch set of the sampling loops (context vectors and
eng.emb_r = eng.emb_rows(t)                  # time-embedding rows precomputed; the batch is a guidance pair [x, x])
pre_prof, eng.prof = eng.prof, None          # the hoisted launches (once per batch / run, not per UNet call): listed apart
x = torch.cat([x[:B], x[:B]]).contiguous()
eng.cfg_pair = True
torch.cuda.synchronize()
for _ in range(2):
    eng(x, t, c)
torch.cuda.synchronize()
extra = 0
if os.environ.get("LAUNCH_LIST"):
    TAGS = {1: "k_gemm_nt", 2: "k_gemm_nt8", 3: "k_gemm_p", 4: "k_gemm_ntq", 5: "k_conv3_direct", 6: "k_gemm_split2", 7: "k_gemm_br"}
    eng.prof = []
    eng(x, t, c)
    prof, eng.prof = eng.prof, None
    buf = (ctypes.c_int32 * 8)()
    take = lib.load().edadm_diag_launch_kernels
    def listed(pr):
        rows = []
        for mode, name, M, N, K, flop, run, by in pr:
            take(buf)
            run()
            n = take(buf)
            rows.append({"type": mode, "layer": name, "M": M, "N": N, "K": K, "flop": flop, "kind": by["kind"],
                         "bytes": {k: v for k, v in by.items() if k != "kind"}, "kernels": [TAGS.get(int(buf[i]), "?") for i in range(n)]})
        return rows
    json.dump({"unet_calls_extra": 2, "rows": listed(prof), "hoisted_rows": listed(pre_prof), "hoisted_issued": 2},
              open(os.environ["LAUNCH_LIST"], "w"), indent=0)
    extra = 2                                 # the profiled call + the re-issue of every recorded launch
    torch.cuda.synchronize()
print("MARK")
for _ in range(int(os.environ.get('N_CALLS', '10'))):
    eng(x, t, c)
torch.cuda.synchronize()
print("UNET_CALLS", 2 + extra + int(os.environ.get('N_CALLS', '10')))
