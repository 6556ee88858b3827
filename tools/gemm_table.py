"""Diagnostic (not part of the product): per-layer-shape table of the int8 GEMM launches of one UNet call as a DDIM step
issues it (guidance pair, context vectors and time-embedding rows precomputed): M, N, K, launches, microseconds, TOP/s,
algorithmic bytes and the HBM-roof time they imply.  python tools/gemm_table.py [out.json]"""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch, bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev)
eng = qnn.freeze()
B = 50
x = torch.randn(B, 3, 64, 64, device=dev); x = torch.cat([x, x]).contiguous()
t = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
c = torch.randn(2 * B, 1, 512, device=dev)
eng.ctx_r = eng.context_branches(c)
eng.emb_r = eng.emb_rows(t)
eng.cfg_pair = True
for _ in range(2):
    eng(x, t, c)
torch.cuda.synchronize()
eng.prof = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng(x, t, c); e1.record(); torch.cuda.synchronize()
print("eager unet ms", e0.elapsed_time(e1))
rows = {}
def kms(run, reps=5):
    run(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
prof, eng.prof = eng.prof, None
for mode, name, M, N, K, f, run, by in prof:
    key = (mode, by["kind"], M, N, K, by["out"] / (M * N), by["res"] > 0)
    r = rows.setdefault(key, dict(n=0, ms=0.0, flop=0.0, bytes=0.0, names=[]))
    r["n"] += 1; r["ms"] += kms(run); r["flop"] += f; r["bytes"] += sum(v for k, v in by.items() if k != "kind")
    r["names"].append(name)
tot = sum(r["ms"] for r in rows.values())
print("gemm total ms %.3f  PFLOP %.3f  algorithmic GB %.2f" % (tot, sum(r["flop"] for r in rows.values()) / 1e15, sum(r["bytes"] for r in rows.values()) / 1e9))
out = []
print("%-4s %-6s %7s %5s %5s %4s %3s %3s %8s %8s %9s %9s %6s" % ("type", "kind", "M", "N", "K", "oB", "res", "n", "us/call", "TOP/s", "alg MB", "hbm us", "bound"))
for key, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
    us = 1e3 * r["ms"] / r["n"]
    tops = r["flop"] / r["ms"] / 1e9
    mb = r["bytes"] / r["n"] / 1e6
    hbm_us = r["bytes"] / r["n"] / 6.3e12 * 1e6            # at the 6.3 TB/s a streaming copy reaches
    mfma_us = r["flop"] / r["n"] / 5.033e15 * 1e6
    print("%-4s %-6s %7d %5d %5d %4.1f %3d %3d %8.1f %8.1f %9.2f %9.1f %6s" % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], r["n"], us, tops, mb, hbm_us, "hbm" if hbm_us > mfma_us else "mfma"))
    out.append(dict(type=key[0], kind=key[1], M=key[2], N=key[3], K=key[4], out_bytes_per_elem=key[5], residual=bool(key[6]), launches=r["n"],
                    us_per_call=us, tops=tops, algorithmic_MB=mb, hbm_roof_us=hbm_us, mfma_roof_us=mfma_us, layers=r["names"][:3]))
if len(sys.argv) > 1:
    json.dump({"total_ms": tot, "rows": out}, open(sys.argv[1], "w"), indent=1)
