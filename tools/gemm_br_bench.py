#!/usr/bin/env python3
"""Times the weight-resident grouped kernel (k_gemm_br, edadm_qgemm_i8_grouped_q) against the launches it replaces (edadm_qgemm_i8_q:
k_gemm_p / k_gemm_ntq) on the production shapes of one LDM-4 UNet call at 100 rows, HIP events on the launch stream, warm (20 back
to back) and cold (a 320 MB buffer rewritten in front of each launch).  Prints one line per shape; --json writes them to a file."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

SHAPES = [
    ("geglu 32x32", 102400, 384, [(3072, 3)]),
    ("geglu 16x16", 25600, 576, [(4608, 3)]),
    ("qkv 32x32 (i8,i8,f16)", 102400, 384, [(384, 2), (384, 2), (384, 1)]),
    ("qkv 16x16 (f16,f16,f16T)", 25600, 576, [(576, 1), (576, 1), (576, 4)]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--stamps", action="store_true", help="EDADM_LIB_PATH points at the stamps build: print k_gemm_br's in-kernel stamps")
    args = ap.parse_args()
    from edadm import ops
    dev = torch.device("cuda", 0)
    flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(3)
    rows = []

    def timed(fn, cold):
        fn()
        torch.cuda.synchronize()
        tot = 0.0
        n = 5 if cold else 1
        for r in range(n):
            if cold:
                flush.fill_(float(r))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(1 if cold else args.reps):
                fn()
            e1.record()
            e1.synchronize()
            tot += e0.elapsed_time(e1) / (1 if cold else args.reps)
        return 1e3 * tot / n

    for name, M, K, specs in SHAPES:
        probs = []
        for i, (N, mode) in enumerate(specs):
            a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
            w = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8).to(dev)
            # scaled like a real layer: pre-activations of unit size, a step size that leaves the exact-division branch rare
            scale = (1.5e-4 * (0.5 + torch.rand(N, generator=g))).to(dev)
            bias = (0.1 * torch.randn(N, generator=g)).to(dev)
            oqp = ops.qp_tensor([(0.02, 128.0, 255.0)], dev)
            probs.append(dict(A=a, W=w, N=N, scale=scale, bias=bias, out_mode=mode, oqp=oqp, rows_per_batch=256 if mode == 4 else 0))
        sep = lambda: [ops.qgemm_i8_q(p["A"], p["W"], M, p["N"], K, p["scale"], p["bias"], p["out_mode"], p["oqp"],
                                      rows_per_batch=p["rows_per_batch"] or 1) for p in probs]
        grp = lambda: ops.qgemm_i8_grouped_q(probs, M, K)
        same = all(torch.equal(x, y) for x, y in zip(sep(), grp()))
        flop = 2.0 * M * K * sum(n for n, _ in specs)
        r = {"shape": name, "M": M, "K": K, "N": [n for n, _ in specs], "modes": [m for _, m in specs], "bit_identical": same,
             "separate_us_warm": timed(sep, False), "grouped_us_warm": timed(grp, False),
             "separate_us_cold": timed(sep, True), "grouped_us_cold": timed(grp, True)}
        r["grouped_TOPs_warm"] = flop / r["grouped_us_warm"] / 1e6
        r["mfma_roof_us"] = flop / 5033e12 * 1e6
        if args.stamps:
            import ctypes
            from edadm import lib
            L = lib.load()
            buf = (ctypes.c_ulonglong * 8)()
            grp(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
            grp(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
            n0, n1 = max(buf[3], 1), max(buf[7], 1)
            r["stamps"] = {"group0_wave0": {"tiles": int(buf[3]), "waiting": buf[0] / n0, "mfma_phase": buf[1] / n0, "epilogue": buf[2] / n0},
                           "group1_wave4": {"tiles": int(buf[7]), "waiting": buf[4] / n1, "mfma_phase": buf[5] / n1, "epilogue": buf[6] / n1}}
            print("   stamps (cycles per tile; waits are inside the MFMA phase): group 0 wave 0: waiting %.0f, MFMA phase %.0f, epilogue %.0f (%d tiles) | "
                  "group 1 wave 4: waiting %.0f, MFMA phase %.0f, epilogue %.0f (%d tiles)"
                  % (buf[0] / n0, buf[1] / n0, buf[2] / n0, buf[3], buf[4] / n1, buf[5] / n1, buf[6] / n1, buf[7]), flush=True)
        rows.append(r)
        print("%-26s M %6d K %3d  separate %7.1f / %7.1f us   grouped %7.1f / %7.1f us (warm / cold)  %.0f TOP/s  roof %.0f us  same bits: %s"
              % (name, M, K, r["separate_us_warm"], r["separate_us_cold"], r["grouped_us_warm"], r["grouped_us_cold"],
                 r["grouped_TOPs_warm"], r["mfma_roof_us"], same), flush=True)
    ops.device_status()
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
