"""Diagnostic: the quantised-output dense layers of LDM-4 (q / k / v projections, GEGLU, ff.net.2 + residual) and the fp32-output ones
on several builds / launch heuristics of the library on ONE box, alternating, with a checksum of every output (same codes or not):
    python tools/dense_ab.py label=lib.so[,ENV=VAL...] [label2=...]      e.g.  base=eda-dm_amd/csrc/libedadm_base.so new=eda-dm_amd/csrc/libedadm.so
Every variant runs in its own process (the library is chosen at import through EDADM_LIB_PATH)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # M, N, K, out_mode (0 fp32), residual
    (102400, 3072, 384, 3, 0), (25600, 4608, 576, 3, 0), (6400, 7680, 960, 3, 0),
    (102400, 384, 1536, 2, 1), (25600, 576, 2304, 2, 1), (6400, 960, 3840, 2, 1),
    (102400, 384, 384, 1, 0), (102400, 384, 384, 2, 0), (25600, 576, 576, 1, 0), (6400, 960, 960, 1, 0),
    (102400, 384, 384, 0, 1), (25600, 576, 576, 0, 1), (6400, 960, 960, 0, 1), (102400, 384, 384, 0, 0),
]
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    variants = []
    for a in sys.argv[1:]:
        label, rest = a.split("=", 1)
        parts = rest.split(",")
        env = dict(os.environ, EDADM_LIB_PATH=os.path.join(ROOT, parts[0]) if not os.path.isabs(parts[0]) else parts[0])
        for kv in parts[1:]:
            k, v = kv.split("=")
            env[k] = v
        variants.append((label, env))
    res = {}
    for rep in range(2):
        for label, env in variants:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            if r.returncode:
                print(label, "failed:", r.stderr[-800:])
                continue
            res.setdefault(label, []).append(json.loads(r.stdout.strip().splitlines()[-1]))
    labels = [l for l, _ in variants if l in res]
    print("%-28s" % "M x N x K mode res" + "".join("%22s" % l for l in labels))
    for i, sh in enumerate(SHAPES):
        row = "%-28s" % ("%dx%dx%d m%d r%d" % sh)
        sums = set()
        for l in labels:
            us = [r[i][0] for r in res[l]]
            sums.add(res[l][0][i][1])
            row += "%22s" % ("%.1f / %.1f us" % (min(us), max(us)))
        print(row + ("   SAME" if len(sums) == 1 else "   DIFFERENT OUTPUT %s" % sorted(sums)))
    tot = {l: sum(min(r[i][0] for r in res[l]) for i in range(len(SHAPES))) for l in labels}
    print("sum of minima (us):", tot)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
flush = torch.empty(80 << 20, dtype=torch.float32, device=dev)


def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for r in range(n):                                     # cold: a 320 MB buffer rewritten in front of every timed launch
        flush.fill_(float(r))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


torch.manual_seed(0)
oqp = torch.tensor([0.05, 128.0, 255.0, 0.0], device=dev)
out = []
for M, N, K, mode, res in SHAPES:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    if mode == 0:
        o = torch.empty(M, N, device=dev)
        fn = lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, o, residual=r)
        us = timeit(fn); fn()
        chk = float(o.double().sum().item())
    else:
        fn = lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=r)
        us = timeit(fn)
        o = fn()
        chk = int(o.view(torch.int8 if mode != 1 else torch.float16).float().double().sum().item())
    out.append((us, chk))
print(json.dumps(out))
