"""HBM evidence table for the HBM-bound kernel groups of the hot path (north_star: "evidenced by rocprof HBM GB/s"):
    python tools/elementwise_hbm.py <label> <stats_dir> <fetch_dir> <write_dir> <bytes.json> <out.txt>
stats_dir: rocprofv3 --kernel-trace --stats of a workload; fetch_dir / write_dir: --pmc FETCH_SIZE / WRITE_SIZE passes of the SAME
command (separate passes; corrections of /opt/skills/guides/MI355X_MICROARCH.md: counters in KB, FETCH_SIZE x2 on gfx950 for wide
streaming reads, WRITE_SIZE exact); bytes.json: the algorithmic bytes per entry point accounted at the C ABI over the life of that
process (edadm/trace_bytes.py).  One row per kernel group: launches, algorithmic bytes, counter bytes, device time, GB/s on both byte
counts, fraction of the 8 TB/s HBM peak.  Whole-process totals on all four inputs, so the rows compare like with like."""
import csv
import glob
import json
import os
import re
import sys

PEAK = 8000.0
# (K#, what, entry points whose algorithmic bytes belong to the group, kernel-name patterns of the group)
GROUPS = [
    ("K1", "fake-quant forward (quant_layer.py:266-276)", ["edadm_fake_quant_fwd"], [r"\bk_fq_fwd\b"]),
    ("K1", "fake-quant backward + d delta partial sums", ["edadm_fake_quant_bwd"], [r"\bk_fq_bwd\b"]),
    ("K2", "AdaRound soft/hard weights forward (adaptive_rounding.py:49-61)", ["edadm_adaround_fwd"], [r"\bk_ar_fwd\b"]),
    ("K2", "AdaRound d alpha", ["edadm_adaround_bwd"], [r"\bk_ar_bwd\b"]),
    ("K5", "GroupNorm statistics (sampling): reduces the producers' epilogue partials, no pass over x", ["edadm_groupnorm_stats", "edadm_groupnorm_stats_cat", "edadm_groupnorm_stats_cat_rep"],
     [r"\bk_gn_partial\b", r"\bk_gn_final\b"]),
    ("K5", "GroupNorm apply + SiLU + int8 operands (sampling)", ["edadm_groupnorm_apply", "edadm_groupnorm_apply_cat", "edadm_groupnorm_apply_cat_raw"],
     [r"\bk_gn_apply16\b", r"\bk_gn_apply\b"]),
    ("K5", "LayerNorm + int8 operands (sampling)", ["edadm_layernorm_quant", "edadm_layernorm_quant_radd"], [r"\bk_ln_quant_v4\b", r"\bk_ln_quant\b"]),
    ("K5", "stand-alone activation quantiser", ["edadm_quant_i8", "edadm_quant_i8_cat", "edadm_quant_i8_cat_rep"], [r"\bk_quant_i8\b"]),
    ("K7", "lp loss forward", ["edadm_lp_loss_fwd"], [r"\bk_lp_fwd\b"]),
    ("K7", "lp loss backward / per-module gradient injection", ["edadm_lp_loss_bwd", "edadm_lp_loss_inject"], [r"\bk_lp_bwd\b", r"\bk_lp_inject\b"]),
    ("K8", "Adam + cosine step (block_recon.py:199-206)", ["edadm_adam_step"], [r"\bk_adam\b"]),
    ("K9", "DDIM step with CFG combine", ["edadm_ddim_step"], [r"\bk_ddim\b"]),
    ("K10", "input mix where(u < p, q, fp) (block_recon.py:141-145)", ["edadm_mix_where"], [r"\bk_mix\b"]),
    ("K12", "GroupNorm forward (reconstruction graph, NHWC)", ["edadm_gn_fwd_nhwc"], [r"\bk_gnt_partial<0>", r"\bk_gnt_final<0>", r"\bk_gnt_apply<0>"]),
    ("K12", "GroupNorm backward", ["edadm_gn_bwd_nhwc"], [r"\bk_gnt_partial<1>", r"\bk_gnt_final<1>", r"\bk_gnt_apply<1>"]),
    ("K12", "LayerNorm forward / backward", ["edadm_ln_fwd", "edadm_ln_bwd"], [r"\bk_ln_fwd<", r"\bk_ln_bwd<"]),
    ("K12", "GEGLU forward / backward", ["edadm_geglu_fwd", "edadm_geglu_bwd"], [r"\bk_geglu_fwd\b", r"\bk_geglu_bwd\b"]),
    ("K12", "SiLU backward", ["edadm_silu_bwd"], [r"\bk_silu_bwd\b"]),
    ("K12", "softmax forward / backward", ["edadm_softmax_fwd_any", "edadm_softmax_bwd"], [r"\bk_softmax_fwd_any\b", r"\bk_softmax_bwd\b"]),
    ("K11", "operand expansion: maxima, [hi|lo] f16 split, transposing split", ["edadm_absmax_parts", "edadm_split_f16", "edadm_transpose_split_f16"],
     [r"\bk_absmax_part\b", r"\bk_split_f16\b", r"\bk_transpose_split_f16\b"]),
]


def kernel_stats(d):
    """{kernel name: (calls, total ns)} from the --stats kernel summary (falls back to summing the kernel trace)"""
    out = {}
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        for f in files:
            for r in csv.DictReader(open(f)):
                name = r.get("Name") or r.get("Kernel_Name")
                c, ns = int(r["Calls"]), float(r["TotalDurationNs"])
                a = out.get(name, (0, 0.0))
                out[name] = (a[0] + c, a[1] + ns)
        return out
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            a = out.get(name, (0, 0.0))
            out[name] = (a[0] + 1, a[1] + float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


def counter(d, which):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == which:
                a = out.get(r["Kernel_Name"], (0, 0.0))
                out[r["Kernel_Name"]] = (a[0] + 1, a[1] + float(r["Counter_Value"]))
    return out


def main():
    label, sdir, fdir, wdir, bjson, outp = sys.argv[1:7]
    stats, fetch, write = kernel_stats(sdir), counter(fdir, "FETCH_SIZE"), counter(wdir, "WRITE_SIZE")
    alg = json.load(open(bjson))
    lines = ["# %s -- HBM-bound kernel groups: rocprofv3 --kernel-trace --stats (time), --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes of the same"
             % label,
             "# command; KB -> x1024, FETCH_SIZE x2 on gfx950), algorithmic bytes accounted at the C ABI over the whole process (edadm/trace_bytes.py).",
             "# GB/s (alg) = algorithmic bytes / device time: the roofline figure (peak 8000 GB/s spec; ~6300 measured by a float4 copy).",
             "%-4s %-66s %8s %10s %10s %9s %9s %8s %8s %7s" % ("K", "group", "launches", "alg MB", "PMC MB", "PMC/alg", "ms", "GB/s alg", "GB/s PMC", "of 8TB/s")]
    tot = [0.0, 0.0, 0.0]
    for k, what, entries, pats in GROUPS:
        sel = lambda table: [(n, v) for n, v in table.items() if any(re.search(p, n) for p in pats)]
        ks = sel(stats)
        if not ks:
            continue
        launches = sum(v[0] for _, v in ks)
        ns = sum(v[1] for _, v in ks)
        fb = sum(v[1] for _, v in sel(fetch)) * 1024 * 2
        wb = sum(v[1] for _, v in sel(write)) * 1024
        ab = sum(alg.get(e, {}).get("bytes", 0) for e in entries)
        if ns <= 0:
            continue
        pmc = fb + wb
        lines.append("%-4s %-66s %8d %10.1f %10.1f %9s %9.3f %8.0f %8.0f %7.3f"
                     % (k, what, launches, ab / 1e6, pmc / 1e6, ("%.2f" % (pmc / ab)) if ab else "-", ns / 1e6, ab / ns if ab else 0.0, pmc / ns,
                        (ab / ns) / PEAK if ab else (pmc / ns) / PEAK))
        tot[0] += ab
        tot[1] += pmc
        tot[2] += ns
    lines.append("%-4s %-66s %8s %10.1f %10.1f %9.2f %9.3f %8.0f %8.0f %7.3f" % ("", "all groups above", "", tot[0] / 1e6, tot[1] / 1e6, tot[1] / max(tot[0], 1), tot[2] / 1e6,
                                                                                 tot[0] / max(tot[2], 1), tot[1] / max(tot[2], 1), tot[0] / max(tot[2], 1) / PEAK))
    lines.append("")
    open(outp, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
