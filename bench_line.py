"""The printed form of bench.py's result: ONE compact JSON line (<= LINE_LIMIT bytes) for the driver, everything else in a sidecar.

bench.py collects a detailed result dict (per-unit iteration times, slowest-layer tables, definitions in prose).  The driver reads
the LAST stdout line and keeps only a bounded tail of stdout, so the printed line carries numbers and short labels only; the
detail goes to `bench_detail.json` (repo root, and `gpurun_out/` when that directory exists).  Pure Python: tests/test_bench_line.py
runs it on a canned result without a GPU.
"""
import json
import os

LINE_LIMIT = 6144
STR_LIMIT = 200

_TOP = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data")
_ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel", "unet_call_ms")
_CPU = ("value", "unit", "cores", "kind", "sample")


def _num(v):
    """5 significant digits for floats: the sidecar keeps the full precision"""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.5g" % v)
    return v


def _short(s, n=STR_LIMIT):
    s = " ".join(str(s).split())
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(src, keys):
    out = {}
    for k in keys:
        if isinstance(src, dict) and k in src:
            v = src[k]
            out[k] = _short(v) if isinstance(v, str) else _num(v)
    return out


def compact(line):
    """detailed result dict -> the dict that is printed.  Never raises on a missing key: a partial result still prints."""
    out = _pick(line, _TOP)
    if "metric" in out:
        out["metric"] = _short(line["metric"], 120)
    if "dtype" in out:
        out["dtype"] = _short(line["dtype"], 60)
    cfg = line.get("config") or {}
    out["config"] = {k: (_short(v) if isinstance(v, str) else _num(v)) for k, v in cfg.items()
                     if not isinstance(v, (dict, list)) and not k.endswith(("_note", "_definition"))}
    so = line.get("sampling_only")
    if so:
        out["sampling_only"] = _pick(so, ("value", "ms_per_step"))
    one = line.get("one_batch_in_flight")
    if one:
        out["one_batch_in_flight"] = _pick(one, ("value", "sampling_only"))
    rf = line.get("roofline")
    if rf:
        r = _pick(rf, _ROOF)
        f8 = rf.get("frac_survey_8d") or {}
        r["frac_survey_8d"] = _pick(f8, ("sampling_only", "sampled_and_decoded"))
        hbm = rf.get("hbm") or {}
        if hbm:
            r["hbm"] = _pick(hbm, ("achieved_GBps", "frac"))
        fl = rf.get("in_flight") or {}
        if fl:
            r["in_flight"] = _pick(fl, ("batches", "sustained_int8_gemm_tflops", "frac"))
        out["roofline"] = r
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, _CPU)
    cal = line.get("calibration")
    if cal:
        c = _pick(cal, ("value_s", "units", "calib_samples", "iters_per_unit", "peak_hbm_gb", "graphed_units", "retried_after_oom_with_budgets_at", "error"))
        if cal.get("stages"):
            c["stages"] = {k: _num(v) for k, v in cal["stages"].items()}
        h1 = cal.get("h1_roofline")
        if h1:
            c["h1_roofline"] = _pick(h1, ("bound", "achieved", "unit", "peak_f16_three_product", "frac", "frac_of_fp32_mfma"))
        if cal.get("cpu_baseline"):
            c["cpu_baseline"] = _pick(cal["cpu_baseline"], _CPU)
        mr = cal.get("multi_rank")
        if mr:
            m = _pick(mr, ("ranks", "sharded_s_this_run", "replicated_s_this_run"))
            for key in ("ceiling", "ceiling_measured_rows"):
                if isinstance(mr.get(key), dict):
                    m[key] = {k: _num(v) for k, v in mr[key].items()}
            c["multi_rank"] = m
        rb = cal.get("reconstruction_bounded")
        if rb:
            c["reconstruction_bounded"] = _pick(rb, ("units", "calib_samples", "iters_per_unit", "wall_s", "caching_s", "loop_s"))
            if isinstance(rb.get("extrapolated_full_s"), dict):
                c["reconstruction_bounded"]["extrapolated_full_s"] = _num(rb["extrapolated_full_s"].get("total"))
        mrb = cal.get("multi_rank_bounded_walk")
        if mrb:
            c["multi_rank_bounded_walk"] = _pick(mrb, ("ranks", "units", "calib_samples", "caching_s", "gathered_bytes", "loop_s"))
        out["calibration"] = c
    fd = line.get("first_stage_decode")
    if fd:
        d = _pick(fd, ("ms_per_image", "tflops_fp32", "error"))
        if isinstance(fd.get("roofline"), dict):
            d["roofline"] = _pick(fd["roofline"], ("bound", "achieved", "peak", "unit", "frac"))
        out["first_stage_decode"] = d
    cfgs = line.get("configs")
    if cfgs:
        cc = {}
        for kind, r in cfgs.items():
            if not isinstance(r, dict):
                cc[kind] = _short(r)
                continue
            e = _pick(r, ("images_per_sec", "images_per_sec_at_500_steps", "images_per_sec_at_20_steps", "unet_call_ms", "batch",
                          "batches_in_flight", "error"))
            rr = r.get("roofline") or {}
            if rr:
                e["roofline"] = _pick(rr, ("bound", "achieved", "peak", "unit", "frac"))
                gg = rr.get("int8_gemm_group") or {}
                if gg:
                    e["roofline"]["int8_gemm_group"] = _pick(gg, ("achieved", "frac", "launches", "ms"))
            cc[kind] = e
        out["configs"] = cc
    out["detail"] = "bench_detail.json"
    return out


def dumps(line):
    """The printed line: compact separators, guaranteed <= LINE_LIMIT (optional sections dropped, largest first, if it is not)."""
    out = compact(line)
    s = json.dumps(out, separators=(",", ":"))
    for key in ("configs", "first_stage_decode", "one_batch_in_flight", "sampling_only"):
        if len(s) <= LINE_LIMIT:
            break
        out.pop(key, None)
        out["dropped_for_size"] = out.get("dropped_for_size", []) + [key]
        s = json.dumps(out, separators=(",", ":"))
    if len(s) > LINE_LIMIT:                              # cannot happen with the whitelists above; never print an unparseable line
        core = {k: out[k] for k in _TOP if k in out}
        core["roofline"] = {k: out.get("roofline", {}).get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        core["cpu_baseline"] = out.get("cpu_baseline")
        core["dropped_for_size"] = "all optional sections"
        s = json.dumps(core, separators=(",", ":"))
    return s


def write_detail(line, root):
    """The detailed result next to the repo root (and under gpurun_out/, which travels back from the GPU box)."""
    paths = [os.path.join(root, "bench_detail.json")]
    scratch = os.path.join(root, "gpurun_out")
    if os.path.isdir(scratch):
        paths.append(os.path.join(scratch, "bench_detail.json"))
    written = []
    for p in paths:
        try:
            with open(p, "w") as fh:
                json.dump(line, fh, indent=1)
            written.append(p)
        except OSError:
            pass
    return written
