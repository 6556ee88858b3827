"""HBM bytes per launch of the int8 GEMM / convolution kernels from rocprofv3 --pmc counter CSVs (separate FETCH_SIZE and
WRITE_SIZE passes), per kernel, against each kernel's own ALGORITHMIC bytes:
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> <launch_list.json> [N_CALLS]
The kernels are selected by the ENGINE'S launch list (tools/unet_prof.py LAUNCH_LIST=...: every int8 layer launch of one UNet
call with the kernel structure the library picked), not by a hand-kept substring list: a kernel the list names that the profile
does not hold, a profile row of the GEMM family that no launch claims, or a launch count that is not a whole number of UNet calls
fails the run (round 4 silently dropped k_gemm_ntq).  Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
counters are in KB (x1024); on gfx950 FETCH_SIZE reports half of a wide (16 B/lane) streaming read (x2); WRITE_SIZE is exact."""
import csv, glob, hashlib, json, os, re, sys

# kernel structure (tag of edadm_diag_launch_kernels) -> pattern of its int8 instantiations in a kernel trace
PATTERNS = {
    "k_gemm_nt": r"\bk_gemm_nt<0,", "k_gemm_nt8": r"\bk_gemm_nt8<0,", "k_gemm_p": r"\bk_gemm_p<0,", "k_gemm_ntq": r"\bk_gemm_ntq<",
    "k_conv3_direct": r"\bk_conv3_direct<0,", "k_gemm_split2": r"\bk_gemm_split2<", "k_gemm_br": r"\bk_gemm_b[rw]<",
}
# the same templates on other operand types (f16 attention products, fp32 / three-product f16 calibration graph): not int8 layers
# ... and the packed-nibble kernel of the few-row layers (launch-list type "w4": weights are its traffic, a group of its own)
OTHER = r"\bk_(gemm_nt|gemm_nt8|gemm_p|conv3_direct)<[123],|\bk_gemm_w4\b"
FAMILY = r"\bk_(gemm|conv3_direct)"


def load(d, counter):
    """[(dispatch id, kernel name, counter value)] in dispatch order"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def tail_sums(rows, per_call, calls):
    """per structure: the LAST calls x per_call[structure] dispatches of the run -- the UNet calls unet_prof.py issues after its
    MARK.  Everything earlier (scale initialisation, the hoisted one-token context branches and time-embedding rows, warm-up calls,
    the launch-list pass) is left out by construction instead of by arithmetic on launch counts."""
    by = {}
    for _, name, v in rows:
        k = classify(name)
        if k is not None:
            by.setdefault(k, []).append((name, v))
    out = {}
    for k, n in per_call.items():
        if k not in by:
            raise SystemExit("the launch list names %s, the profile holds no such kernel" % k)
        want = n * calls
        if len(by[k]) < want:
            raise SystemExit("%s: %d dispatches in the profile, %d needed for %d UNet calls" % (k, len(by[k]), want, calls))
        tail = by[k][-want:]
        out[k] = {"launches": want, "KB": sum(v for _, v in tail), "names": sorted(set(nm.split("(")[0] for nm, _ in tail))}
    for k in by:
        if k not in per_call:
            raise SystemExit("the profile holds int8 %s launches, the launch list names none" % k)
    return out


def classify(name):
    hits = [k for k, p in PATTERNS.items() if re.search(p, name)]
    if len(hits) > 1:
        raise SystemExit("kernel name matches two structures: %s %s" % (name, hits))
    if hits:
        return hits[0]
    if re.search(FAMILY, name) and not re.search(OTHER, name):
        raise SystemExit("a GEMM-family kernel that no launch-list structure claims (add it to PATTERNS): " + name)
    return None


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    lst = json.load(open(sys.argv[4]))
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 2       # N_CALLS of unet_prof.py: the calls after its MARK
    # per structure: launches and algorithmic bytes of ONE UNet call (a launch that issued two kernels -- tail re-tiling -- is
    # split between them by launch count only: its bytes stay with the first, stated in the output)
    alg = {}
    for r in lst["rows"]:
        if r["type"] != "i8":
            continue
        if not r["kernels"] or "?" in r["kernels"]:
            raise SystemExit("launch without a known kernel tag: %r" % r)
        b = sum(r["bytes"].values())
        for i, k in enumerate(r["kernels"]):
            a = alg.setdefault(k, {"launches": 0, "bytes": 0.0, "flop": 0.0})
            a["launches"] += 1
            if i == 0:
                a["bytes"] += b
                a["flop"] += r["flop"]
    per_call = {k: a["launches"] for k, a in alg.items()}
    fs, ws = tail_sums(fetch, per_call, calls), tail_sums(write, per_call, calls)
    per_kernel = {}
    for k, a in sorted(alg.items()):
        hbm = (fs[k]["KB"] * 2 + ws[k]["KB"]) * 1024 / calls
        per_kernel[k] = {"names": fs[k]["names"], "launches_per_unet_call": a["launches"],
                         "fetch_bytes_per_call_corrected": fs[k]["KB"] * 2 * 1024 / calls, "write_bytes_per_call": ws[k]["KB"] * 1024 / calls,
                         "hbm_bytes_per_call": hbm, "algorithmic_bytes_per_call": a["bytes"], "traffic_over_algorithmic": hbm / a["bytes"] if a["bytes"] else None,
                         "algorithmic_flop_per_call": a["flop"]}
    nl = sum(a["launches"] for a in alg.values())
    hbm_call = sum(v["hbm_bytes_per_call"] for v in per_kernel.values())
    alg_call = sum(a["bytes"] for a in alg.values())
    out = {
        "kernel": "every int8 GEMM / convolution launch of one UNet call of the frozen LDM-4 engine (%s)" % ", ".join(sorted(per_kernel)),
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 "
                   "tools/unet_prof.py  [100 rows = one guidance pair of 50; LAUNCH_LIST: the engine's launch list]",
        "unet_calls": calls, "launches_per_unet_call": nl,
        "correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE reports 1/2 of a wide (16 B/lane) streaming read on gfx950 -> x2; WRITE_SIZE exact; unit KB -> x1024",
        "hbm_bytes_per_unet_call": hbm_call, "algorithmic_bytes_per_unet_call": alg_call, "traffic_over_algorithmic": hbm_call / alg_call,
        "hbm_bytes_per_launch": hbm_call / nl, "algorithmic_bytes_per_launch": alg_call / nl,
        "fetch_bytes_per_launch_corrected": sum(v["fetch_bytes_per_call_corrected"] for v in per_kernel.values()) / nl,
        "write_bytes_per_launch": sum(v["write_bytes_per_call"] for v in per_kernel.values()) / nl,
        "launches": nl * calls,
        "per_kernel": per_kernel,
        "tail_note": "a layer launch that issued two kernels (tail re-tiling: k_gemm_nt8 + k_gemm_nt, two k_conv3_direct tiles) counts its "
                     "algorithmic bytes with the first",
    }
    # the sources these bytes belong to: bench.py drops the file (roofline.traffic = null) when csrc/gemm.hip has changed since
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "eda-dm_amd", "csrc", "gemm.hip"), "rb") as fh:
        out["gemm_hip_sha256"] = hashlib.sha256(fh.read()).hexdigest()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("unet_calls", "launches_per_unet_call", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
    for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["hbm_bytes_per_call"]):
        print("%-16s %4d launches/call  hbm %8.1f MB  algorithmic %8.1f MB  x%.3f" % (k, v["launches_per_unet_call"], v["hbm_bytes_per_call"] / 1e6,
                                                                                     v["algorithmic_bytes_per_call"] / 1e6, v["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
