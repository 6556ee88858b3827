"""Aggregate rocprofv3 --pmc counter CSVs (separate FETCH_SIZE and WRITE_SIZE passes) into HBM bytes per launch of
the int8 GEMM kernels: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are in KB (x1024); on gfx950 FETCH_SIZE
reports half of a wide (16 B/lane) streaming read (x2); WRITE_SIZE is exact."""
import csv, glob, json, os, sys


def load(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            c = per.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
is_i8 = lambda k: ("k_gemm_nt8<0" in k) or ("k_gemm_nt<0" in k) or ("k_gemm_p<0" in k) or ("k_conv3_direct<0" in k) or ("k_gemm_split2" in k)
nl = sum(v[0] for k, v in fetch.items() if is_i8(k))
fkb = sum(v[1] for k, v in fetch.items() if is_i8(k))
wkb = sum(v[1] for k, v in write.items() if is_i8(k))
out = {
    "kernel": "int8 GEMM launches of edadm_qgemm_i8 / _q / edadm_qconv3_i8_direct (k_gemm_nt8<0,*>, k_gemm_p<0,*>, k_gemm_nt<0,*>, k_conv3_direct<0,3,2> / <0,3,1>, k_gemm_split2<3,1>)",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python "
               "tools/unet_prof.py  [N_CALLS=2: 4 eager UNet calls of the frozen LDM-4 engine as a DDIM step issues them, 100 rows]",
    "launches": nl, "unet_calls": int(sys.argv[4]) if len(sys.argv) > 4 else 4, "FETCH_SIZE_sum_KB": fkb, "WRITE_SIZE_sum_KB": wkb,
    "correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE reports 1/2 of a wide (16 B/lane) streaming read on gfx950 -> x2; WRITE_SIZE exact; unit KB -> x1024",
    "fetch_bytes_per_launch_corrected": fkb * 2 * 1024 / max(nl, 1),
    "write_bytes_per_launch": wkb * 1024 / max(nl, 1),
    "hbm_bytes_per_launch": (fkb * 2 + wkb) * 1024 / max(nl, 1),
    "per_kernel": {k: {"launches": v[0], "fetch_KB": v[1], "write_KB": write.get(k, [0, 0.0])[1]}
                   for k, v in sorted(fetch.items(), key=lambda kv: -kv[1][1])[:24]},
}
# the sources these bytes belong to: bench.py drops the file (roofline.traffic = null) when csrc/gemm.hip has changed since
import hashlib
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(_root, "eda-dm_amd", "csrc", "gemm.hip"), "rb") as _fh:
    out["gemm_hip_sha256"] = hashlib.sha256(_fh.read()).hexdigest()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches", "hbm_bytes_per_launch", "fetch_bytes_per_launch_corrected", "write_bytes_per_launch")}))
