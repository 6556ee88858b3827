# VALU / MFMA pipe occupancy of the weight-resident kernels (k_gemm_bw, k_gemm_br) on their production shapes: one SQ counter pass over
# tools/gemm_br_bench.py, raw sums per kernel.   bash tools/prof_gemm_bw.sh  -> gpurun_out/${TAG}_gemm_bw_pmc.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r06z}
rm -rf /tmp/pbw
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/pbw -- python3 $R/tools/gemm_br_bench.py --reps 4 > /tmp/pbw.log 2>&1
python3 - <<'PY' > $R/gpurun_out/${T}_gemm_bw_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("/tmp/pbw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_gemm_b" not in k and "k_gemm_p" not in k and "k_gemm_ntq" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print("per launch (sums over all waves / SIMDs of the chip; SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* in quad-cycles, SQ_BUSY_CYCLES and")
print("SQ_VALU_MFMA_BUSY_CYCLES in cycles -- MI355X_MICROARCH.md).  VALU issue share = 4 x SQ_ACTIVE_INST_VALU / (4 x SQ_WAVE_CYCLES / waves per SIMD)")
for k, c in sorted(acc.items()):
    L = max(len(n[k]), 1)
    print(k[:60], "launches", L)
    for name in sorted(c):
        print("   %-28s %14.0f" % (name, c[name] / L))
    if c.get("SQ_INSTS_VALU"):
        print("   cycles of VALU issue per VALU instruction (4 x ACTIVE_INST_VALU / INSTS_VALU): %.2f" % (4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]))
PY
cat $R/gpurun_out/${T}_gemm_bw_pmc.txt
