# PMC counters of the wide-head attention kernel alone (tools/attn_wide_bench_plain): bash tools/prof_attn.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/tools/attn_wide_bench_plain
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d /tmp/pa1 -- $B 100 1024 > /tmp/pa1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pa2 -- $B 100 1024 > /tmp/pa2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pa1", "/tmp/pa2"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_attn_wide" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print("%-28s %16.0f per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
tail -2 /tmp/pa1.log
