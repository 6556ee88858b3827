"""Aggregate a rocprofv3 --pmc SQ pass into per-kernel MFMA utilisation: python tools/pmc_mfma.py <dir>
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY quad-cycles (MI355X_MICROARCH.md, cycle
constants: 's_memtime tick vs SQ PMC units'); utilisation = MFMA-busy cycles / (4 x wave quad-cycles / waves per SIMD), printed per
kernel name with the wave's time split into issuing / issue-stalled / parked."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
rows = []
for k, c in acc.items():
    wc = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    rows.append((wc, k, c))
rows.sort(reverse=True)
print("%-46s %8s %10s %8s %8s %8s %8s %10s" % ("kernel", "launches", "MFMA busy", "issuing", "stalled", "parked", "LDS act", "bank confl"))
for wc, k, c in rows[:16]:
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("%-46s %8d %9.1f%% %7.1f%% %7.1f%% %7.1f%% %7.1f%% %10.0f" % (
        k[:46], cnt[k], 100.0 * mf / wc * 1.0, 100.0 * 4 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100.0 * 4 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
        100.0 * 4 * c.get("SQ_WAIT_ANY", 0) / wc, 100.0 * c.get("SQ_LDS_IDX_ACTIVE", 0) / wc, c.get("SQ_LDS_BANK_CONFLICT", 0)))
print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / wave cycles: the share of a WAVE's lifetime during which the matrix pipe of its SIMD was busy\n"
      "with MFMAs of any wave on that SIMD is not separable here; with W waves per SIMD the SIMD's utilisation is W x this / ... -- read as\n"
      "a per-wave figure: 100 % / waves per SIMD would be a saturated pipe.")
