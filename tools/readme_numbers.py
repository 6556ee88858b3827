"""README.md = docs/README.template.md with its @@KEY@@ fields filled from a bench line (the driver-format JSON that `python bench.py`
writes next to its printed line, bench_detail.json): the numbers in the README are regenerated, not hand-edited.
    python tools/readme_numbers.py profiles/r06z_bench_detail.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
c = d["calibration"]
st = c["stages"]
cfg = d.get("configs", {})
ce = c["multi_rank"]["ceiling"]
f = {
    "VALUE": "%.1f" % d["value"], "SAMPLING": "%.1f" % d["sampling_only"]["value"], "FRAC": "%.3f" % d["roofline"]["frac"],
    "DECODE": "%.2f" % d["first_stage_decode"]["ms_per_image"], "CALIB": "%.1f" % c["wall_s"], "TDAC": "%.1f" % st["tdac_s"],
    "SCALE": "%.1f" % st["scale_init_s"], "RECON": "%.1f" % st["reconstruction_s"],
    "CEIL": "%.2f / %.2f / %.2f x" % (ce["2"], ce["4"], ce["8"]), "CPU": "%.4f" % d["cpu_baseline"]["value"],
    "CPUIT": "%.2f" % c["cpu_baseline"]["value"], "CIFAR": "%.0f" % cfg["cifar"]["images_per_sec"],
    "CHURCH": "%.1f" % cfg["church"]["images_per_sec_at_500_steps"], "SD": "%.2f" % cfg["sd"]["images_per_sec"],
    "ONE": "%.1f" % d["one_batch_in_flight"]["value"], "ONE_S": "%.1f" % d["one_batch_in_flight"]["sampling_only"],
    "CEILM": "%.2f / %.2f / %.2f x" % tuple(c["multi_rank"].get("ceiling_measured_rows", ce)[k] for k in ("2", "4", "8")),
}
text = open(os.path.join(ROOT, "docs", "README.template.md")).read()
for k, v in f.items():
    text = text.replace("@@%s@@" % k, v)
assert "@@" not in text, [w for w in text.split() if "@@" in w][:5]
open(os.path.join(ROOT, "README.md"), "w").write(text)
print("README.md written from", sys.argv[1])
