"""Per-kernel MFMA utilisation and LDS statistics from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE` pass.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs) / 1024 SIMDs (256 CUs x 4).
usage: python tools/pmc_sq.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "tmf::" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
out = {}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    gui = v["GRBM_GUI_ACTIVE"] / 8.0
    if gui <= 0:
        continue
    out[k] = {"launches": cnt[k], "mfma_utilisation": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / gui / 1024.0, 4),
              "lds_bank_conflict_per_active_lds_cycle": round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"], 1), 4),
              "lds_issue_stall_fraction_of_wave_cycles": round(v["SQ_WAIT_INST_LDS"] / max(v["SQ_WAVE_CYCLES"], 1), 4)}
json.dump({"note": __doc__.split("usage")[0].strip(), "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:10]:
    print(f"{k[:60]:60s} MFMA {v['mfma_utilisation']:.3f}  LDS conflicts {v['lds_bank_conflict_per_active_lds_cycle']:.3f}")
