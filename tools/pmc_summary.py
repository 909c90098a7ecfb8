"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name cut at '('), mean counter value per dispatch and
mean duration, plus DERIVED columns where their counters were collected (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs):
  clock_ghz          = GRBM_GUI_ACTIVE / 8 / duration
  mfma_util          = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   -- matrix pipe busy share at the clock the chip held
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_share         = SQ_WAIT_ANY / SQ_WAVE_CYCLES ; issue_stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
usage: python tools/pmc_summary.py DIR [DIR ...] [--match attn] -> CSV on stdout"""
import csv, sys, collections, glob, os
csv.field_size_limit(1 << 30)
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
meta = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if match and match not in name:
                continue
            key = (name, r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"]))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
ctrs = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout)
derived = ["clock_ghz", "mfma_util", "lds_conflict_share", "wait_share", "issue_stall_share"]
w.writerow(["kernel", "grid", "vgpr", "agpr", "lds", "wg", "dispatches", "avg_us"] + derived + ctrs)
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    m = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else None) for c in ctrs}
    us = sum(dur[k]) / len(dur[k]) / 1e3
    g = lambda c: m.get(c)
    ratio = lambda a, b: round(g(a) / g(b), 4) if g(a) is not None and g(b) else ""
    d = [round(g("GRBM_GUI_ACTIVE") / 8 / (us * 1e3), 3) if g("GRBM_GUI_ACTIVE") and us else "",
         round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024), 4) if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE") else "",
         ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"), ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")]
    w.writerow([k[0], k[1], *meta[k], n, round(us, 1)] + d + [round(m[c], 1) if m[c] is not None else "" for c in ctrs])
