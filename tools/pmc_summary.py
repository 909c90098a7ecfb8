"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name cut at '('), mean counter value per dispatch and
mean duration.  usage: python tools/pmc_summary.py DIR [DIR ...] [--match attn] -> CSV on stdout"""
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
w.writerow(["kernel", "grid", "vgpr", "agpr", "lds", "wg", "dispatches", "avg_us"] + ctrs)
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    w.writerow([k[0], k[1], *meta[k], n, round(sum(dur[k]) / len(dur[k]) / 1e3, 1)] +
               [round(sum(acc[k][c]) / len(acc[k][c]), 1) if acc[k][c] else "" for c in ctrs])
