"""profiles/rNN_pmc_gemm.{csv,json} from rocprofv3 --pmc passes over tools/pmc_gemm_step.py (see tools/gpu/final.sh).
Rows are joined to the workload's manifest by (kernel instance, grid size); the last 3 dispatches of each key are averaged (the
first is the warm-up) and a case whose key does not match exactly 4 dispatches in every pass is reported as an error, not guessed.
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE [KB] reports half of a wide coalesced stream -> x2;
WRITE_SIZE [KB] exact; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32_bf16
summed over all SIMDs, so MFMA utilisation = busy / (GRBM / 8 * 1024 SIMDs); duration from the pass's own dispatch timestamps.
usage: pmc_to_json.py MANIFEST.json PMC_DIR... OUT_PREFIX"""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
man_path, *dirs, out = sys.argv[1:]
manifest = json.load(open(man_path))
# (kernel name, grid) -> counter -> {dispatch id: [values]}, and -> dispatch id -> duration ns
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
dur = collections.defaultdict(dict)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "")
            if "gemm" not in name:
                continue
            name = name.split("(")[0]
            key = (name, int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]][(d, int(r["Dispatch_Id"]))].append(float(r["Counter_Value"]))
            dur[key][(d, int(r["Dispatch_Id"]))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows, errors = [], []
for c in manifest:
    keys = [k for k in acc if k[0].startswith(c["kernel"]) and k[1] == c["grid_size"]]
    if len(keys) != 1:
        errors.append(f"{c['label']}: {len(keys)} kernel instances match ({c['kernel']}, grid {c['grid_size']}): {[k[0] for k in keys]}")
        continue
    key = keys[0]

    def mean_last3(counter):
        per = acc[key].get(counter)
        if not per:
            return None
        by_pass = collections.defaultdict(list)
        for (d, did), vals in per.items():
            by_pass[d].append((did, sum(vals)))          # a counter may be reported per XCD / instance: summed per dispatch
        out_ = []
        for d, lst in by_pass.items():
            lst.sort()
            if len(lst) != 4:
                errors.append(f"{c['label']}: {len(lst)} dispatches of {key} in {d} (expected 1 warm-up + 3)")
            out_ += [v for _, v in lst[-3:]]
        return sum(out_) / len(out_)
    fetch, write = mean_last3("FETCH_SIZE"), mean_last3("WRITE_SIZE")
    busy, grbm = mean_last3("SQ_VALU_MFMA_BUSY_CYCLES"), mean_last3("GRBM_GUI_ACTIVE")
    by_pass = collections.defaultdict(list)
    for (d, did), ns in dur[key].items():
        by_pass[d].append((did, ns))
    ds = [ns for lst in by_pass.values() for _, ns in sorted(lst)[-3:]]
    ms = sum(ds) / len(ds) / 1e6
    fb = fetch * 1024 * 2 if fetch is not None else None
    wb = write * 1024 if write is not None else None
    tr = fb + wb if fb is not None and wb is not None else None
    rows.append(dict(label=c["label"], kernel=key[0], grid_size=key[1], variant=c["variant"], shape=c["shape"], a_kstrided=c["a_kstrided"],
                     b_kstrided=c["b_kstrided"], epilogue=c["epilogue"], ms_under_pmc=round(ms, 4), tflops_under_pmc=round(c["flop"] / ms / 1e9, 1),
                     fetch_bytes=fb, write_bytes=wb, traffic_bytes_per_launch=tr, algorithmic_bytes=c["algorithmic_bytes"],
                     traffic_over_algorithmic=round(tr / c["algorithmic_bytes"], 3) if tr else None, mfma_busy_cycles=busy, grbm_gui_active=grbm,
                     mfma_util=round(busy / (grbm / 8 * 1024), 4) if busy and grbm else None,
                     clock_ghz=round(grbm / 8 / (ms * 1e6), 3) if grbm else None))
for e in errors:
    print("ERROR:", e, file=sys.stderr)
if not rows:
    sys.exit("no rows")
with open(out + ".csv", "w") as f:
    f.write("# " + __doc__.replace("\n", "\n# ") + "\n")
    w = csv.DictWriter(f, fieldnames=list(rows[0]))
    w.writeheader()
    for r in rows:
        w.writerow(r)
top = rows[0]         # the manifest leads with the kernel instance that has the largest share of the step
json.dump({"kernel": top["kernel"], "label": top["label"], "shape": top["shape"], "traffic_bytes_per_launch": top["traffic_bytes_per_launch"],
           "algorithmic_bytes": top["algorithmic_bytes"], "mfma_util": top["mfma_util"], "rows": rows, "errors": errors,
           "source": os.path.basename(out) + ".csv (rocprofv3 --pmc passes, tools/gpu/final.sh; rows keyed on (kernel instance, grid size))"},
          open(out + ".json", "w"), indent=1)
print(json.dumps(rows, indent=1))
sys.exit(1 if errors else 0)
