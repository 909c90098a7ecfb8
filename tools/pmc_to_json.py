"""profiles/rNN_pmc_gemm.{csv,json} from rocprofv3 --pmc passes over tools/pmc_gemm_step.py (see tools/gpu/final.sh).
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE [KB] reports half of a wide coalesced stream -> x2;
WRITE_SIZE [KB] exact; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32_bf16
summed over all SIMDs, so MFMA utilisation = busy / (GRBM / 8 * 1024 SIMDs).  usage: pmc_to_json.py PMC_DIR... OUT_PREFIX M"""
import csv, json, sys, collections, glob, os
csv.field_size_limit(1 << 30)
*dirs, out, M = sys.argv[1:]
M = int(M)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Dispatch_Id"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
# dispatches come in groups of 3 per shape, in the workload's order
shapes = [(10240, 2560, "fwd up-proj (KC,KC)"), (2560, 10240, "fwd down-proj (KC,KC)"), (10240, 2560, "dX (KC,KS)"), (2560, 10240, "dX (KC,KS)")]
keys = sorted(acc, key=lambda k: k[1])
rows = []
for si, (n, k, what) in enumerate(shapes):
    grp = keys[3 * si:3 * si + 3]
    if len(grp) < 3:
        break
    g = lambda c: sum(sum(acc[q][c]) / max(1, len(acc[q][c])) for q in grp) / 3
    fetch, write, busy, grbm = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024, g("SQ_VALU_MFMA_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE")
    alg = (M * k + n * k + M * n) * 2
    rows.append(dict(kernel=grp[0][0], shape=[M, n, k], what=what, fetch_bytes=fetch, write_bytes=write, traffic_bytes_per_launch=fetch + write,
                     algorithmic_bytes=alg, traffic_over_algorithmic=round((fetch + write) / alg, 3), mfma_busy_cycles=busy, grbm_gui_active=grbm,
                     mfma_util=round(busy / (grbm / 8 * 1024), 4) if grbm else None))
with open(out + ".csv", "w") as f:
    f.write("# " + __doc__.replace("\n", "\n# ") + "\n")
    w = csv.DictWriter(f, fieldnames=list(rows[0]))
    w.writeheader()
    for r in rows:
        w.writerow(r)
top = rows[0]
json.dump({"kernel": top["kernel"], "shape": top["shape"], "traffic_bytes_per_launch": top["traffic_bytes_per_launch"], "mfma_util": top["mfma_util"],
           "source": os.path.basename(out) + ".csv (rocprofv3 --pmc passes on the final tree of the round, tools/gpu/final.sh)"}, open(out + ".json", "w"))
print(json.dumps(rows, indent=1))
