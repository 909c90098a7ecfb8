"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name cut at '('), mean counter value per dispatch and
mean duration, plus DERIVED columns where their counters were collected (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs).  Round 6: every derived column is computed PER
DISPATCH from counters and the duration of THAT dispatch -- one pass, one clock -- and then averaged; rounds 4-5 divided one pass's
GRBM_GUI_ACTIVE by the mean duration over ALL passes (a counter pass stretches the kernels differently per counter set: 2.7 "GHz").
  clock_ghz          = GRBM_GUI_ACTIVE / 8 / that dispatch's duration          (sanity: <= 2.4, the chip's maximum; flagged in clock_ok)
  mfma_util          = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   -- matrix pipe busy share at the clock the chip held
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_share         = SQ_WAIT_ANY / SQ_WAVE_CYCLES ; issue_stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  avg_us             = mean duration over every dispatch of every pass; avg_us_clock_pass = over the dispatches that carry GRBM_GUI_ACTIVE
usage: python tools/pmc_summary.py DIR [DIR ...] [--match attn] -> CSV on stdout"""
import csv, sys, collections, glob, os
csv.field_size_limit(1 << 30)
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
der = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
DERIVED = {"clock_ghz": lambda c, ns: c["GRBM_GUI_ACTIVE"] / 8 / ns,
           "mfma_util": lambda c, ns: c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024),
           "lds_conflict_share": lambda c, ns: c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
           "wait_share": lambda c, ns: c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
           "issue_stall_share": lambda c, ns: c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
           # L2 / fabric side (round 6): hit rate of the L2's tag lookups; share of the L2's fabric read requests whose ADDRESS belongs to local DRAM (the other
           # destinations are GMI = another GPU and IO; an Infinity-Cache hit is still "destined for DRAM": the MALL sits behind this counter, its hits are
           # not exposed by any TCC counter of this stack -- rocprofv3 -L, profiles/r06_tcc_counters_available.txt); share of 128-byte read requests
           "l2_hit_rate": lambda c, ns: c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
           "fabric_rd_dram_share": lambda c, ns: c["TCC_EA0_RDREQ_DRAM_sum"] / c["TCC_EA0_RDREQ_sum"],
           "fabric_rd_128B_share": lambda c, ns: c["TCC_EA0_RDREQ_128B_sum"] / c["TCC_EA0_RDREQ_sum"],
           "fabric_rd_GBps": lambda c, ns: (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_128B_sum"] - c["TCC_EA0_RDREQ_32B_sum"]) * 64 / ns + c["TCC_EA0_RDREQ_128B_sum"] * 128 / ns + c["TCC_EA0_RDREQ_32B_sum"] * 32 / ns}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        disp = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if match and match not in name:
                continue
            key = (name, r["Grid_Size"])
            e = disp.setdefault(r["Dispatch_Id"], (key, {}, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            e[1][r["Counter_Name"]] = e[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
        for key, c, ns in disp.values():
            dur[key].append(ns)
            for n, v in c.items():
                acc[key][n].append(v)
            if "GRBM_GUI_ACTIVE" in c:
                der[key]["_us_clock_pass"].append(ns / 1e3)
            for n, fn in DERIVED.items():
                try:
                    der[key][n].append(fn(c, ns))
                except (KeyError, ZeroDivisionError):
                    pass
ctrs = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "vgpr", "agpr", "lds", "wg", "dispatches", "avg_us", "avg_us_clock_pass", "clock_ok"] + list(DERIVED) + ctrs)
mean = lambda v: sum(v) / len(v) if v else None
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    m = {c: mean(acc[k][c]) for c in ctrs}
    us = mean(dur[k]) / 1e3
    dv = {n_: mean(der[k][n_]) for n_ in DERIVED}
    ck = "" if dv["clock_ghz"] is None else ("ok" if dv["clock_ghz"] <= 2.45 else "IMPOSSIBLE")
    upc = mean(der[k]["_us_clock_pass"])
    w.writerow([k[0], k[1], *meta[k], n, round(us, 1), round(upc, 1) if upc else "", ck] +
               [round(dv[n_], 4) if dv[n_] is not None else "" for n_ in DERIVED] + [round(m[c], 1) if m[c] is not None else "" for c in ctrs])
