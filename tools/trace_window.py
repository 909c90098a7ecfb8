"""Kernel-time statistics of the TIMED steps of bench.py only: cuts a rocprofv3 --kernel-trace CSV to the region between the two
marker kernels bench.py launches around its timed loop (unimp_marker_kernel with grid 101 x 64 and 102 x 64 threads), so model
construction, autotuning and warm-up are not in the numbers ("nobody knows what the step really pays").
usage: trace_window.py KERNEL_TRACE.csv STEPS [OUT.csv] [gaps]   -> per-kernel calls / total / average inside the window, grouped
totals (GEMM / attention / norm / optimizer / ATen / other) per step, and the window's wall time per step; with a fourth argument
also where the device idles: the largest gaps between consecutive kernels of the window and the kernels on either side."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
path, steps = sys.argv[1], int(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(path)))
name_k = "Kernel_Name"
gs = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
marks = {gs(r) // 64: int(r["End_Timestamp"]) for r in rows if "unimp_marker_kernel" in r[name_k]}
if 101 not in marks or 102 not in marks:
    sys.exit(f"markers 101 / 102 not found in {path} (found {sorted(marks)})")
t0, t1 = marks[101], marks[102]
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or e > t1 or "unimp_marker_kernel" in r[name_k]:
        continue
    a = agg[r[name_k]]
    a[0] += 1
    a[1] += e - s
    busy += e - s


def group(n):
    if "gemm" in n or "splitk" in n:
        return "GEMM"
    if "attn" in n:
        return "attention"
    if "ln_" in n:
        return "LayerNorm"
    if "adamw" in n or "sumsq" in n:
        return "optimizer"
    if "at::native" in n or "at::" in n:
        return "ATen (torch elementwise / copy / fill / index)"
    return "other unimp kernels"


groups = collections.defaultdict(lambda: [0, 0])
for n, (c, ns) in agg.items():
    g = groups[group(n)]
    g[0] += c
    g[1] += ns
lines = [f"# window between the bench markers: {(t1 - t0) / 1e6 / steps:.2f} ms wall per step over {steps} steps; kernel time {busy / 1e6 / steps:.2f} ms per step"]
for g, (c, ns) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"# {g:48s} {c / steps:9.1f} launches/step {ns / 1e6 / steps:9.3f} ms/step {100.0 * ns / busy:6.2f} %")
print("\n".join(lines))
if out:
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerStep", "MsPerStep", "AverageUs", "Percentage"])
        for n, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, round(c / steps, 2), round(ns / 1e6 / steps, 4), round(ns / 1e3 / c, 2), round(100.0 * ns / busy, 3)])

if len(sys.argv) > 4:
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k]) for r in rows
                 if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1 and "unimp_marker_kernel" not in r[name_k]))
    gaps, end = [], ks[0][1]
    for i in range(1, len(ks)):
        if ks[i][0] > end:
            gaps.append((ks[i][0] - end, i))
        end = max(end, ks[i][1])
    tot = sum(g for g, _ in gaps)
    print(f"# idle inside the window: {tot / 1e6 / steps:.3f} ms per step in {len(gaps) / steps:.0f} gaps per step; "
          f"gaps > 20 us: {sum(g for g, _ in gaps if g > 20000) / 1e6 / steps:.3f} ms, 5-20 us: {sum(g for g, _ in gaps if 5000 < g <= 20000) / 1e6 / steps:.3f} ms, "
          f"< 5 us: {sum(g for g, _ in gaps if g <= 5000) / 1e6 / steps:.3f} ms")
    by = collections.defaultdict(lambda: [0, 0])
    for g, i in gaps:
        k = (ks[i - 1][2][:50], ks[i][2][:50])
        by[k][0] += 1
        by[k][1] += g
    for (a, b), (c, ns) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"#   {ns / 1e6 / steps:7.3f} ms/step  {c / steps:6.1f} gaps/step  avg {ns / 1e3 / c:7.1f} us   after [{a}]  before [{b}]")
