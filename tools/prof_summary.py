"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, busy vs idle time, and (for the GEMM kernels) time by grid size."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5   # fraction of the trace (from the start) to drop as warm-up
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
busy, last_end, by, byg = 0, 0, collections.Counter(), collections.Counter()
cnt = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - max(s, last_end)) if e > last_end else 0
    last_end = max(last_end, e)
    name = r["Kernel_Name"][:70]
    by[name] += e - s
    cnt[name] += 1
    if "conv_gemm" in name or "wgrad" in name:
        byg[(name[:48], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))] += e - s
print(f"window {span:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/1e6/span:.1f}%), launches {len(rows)}")
for k, v in by.most_common(22):
    print(f"{v/1e6:9.3f} ms {cnt[k]:6d}  {k}")
print("--- GEMM kernels by grid size (x = threads)")
for (k, g), v in sorted(byg.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{v/1e6:9.3f} ms  grid {g:>9s}  {k}")
