"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream, dependent launches):
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python tools/profile_ops.py --reps 1
    python tools/trace_gaps.py /tmp/tr
Prints the busy time, the summed gaps and the gap histogram of the LAST forward in the trace (`NFWD` kernels, default: auto)."""
import csv, glob, os, sys
d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last forward = from the last time_embed kernel on
last = max(i for i, n in enumerate(names) if "time_embed" in n)
rows = rows[last:]
# stop at the next kernel that is not part of the forward (none after the last forward in profile_ops)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, gaps {sum(gaps) / 1e6:.3f} ms "
      f"(mean {sum(gaps) / len(gaps) / 1e3:.2f} us, min {min(gaps) / 1e3:.2f}, max {max(gaps) / 1e3:.2f})")
by = {}
for a, g in zip(rows, gaps):
    k = a["Kernel_Name"].split("(")[0][-40:]
    by.setdefault(k, []).append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  after {k:42s} n={len(v):4d} mean gap {sum(v) / len(v) / 1e3:6.2f} us")
