#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace csv and prints, for a window of launches, start/end offsets, queue ids and how much
kernels of different streams overlapped.   python tools/timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if "--all" not in sys.argv:
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_trace", "k_shade", "k_raygen", "k_resolve"))]
sys.argv = [a for a in sys.argv if a != "--all"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
if first < 0:  # negative: start at the |first|-th k_trace launch
    tr = [i for i, r in enumerate(rows) if "k_trace" in r["Kernel_Name"]]
    first = tr[min(-first, len(tr) - 1)]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hr::", "")
    print(f"{name:12s} queue {r.get('Queue_Id', '?'):>3s} stream {r.get('Stream_Id', '?'):>3s}  start {s / 1e3:9.1f} us  end {e / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap-after-prev-end {(s - prev_end) / 1e3:8.1f}")
    prev_end = max(prev_end, e)
total = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"span {total / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms (ratio {busy / total:.2f}: > 1 means overlap)")
