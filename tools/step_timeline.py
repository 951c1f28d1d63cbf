#!/usr/bin/env python3
"""Timeline of one timed bench step from a rocprofv3 kernel trace: every dispatch and memory copy between the
last two launches of the timed search kernel, with the idle gap in front of each.
Usage: python tools/step_timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [kernel substring]"""
import csv, glob, sys

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_search_fast"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
idx = [i for i, e in enumerate(ev) if key in e[2]]
if len(idx) < 2:
    sys.exit("fewer than two launches of " + key)
a, b = idx[-2], idx[-1]
t0 = ev[a][0]
prev_end = ev[a][0]
print(f"step = {1e-6 * (ev[b][0] - ev[a][0]):.3f} ms between the last two launches of {key}")
for s, e, n in ev[a:b]:
    print(f"{1e-6 * (s - t0):9.3f} ms  +gap {1e-3 * max(0, s - prev_end):8.1f} us  dur {1e-3 * (e - s):9.1f} us  {n}")
    prev_end = max(prev_end, e)
