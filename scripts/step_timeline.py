#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (CSV) of bench.py and prints the timeline of its last steps:
every kernel outside the iteration loop with the idle time in front of it, and the loop as one
line.   scripts/step_timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys


def short(name):
    for key in ("mel_native", "finalize", "peak_scale", "copyBuffer", "fillBuffer", "init_native"):
        if key in name:
            return key
    if "gl_engine" in name:
        return "engine" + name[name.index("<"):name.index(">") + 1]
    return name[:40]


def main():
    path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "mel_native" in r["Kernel_Name"]]
    first = starts[-4] if len(starts) >= 4 else starts[0]
    t0 = int(rows[first]["Start_Timestamp"])
    prev_end = None
    loop = None
    for r in rows[first:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = short(r["Kernel_Name"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        is_iter = name.startswith("engine") and name.rstrip(">").endswith(("8, false", "0, false", "6, false"))
        if is_iter:
            if loop is None:
                loop = [s, e, 1, gap, 0.0]
            else:
                loop[1], loop[2], loop[4] = e, loop[2] + 1, loop[4] + gap
        else:
            if loop:
                print(f"{'iteration loop':30s} at {(loop[0]-t0)/1e3:9.1f} us  {loop[2]} launches, {(loop[1]-loop[0])/1e3:8.1f} us, "
                      f"idle before {loop[3]:6.1f}, idle inside {loop[4]:6.1f}")
                loop = None
            print(f"{name:30s} at {(s-t0)/1e3:9.1f} us  {(e-s)/1e3:8.1f} us, idle before {gap:6.1f}")
        prev_end = e


if __name__ == "__main__":
    main()
