#!/usr/bin/env python3
"""Condenses the rocprofv3 output of scripts/profile_gpu.sh into
gpurun_out/prof_<tag>/summary_<tag>.{md,json}: per-kernel time statistics and, for the
Griffin-Lim iteration kernel, HBM bytes per launch from FETCH_SIZE / WRITE_SIZE.

Counter handling follows the MI355X notes: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for wide stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(outdir, sub, suffix):
    hits = glob.glob(os.path.join(outdir, sub, "**", f"*{suffix}"), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None   # the newest, if older runs share the directory


def short(name):
    for key in ("gl_persist_kernel", "gl_engine_kernel", "gemm_nt_kernel", "mel_native_kernel", "root_kernel", "init_native_kernel", "finalize_kernel",
                "peak_reduce_kernel", "peak_scale_kernel", "sc_reduce_kernel", "spec_convert_kernel",
                "pad_signal_kernel"):
        if key in name:
            if key in ("gl_persist_kernel", "gl_engine_kernel", "gemm_nt_kernel"):
                # template args <A, HB, MODE, SC>
                tail = name[name.index(key):]
                return tail[: tail.index(">") + 1] if ">" in tail else tail[:60]
            return key
    return name[:60]


def main():
    outdir, tag = sys.argv[1], sys.argv[2]
    lines = [f"# rocprofv3 summary {tag}", ""]
    summary = {}
    stats = find(outdir, "kt", "kernel_stats.csv")
    if stats:
        lines += ["## kernel-trace --stats", "", "| kernel | calls | total ms | avg us | min us | max us | % |",
                  "|---|---|---|---|---|---|---|"]
        with open(stats) as fh:
            for row in csv.DictReader(fh):
                name = short(row["Name"])
                avg_us = float(row["AverageNs"]) / 1e3
                lines.append(f"| {name} | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | {avg_us:.2f} | "
                             f"{float(row['MinNs'])/1e3:.2f} | {float(row['MaxNs'])/1e3:.2f} | {float(row['Percentage']):.2f} |")
                summary.setdefault("kernels", {})[name] = {"calls": int(row["Calls"]), "avg_us": avg_us}
        lines.append("")
    # the bench line printed by the profiled process itself: its HIP-event time per launch of
    # the iteration kernel, to set beside the profiler's average for the same launches
    log = os.path.join(outdir, "kt.log")
    if os.path.exists(log):
        for text in open(log, errors="replace"):
            if text.startswith("{") and '"roofline"' in text:
                roof = json.loads(text)["roofline"]
                summary["bench_under_profiler_us_per_launch"] = roof["kernel_ms_per_launch"] * 1e3
                summary["iterations_per_launch"] = roof.get("iterations_per_launch", 1)
                # which workload, variant and kernel sources these counters belong to
                summary["traffic_key"] = roof.get("traffic_key")
                summary["engine_source_digest"] = roof.get("engine_source_digest")
                lines += ["## the profiled process's own measurement", "",
                          f"HIP events around the {roof['kernel']} launches, in the same process the table above was "
                          f"taken from: {roof['kernel_ms_per_launch'] * 1e3:.2f} us per launch (kernel-trace serialises "
                          "dispatches and times each one from its own start to its own end, so it reads a few "
                          "percent above the back-to-back figure; without the profiler the same events give the "
                          "bench line's value).", ""]
    per = {}
    for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = find(outdir, sub, "counter_collection.csv")
        if not path:
            continue
        acc = defaultdict(list)
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        per[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
    if per:
        lines += ["## HBM traffic per launch (PMC, separate passes)", "",
                  "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | read MB (2x FETCH) | write MB | total MB |",
                  "|---|---|---|---|---|---|"]
        names = sorted(set(per.get("FETCH_SIZE", {})) | set(per.get("WRITE_SIZE", {})))
        for k in names:
            f = per.get("FETCH_SIZE", {}).get(k, 0.0)
            w = per.get("WRITE_SIZE", {}).get(k, 0.0)
            rd, wr = 2 * f * 1024 / 1e6, w * 1024 / 1e6
            lines.append(f"| {k} | {f:.0f} | {w:.0f} | {rd:.2f} | {wr:.2f} | {rd + wr:.2f} |")
            # the iteration kernel: MODE_ITER_LIN (8) by default, MODE_ITER (0) with option linear = 0
            # (gl_persist_kernel: all iterations of a call in one launch; bench.py's per-launch
            # figures are per launch of whichever kernel ran)
            if "gl_persist_kernel" in k or ("gl_engine_kernel" in k and "gl_engine_iter_kernel" not in summary and
                                            k.rstrip(">").endswith(("8, false", "9, false", "0, false", "6, false", "4, false"))):
                summary["gl_engine_iter_kernel"] = k
                summary["gl_engine_iter_bytes_per_launch"] = int((rd + wr) * 1e6)
                summary["gl_engine_iter_read_bytes"] = int(rd * 1e6)
                summary["gl_engine_iter_write_bytes"] = int(wr * 1e6)
        lines.append("")
    with open(os.path.join(outdir, f"summary_{tag}.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(outdir, f"summary_{tag}.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
