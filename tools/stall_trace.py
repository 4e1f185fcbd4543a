#!/usr/bin/env python3
"""Where does the one long step of a fresh process go?  (one step 40-75 ms after a process starts to load the GPU
takes 60-80 ms.)  Reads a rocprofv3 run of bench.py taken with --kernel-trace --hip-trace --output-format csv and prints
  * the longest gaps between consecutive kernel dispatches (end of one to start of the next) with the kernels around them,
  * the longest kernels,
  * the host API calls (hipLaunchKernel, hipEventRecord, hipStreamWaitEvent, ...) that took longer than a millisecond, with what ran
    on the GPU meanwhile.
A gap with a long host call inside it is the host's (driver) stall; a long kernel is the device's.
usage: python tools/stall_trace.py <rocprofv3 output dir>"""
import csv, glob, sys


def main(base):
    kf = glob.glob(base + "/**/*kernel_trace.csv", recursive=True)
    hf = glob.glob(base + "/**/*hip_api_trace.csv", recursive=True)
    rows = sorted(({"name": r["Kernel_Name"].split("(")[0][:40], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(kf[0]))), key=lambda r: r["s"])
    t0 = rows[0]["s"]
    print("%d kernel dispatches over %.1f ms; t = 0 at the first one" % (len(rows), (rows[-1]["e"] - t0) / 1e6))
    gaps = sorted(((rows[i + 1]["s"] - max(r["e"] for r in rows[max(0, i - 8):i + 1]), i) for i in range(len(rows) - 1)), reverse=True)[:5]
    print("\nlongest gaps with no kernel running:")
    for g, i in gaps:
        print("  %9.3f ms at t = %9.3f ms   after %-28s before %-28s" % (g / 1e6, (rows[i]["e"] - t0) / 1e6, rows[i]["name"], rows[i + 1]["name"]))
    print("\nlongest kernels:")
    for r in sorted(rows, key=lambda r: r["s"] - r["e"])[:5]:
        print("  %9.3f ms at t = %9.3f ms   %s" % ((r["e"] - r["s"]) / 1e6, (r["s"] - t0) / 1e6, r["name"]))
    if hf:
        api = [{"f": r["Function"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(hf[0]))]
        slow = sorted((a for a in api if a["e"] - a["s"] > 1e6 and a["s"] > t0), key=lambda a: a["s"] - a["e"])[:12]
        print("\nhost API calls longer than 1 ms after the first kernel (%d calls traced):" % len(api))
        for a in sorted(slow, key=lambda a: a["s"]):
            busy = [r["name"] for r in rows if r["s"] < a["e"] and r["e"] > a["s"]]
            print("  %9.3f ms at t = %9.3f ms   %-28s kernels running meanwhile: %d %s" % ((a["e"] - a["s"]) / 1e6, (a["s"] - t0) / 1e6, a["f"], len(busy), sorted(set(busy))[:3]))


if __name__ == "__main__":
    main(sys.argv[1])
