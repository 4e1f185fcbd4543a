#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/...) into the small summaries kept under
profiles/: per-kernel time stats and, from separate --pmc passes, FETCH_SIZE / WRITE_SIZE per
launch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of
wide coalesced reads: doubled; units KiB)."""
import csv, collections, glob, sys

def main(base, out):
    lines = []
    ks = glob.glob(base + '/trace/*/*kernel_stats.csv')
    if ks:
        lines.append("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline")
        lines.append("%-64s %8s %12s %10s" % ("kernel", "calls", "avg_us", "percent"))
        for r in list(csv.DictReader(open(ks[0])))[:30]:
            lines.append("%-64s %8s %12.2f %10s" % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
    def pmc(pattern, counter):
        acc = collections.defaultdict(list)
        for f in glob.glob(pattern):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == counter:
                    acc[r['Kernel_Name'][:48]].append(float(r['Counter_Value']))
        return acc
    f = pmc(base + '/fetch/*/*counter_collection.csv', 'FETCH_SIZE')
    w = pmc(base + '/write/*/*counter_collection.csv', 'WRITE_SIZE')
    if f:
        lines.append("")
        lines.append("# separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline")
        lines.append("# averages over the last 30 launches of each kernel; HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024")
        lines.append("%-50s %14s %14s %16s" % ("kernel", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes/launch"))
        for k in sorted(f, key=lambda k: -sum(f[k]))[:16]:
            fl, wl = f[k][-30:], w.get(k, [0])[-30:]
            fa, wa = sum(fl) / len(fl), sum(wl) / len(wl)
            lines.append("%-50s %14.1f %14.1f %16.4e" % (k, fa, wa, (2 * fa + wa) * 1024))
    def table(sub, title):
        fs = glob.glob(base + '/%s/*/*counter_collection.csv' % sub)
        if not fs:
            return
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
        names = sorted({c for k in acc for c in acc[k]})
        lines.append("")
        lines.append(title)
        lines.append("%-50s" % "kernel" + "".join("%21s" % n for n in names))
        for k in sorted(acc, key=lambda k: -sum(acc[k][names[0]][-30:]))[:16]:
            lines.append("%-50s" % k + "".join("%21.2f" % (sum(acc[k][n][-30:]) / max(1, len(acc[k][n][-30:]))) for n in names))
    table('occ', "# separate pass: rocprofv3 --pmc OccupancyPercent MeanOccupancyPerCU (averages over the last 30 launches; % of the chip's wave slots, waves per CU)")
    table('sq', "# separate pass: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (per launch)")
    open(out, 'w').write("\n".join(lines) + "\n")
    print("\n".join(lines))
    traffic_json(base, out, f, w)


def traffic_json(base, out, f, w):
    """profiles/*_traffic.json: what bench.py prints as roofline.traffic.  FETCH_SIZE is calibrated on this box, in this
    run, with tools/calib_fetch.hip (known byte counts in the step's own access patterns: 8-byte coalesced streams and
    32-byte gathered records), as MI355X_MICROARCH.md asks for access widths other than 16 B per lane."""
    import json, os, re
    cal = {}
    known = {}
    log = base + ".calib.log"
    if os.path.exists(log):
        for m in re.finditer(r"known_bytes (k_\w+) (\d+)(?: \(\+(\d+) index\))?", open(log).read()):
            known[m.group(1)] = int(m.group(2)) + int(m.group(3) or 0)
    cf = collections.defaultdict(list)
    for fn in glob.glob(base + '/calib/*/*counter_collection.csv'):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == 'FETCH_SIZE':
                cf[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in cf.items():
        if k in known and v:
            fetch = sum(v[1:]) / max(1, len(v[1:])) * 1024.0          # first launch warms the TLBs
            cal[k] = {"known_bytes": known[k], "fetch_size_bytes": fetch, "bytes_per_fetch_byte": known[k] / fetch if fetch else None}
    bench = {}
    for lg in (base + ".fetch.log", base + ".trace.log"):
        if os.path.exists(lg):
            for ln in open(lg):
                if ln.startswith('{"metric"'):
                    bench = json.loads(ln); break
        if bench:
            break
    n_avg = bench.get("config", {}).get("occupied_dets_per_step"); s_avg = bench.get("config", {}).get("spawns_per_step")
    f_s = (cal.get("k_stream8") or {}).get("bytes_per_fetch_byte") or 2.0
    f_g = (cal.get("k_gather32") or {}).get("bytes_per_fetch_byte") or 2.0
    kernels = {}
    for name, key in (("k_anneal", "k_anneal"), ("k_spawn", "k_spawn"), ("k_diag", "k_diag")):
        ks = [k for k in f if key in k]
        if not ks:
            continue
        k = max(ks, key=lambda q: sum(f[q]))
        fl, wl = f[k][-30:], w.get(k, [0])[-30:]
        fa, wa = sum(fl) / len(fl) * 1024.0, sum(wl) / len(wl) * 1024.0
        if name == "k_anneal" and n_avg:
            a_s, a_g = (8 + 51) * n_avg + 8 * s_avg, 32 * s_avg           # streamed: sorted words + resident walkers; gathered: spawn records
            f_eff = (a_s + a_g) / (a_s / f_s + a_g / f_g)
        else:
            f_eff = f_s
        kernels[name] = {"kernel": k, "fetch_size_bytes": fa, "write_size_bytes": wa, "hbm_bytes_x1": fa + wa, "hbm_bytes_x2": 2 * fa + wa,
                         "fetch_factor_used": f_eff, "hbm_bytes_calibrated": f_eff * fa + wa}
    doc = {"commit": os.environ.get("SQMC_COMMIT", "unknown"), "bench_line": {k: bench.get(k) for k in ("value", "ms_per_step", "steps")},
           "occupied_dets_per_step": n_avg, "spawns_per_step": s_avg, "calibration": cal, "kernels": kernels,
           "note": "FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes, averages over the last 30 launches; calibration factors = known bytes / FETCH_SIZE bytes of tools/calib_fetch.hip in the same session"}
    jp = out.replace("_rocprof_summary.txt", "_traffic.json")
    json.dump(doc, open(jp, "w"), indent=1)
    print("wrote", jp)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
