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

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
