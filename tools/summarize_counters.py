#!/usr/bin/env python3
"""Per-kernel averages of the counters tools/profile_counters.sh collected (separate --pmc passes), with the ratios that say where a
kernel's wave cycles go: WAIT_ANY (parked on s_waitcnt / barrier), WAIT_INST_ANY (issue stalls), ACTIVE_INST_ANY (issuing) as fractions
of WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots), LDS bank-conflict cycles per LDS-active cycle, L2 hit rate."""
import collections, csv, glob, sys


def main(base, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(base + "/pmc*/*/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# rocprofv3 --pmc <group> -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline [args]; one pass per group;",
             "# averages over the last 30 launches of each kernel (SQ_* cycle counters are quad-cycles summed over all waves / SIMDs)"]
    want = [k for k in acc if any(s in k for s in ("k_anneal", "k_spawn", "k_diag", "k_psit", "rs_", "merge_path", "scan_lookback", "k_gate"))]
    for k in sorted(want, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0])[-30:])):
        a = {n: sum(v[-30:]) / max(1, len(v[-30:])) for n, v in acc[k].items()}
        lines.append("")
        lines.append(k)
        for n in sorted(a):
            lines.append("    %-26s %16.1f" % (n, a[n]))
        wc = a.get("SQ_WAVE_CYCLES")
        if wc:
            parts = ["%s/WAVE_CYCLES = %.3f" % (n[3:], a[n] / wc) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if n in a]
            lines.append("    -> " + "; ".join(parts))
            if a.get("SQ_WAVES"):
                lines.append("    -> wave cycles per wave (quad-cycles) = %.0f; VALU instructions per wave = %.0f" % (wc / a["SQ_WAVES"], a.get("SQ_INSTS_VALU", 0) / a["SQ_WAVES"]))
        if a.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("    -> LDS bank-conflict cycles / LDS-active cycles = %.3f" % (a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"]))
        if "TCC_HIT_sum" in a and a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0) > 0:
            lines.append("    -> L2 hit rate = %.3f" % (a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
