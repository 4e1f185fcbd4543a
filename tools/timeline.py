#!/usr/bin/env python3
"""Prints one MC step's kernel timeline (start, duration, gap to the previous kernel's end) from a
rocprofv3 --kernel-trace rocpd database: shows where the GPU waits for the host."""
import sqlite3, glob, sys

def main(base, which=-30):
    db = glob.glob(base + '/trace/*/*.db')[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
    anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_gate'        # a kernel that runs once per step
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    i0, i1 = idx[which], idx[which + 1]
    t0, prev_end = rows[i0][1], None
    for r in rows[i0:i1 + 1]:
        nm = r[0].replace('_Z', '').lstrip('0123456789')[:26]
        print("%-28s q%s start %8.2f dur %6.2f  gap %6.2f" % (nm, r[3], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, ((r[1] - prev_end) / 1e3 if prev_end else 0)))
        prev_end = r[2]
    steps = [(rows[idx[k + 1]][1] - rows[idx[k]][1]) / 1e3 for k in range(len(idx) - 60, len(idx) - 1)]
    print("mean step (gate to gate) over the last %d steps: %.1f us" % (len(steps), sum(steps) / len(steps)))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -30)
