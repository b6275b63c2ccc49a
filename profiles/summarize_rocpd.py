#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite outputs into the small CSV summaries kept in this directory.

    python profiles/summarize_rocpd.py stats  <results.db> <out.csv>     # --kernel-trace --stats run
    python profiles/summarize_rocpd.py pmc    <results.db> <out.csv>     # --pmc run (one counter set)
"""
import csv
import sqlite3
import sys


def kernel_stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
                               max(d.end - d.start)
                        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{100.0 * r[2] / total:.2f}"])


def pmc_stats(db, out):
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info(rocpd_pmc_event)")]
    rows = c.execute("""select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value)
                        from rocpd_pmc_event e
                        join rocpd_info_pmc p on e.pmc_id = p.id
                        join rocpd_kernel_dispatch d on e.event_id = d.event_id
                        join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                        group by s.kernel_name, p.name order by 5 desc""").fetchall()
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Counter", "Dispatches", "AveragePerDispatch", "Total"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}"])
    return cols


if __name__ == "__main__":
    mode, db, out = sys.argv[1:4]
    (kernel_stats if mode == "stats" else pmc_stats)(db, out)
