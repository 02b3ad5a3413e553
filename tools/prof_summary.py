#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (or several) into a per-kernel table.

    python tools/prof_summary.py gpurun_out/prof_x/x_results.db > profiles/round1_x_kernel_stats.txt
"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    one = lambda pat: [r[0] for r in cur.execute(f"select name from sqlite_master where type='table' and name like '{pat}%'")][0]
    kd, ks = one("rocpd_kernel_dispatch"), one("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
         f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
    rows = list(cur.execute(q))
    total = sum(r[5] for r in rows) or 1
    print(f"# {path}")
    print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s} {'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds_B':>7s} {'grid_x':>9s} {'wg':>4s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]/1e3:10.2f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} {r[5]/1e6:9.3f} {100*r[5]/total:6.1f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:9d} {r[10]:4d}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
