#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2) into the
plain-text per-kernel summary kept under profiles/.   usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    lines = ["# rocprofv3 --kernel-trace --stats  (durations in microseconds)", "# source: " + db,
             "%-70s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows:
        lines.append("%-70s %8d %14.3f %14.3f %8.3f" % (name[:70], calls, tot, avg, pct))
    try:
        k = c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, "
                      "min(duration), max(duration), count(*) from kernels where name like '%t2gpu%' or name like '%front_%' or name like '%p1_%' or name like '%cp_correlate%' group by name").fetchall()
        lines.append("")
        lines.append("# library kernels: grid, workgroup, dynamic LDS, VGPR, AGPR, SGPR, scratch, min/max duration (ns), launches")
        for r in k:
            lines.append("  " + " | ".join(str(x) for x in r))
    except sqlite3.Error as e:
        lines.append("# (kernel detail unavailable: %s)" % e)
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
