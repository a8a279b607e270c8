#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table.

  python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/<name>.txt

Same columns as `rocprofv3 --stats` kernel_stats.csv (calls, total, average, min, max, percent)
plus the inter-kernel gap statistics that matter for the launch-bound recurrent chain."""
import sqlite3
import sys


def main(path, limit=40):
  cur = sqlite3.connect(path).cursor()
  rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                     "max(end-start) from kernels group by name order by 3 desc").fetchall()
  total = float(sum(r[2] for r in rows)) or 1.0
  print("# source: %s" % path)
  print("%-72s %7s %12s %10s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
  for r in rows[:limit]:
    print("%-72s %7d %12.1f %10.2f %9.2f %9.2f %6.2f" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3,
                                                        r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total))
  print("# all kernels: %d dispatches, %.1f us of kernel time" % (sum(r[1] for r in rows), total / 1e3))
  tl = cur.execute("select start, end from kernels order by start").fetchall()
  gaps = [max(0, b[0] - a[1]) for a, b in zip(tl, tl[1:])]
  if gaps:
    gaps_s = sorted(gaps)
    print("# inter-kernel gaps: median %.2f us, p90 %.2f us, sum %.1f us (wall %.1f us)"
          % (gaps_s[len(gaps_s) // 2] / 1e3, gaps_s[int(len(gaps_s) * 0.9)] / 1e3, sum(gaps) / 1e3,
             (tl[-1][1] - tl[0][0]) / 1e3))


if __name__ == "__main__":
  main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
