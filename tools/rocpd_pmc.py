#!/usr/bin/env python
"""Per-kernel averages of one PMC counter from a rocprofv3 rocpd database.

  python tools/rocpd_pmc.py <results.db> [kernel-substring ...]

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (128-B requests
tallied at 64 B), so the printed `fetch_corrected_KiB` doubles it; WRITE_SIZE is uncalibrated and
printed as is.  Both come from the L2's memory-side counters, so Infinity-Cache hits are counted."""
import sqlite3
import sys


def main(path, filters):
  cur = sqlite3.connect(path).cursor()
  rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
                     "sum(value) from counters_collection group by kernel_name, counter_name "
                     "order by 7 desc").fetchall()
  print("# source: %s" % path)
  print("%-64s %-11s %6s %12s %12s %12s" % ("kernel", "counter", "calls", "avg_KiB", "min_KiB", "max_KiB"))
  for name, cname, n, avg, mn, mx, _ in rows:
    if filters and not any(f in name for f in filters):
      continue
    extra = "   fetch_corrected_KiB=%.1f" % (2 * avg) if cname == "FETCH_SIZE" else ""
    print("%-64s %-11s %6d %12.1f %12.1f %12.1f%s" % (name[:64], cname, n, avg, mn, mx, extra))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2:])
