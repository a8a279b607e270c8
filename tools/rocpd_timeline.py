#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd database (kernel-trace): every dispatch of the step with
its start offset, duration and queue, the union of busy time and the idle gaps — to see what is on the critical
path and what runs beside it.

  python tools/rocpd_timeline.py <results.db> [anchor-substring] [step-index]

A step is delimited by consecutive dispatches of the anchor kernel (default: conv1_fwd)."""
import sqlite3
import sys


def main(path, anchor="conv1_fwd", which=-2):
  cur = sqlite3.connect(path).cursor()
  cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
  qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
  rows = cur.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
  marks = [i for i, r in enumerate(rows) if anchor in r[0]]
  if len(marks) < 3:
    print("anchor %r seen %d times" % (anchor, len(marks)))
    return
  a, b = marks[which], marks[which + 1]
  step = rows[a:b]
  t0 = step[0][1]
  print("# step of %d dispatches, wall %.1f us (anchor %s)" % (len(step), (rows[b][1] - t0) / 1e3, anchor))
  busy_end, busy, gaps = t0, 0.0, []
  for r in step:
    s, e = r[1], r[2]
    if s > busy_end:
      gaps.append((busy_end - t0, s - busy_end))
      busy += e - s
    else:
      busy += max(0, e - busy_end)
    busy_end = max(busy_end, e)
  print("# union of busy time %.1f us, idle inside the step %.1f us, sum of kernel time %.1f us"
        % (busy / 1e3, sum(g[1] for g in gaps) / 1e3, sum(r[2] - r[1] for r in step) / 1e3))
  print("%9s %9s %5s  %s" % ("start_us", "dur_us", "queue", "kernel"))
  for r in step:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%9.1f %9.1f %5s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", name[:90]))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "conv1_fwd", int(sys.argv[3]) if len(sys.argv) > 3 else -2)
