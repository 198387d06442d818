#!/usr/bin/env python3
"""How many kernels run at the same time: from a rocprofv3 --kernel-trace rocpd database, the time-weighted histogram of the
number of kernels in flight, per-queue dispatch counts, and the busiest stretch.   tools/trace/concurrency.py <results.db> [skip_fraction]"""
import sqlite3
import sys


def main(path, skip=0.5):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else None
    rows = list(c.execute(f"select start, end{', ' + qcol if qcol else ''} from {kd} order by start"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t0 + (t1 - t0) * skip          # look at the last part of the run (the timed passes), not at warm-up
    ev = []
    queues = {}
    for r in rows:
        if r[1] <= cut:
            continue
        ev.append((max(r[0], cut), 1))
        ev.append((r[1], -1))
        if qcol:
            queues[r[2]] = queues.get(r[2], 0) + 1
    ev.sort()
    hist = {}
    cur, last = 0, ev[0][0]
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    total = sum(hist.values())
    print(f"# {path}: {len(ev) // 2} dispatches in the last {100 * (1 - skip):.0f} % of the run, {total / 1e6:.3f} ms")
    for k in sorted(hist):
        print(f"in flight {k:2d}: {hist[k] / 1e6:9.3f} ms  {100 * hist[k] / total:5.1f} %")
    print("mean kernels in flight while any runs:", round(sum(k * v for k, v in hist.items()) / max(1, sum(v for k, v in hist.items() if k)), 2))
    if qcol:
        print("dispatches per queue:", dict(sorted(queues.items())))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
