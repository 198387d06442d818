#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ms, VGPR/LDS) from a rocprofv3 rocpd sqlite DB.
Usage: tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def summarize(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    extra = [x for x in ("arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "group_segment_size") if x in scol]
    q = f"""select s.{name_col}, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6,
            max(d.end-d.start)/1e6 {''.join(', max(s.'+e+')' for e in extra)}
            from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"""
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print(f"# {path}")
    print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_ms':>9} {'min_ms':>9} {'max_ms':>9} {'pct':>6}  " + " ".join(extra))
    for r in rows:
        nm = r[0] if len(r[0]) <= 70 else r[0][:67] + "..."
        print(f"{nm:<70} {r[1]:>6} {r[2]:>10.3f} {r[3]:>9.4f} {r[4]:>9.4f} {r[5]:>9.4f} {100*r[2]/total:>6.1f}  " +
              " ".join(str(x) for x in r[6:]))
    # PMC counters, if any
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    n = c.execute(f"select count(*) from {pm}").fetchone()[0]
    if n:
        pcols = [r[1] for r in c.execute(f"pragma table_info({pm})")]
        icols = [r[1] for r in c.execute(f"pragma table_info({pi})")]
        ev = "event_id" if "event_id" in pcols else "dispatch_id"
        q = f"""select s.{name_col}, i.name, count(*), sum(p.value), avg(p.value)
                from {pm} p join {pi} i on p.pmc_id = i.id join {kd} d on p.{ev} = d.{'event_id' if 'event_id' in cols else 'id'}
                join {ks} s on d.kernel_id = s.id group by 1, 2 order by 1, 2"""
        print("\n# counters (per-dispatch average)")
        for r in c.execute(q):
            nm = r[0] if len(r[0]) <= 60 else r[0][:57] + "..."
            print(f"{nm:<60} {r[1]:<28} n={r[2]:<5} avg={r[4]:.4g}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarize(p)
        print()
