#!/usr/bin/env python
"""Turn a rocprofv3 results .db (--kernel-trace --stats, optionally --pmc) into the small text summary that
is committed under profiles/.   usage: prof_summary.py <results.db> <out.txt> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    lines = [f"# {title}", "# source: rocprofv3 --kernel-trace --stats (view top_kernels); durations in milliseconds", ""]
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines.append(f"{'calls':>6} {'total_ms':>14} {'avg_ms':>12} {'pct':>7}  kernel")
    for name, calls, tot, avg, pct in rows[:24]:
        short = name.split("(")[0].replace("void ", "")
        lines.append(f"{calls:>6} {tot/1e3:>14.1f} {avg/1e3:>12.2f} {pct:>7.3f}  {short}")
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "pmc_events" in tabs or any(t.startswith("rocpd_pmc_event") for t in tabs):
        try:
            q = ("select k.kernel_name, p.counter_name, count(*), sum(p.value), avg(p.value) from counters_collection p "
                 "join kernels k on 1=0")
        except Exception:
            pass
        for view in ("counters_collection",):
            if view in tabs:
                cur = c.execute(f"select * from {view} limit 1")
                cols = [d[0] for d in cur.description]
                lines += ["", f"# PMC counters (view {view}; columns {cols})"]
                namecol = "kernel_name" if "kernel_name" in cols else cols[0]
                agg = c.execute(f"select {namecol}, counter_name, count(*), sum(value), avg(value) from {view} "
                                f"group by {namecol}, counter_name order by sum(value) desc").fetchall()
                lines.append(f"{'n':>6} {'sum':>20} {'avg_per_dispatch':>20}  counter  kernel")
                for kn, cn, n, s, a in agg[:60]:
                    lines.append(f"{n:>6} {s:>20.4g} {a:>20.6g}  {cn}  {str(kn).split('(')[0]}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
