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
    # the 24 largest rows of any origin, then EVERY remaining kernel of this library (small ones included: the CLIP / VGG / wgrad
    # kernels are < 1 % of a step and used to fall off the list)
    keep = [r for i, r in enumerate(rows) if i < 24 or "nerfart" in r[0]]
    for name, calls, tot, avg, pct in keep:
        short = name.split("(")[0].replace("void ", "")
        lines.append(f"{calls:>6} {tot/1e3:>14.1f} {avg/1e3:>12.2f} {pct:>7.3f}  {short}")
    other = [r for i, r in enumerate(rows) if not (i < 24 or "nerfart" in r[0])]
    if other:
        lines.append(f"{sum(r[1] for r in other):>6} {sum(r[2] for r in other)/1e3:>14.1f} {'':>12} {sum(r[4] for r in other):>7.3f}  ({len(other)} other kernels, none of this library)")
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc_views = [t for t in tabs if t in ("counters_collection", "pmc_events", "pmc_info") or "pmc" in t.lower() or "counter" in t.lower()]
    lines += ["", f"# pmc-related tables/views: {pmc_views}"]
    for view in ("counters_collection",):
        if view in tabs:
            cur = c.execute(f"select * from {view} limit 1")
            cols = [d[0] for d in cur.description]
            lines += [f"# view {view} columns: {cols}"]
            namecol = next((x for x in ("kernel_name", "name", "kernel") if x in cols), None)
            cntcol = next((x for x in ("counter_name", "pmc_name", "counter") if x in cols), None)
            valcol = next((x for x in ("value", "counter_value") if x in cols), None)
            if namecol and cntcol and valcol:
                agg = c.execute(f"select {namecol}, {cntcol}, count(*), sum({valcol}), avg({valcol}) from {view} "
                                f"group by {namecol}, {cntcol} order by {cntcol}, sum({valcol}) desc").fetchall()
                lines.append(f"{'dispatches':>10} {'sum':>20} {'avg_per_dispatch':>20}  counter  kernel")
                for kn, cn, n, sm, av in agg:
                    if "nerfart" in str(kn):
                        lines.append(f"{n:>10} {sm:>20.6g} {av:>20.6g}  {cn}  {str(kn).split('(')[0].replace('void ', '')}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
