#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one db per --pmc pass).

    python tools/rocpd_pmc.py gpurun_out/pmc_x_sq/sq_results.db gpurun_out/pmc_x_fetch/fetch_results.db ... [--md out.md]

Counter values are summed over the dimension instances of a dispatch (XCC/SE/...), then averaged over the
dispatches of a kernel (first --skip dispatches dropped).
"""
import argparse
import sqlite3
from collections import defaultdict


def collect(db, skip):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, dispatch_id, counter_name, sum(value), min(end - start), grid_size "
                       "from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
    per = defaultdict(lambda: defaultdict(dict))      # kernel -> counter -> dispatch -> value
    dur = defaultdict(dict)
    for k, d, c, v, t, g in rows:
        key = "%s [grid %d]" % (k.split("(")[0], g)
        per[key][c][d] = v
        dur[key][d] = t / 1e3
    out = {}
    for k, counters in per.items():
        ds = sorted(dur[k])[skip:] or sorted(dur[k])
        o = {"dispatches": len(ds), "avg_us_profiled": sum(dur[k][d] for d in ds) / len(ds)}
        for c, vals in counters.items():
            o[c] = sum(vals[d] for d in ds if d in vals) / len(ds)
        out[k] = o
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--skip", type=int, default=5)
    ap.add_argument("--md")
    ap.add_argument("--filter", default="smot::")
    a = ap.parse_args()
    merged = defaultdict(dict)
    for db in a.dbs:
        for k, o in collect(db, a.skip).items():
            for c, v in o.items():
                merged[k].setdefault(c, v)
    lines = []
    for k in sorted(merged):
        if a.filter and a.filter not in k:
            continue
        lines.append("### `%s`" % k)
        for c in sorted(merged[k]):
            lines.append("- %s: %.6g" % (c, merged[k][c]))
        lines.append("")
    text = "\n".join(lines)
    if a.md:
        open(a.md, "w").write(text)
    print(text)
