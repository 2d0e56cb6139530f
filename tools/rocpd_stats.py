#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats`
in ROCm 7.2) as a per-kernel table: calls, avg/min/max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--md profiles/r01_x.md] [--skip N]

--skip N drops the first N dispatches of every kernel (warm-up) before averaging.
"""
import argparse
import sqlite3


def kernel_stats(db, skip=0, by_grid=False):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                       "accum_vgpr_count, sgpr_count from kernels order by start").fetchall()
    per = {}
    for r in rows:
        key = r[0] if not by_grid else "%s  [grid %dx%dx%d]" % (r[0], r[3], r[4], r[5])
        per.setdefault(key, []).append(r)
    out = []
    for name, rs in per.items():
        rs = rs[skip:] if len(rs) > skip else rs
        d = [(r[2] - r[1]) / 1e3 for r in rs]
        out.append(dict(name=name, calls=len(d), avg_us=sum(d) / len(d), min_us=min(d), max_us=max(d),
                        total_us=sum(d), grid=(rs[-1][3], rs[-1][4], rs[-1][5]), wg=rs[-1][6], lds=rs[-1][7],
                        vgpr=rs[-1][8], agpr=rs[-1][9], sgpr=rs[-1][10]))
    out.sort(key=lambda r: -r["total_us"])
    tot = sum(r["total_us"] for r in out) or 1.0
    for r in out:
        r["pct"] = 100.0 * r["total_us"] / tot
    return out


def to_markdown(stats, title):
    lines = ["# %s" % title, "",
             "| kernel | calls | avg µs | min µs | max µs | % GPU time | grid | wg | LDS B | VGPR | AGPR | SGPR |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in stats:
        lines.append("| `%s` | %d | %.2f | %.2f | %.2f | %.1f | %s | %d | %d | %d | %d | %d |" % (
            r["name"][:150], r["calls"], r["avg_us"], r["min_us"], r["max_us"], r["pct"],
            "x".join(str(g) for g in r["grid"]), r["wg"], r["lds"], r["vgpr"], r["agpr"], r["sgpr"]))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--md")
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--title", default=None)
    ap.add_argument("--by-grid", action="store_true", help="separate rows per launch grid (problem size)")
    a = ap.parse_args()
    st = kernel_stats(a.db, a.skip, a.by_grid)
    md = to_markdown(st, a.title or ("rocprofv3 --kernel-trace --stats summary of %s" % a.db))
    if a.md:
        open(a.md, "w").write(md)
    print(md)
