#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel count / total / avg /
min / max duration and, if present, per-kernel PMC counter sums and means.

    python scripts/rocpd_summary.py gpurun_out/prof/stats/r1_results.db [--top 15]
"""
import re
import sqlite3
import sys


def short(name):
    grid = ""
    if " grid=" in name:
        name, grid = name.rsplit(" grid=", 1)
        grid = " grid=" + grid
    name = re.sub(r"\(.*$", "", name) + grid
    name = name.replace("void ", "").replace("qamd::", "").replace("qamdk::", "")
    return name[:110]


def main(path, top=15):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda n: [x for x in tabs if x.startswith(n)][0]
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    by_grid = "--by-grid" in sys.argv
    gcol = " || ' grid=' || d.grid_size_x" if by_grid else ""
    rows = c.execute(
        f"select s.display_name{gcol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by 1 order by 3 desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    # the median next to the mean: a kernel's first launches (cold caches, the untimed tree probe and class pass of bench.py,
    # launches that share the chip with another stream's work) pull the mean; the timed region is the bulk of the calls
    durs = {}
    for name, d in c.execute(f"select s.display_name{gcol}, d.end-d.start from {kd} d join {ks} s on d.kernel_id = s.id"):
        durs.setdefault(name, []).append(d)
    med = {k: sorted(v)[len(v) // 2] for k, v in durs.items()}
    print(f"# {path}")
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, s, mn, mx in rows[:top]:
        print(f"{short(name):112s} {n:7d} {s/1e6:10.3f} {s/n/1e3:10.2f} {med[name]/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    print(f"{'TOTAL (all kernels)':112s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")
    # PMC
    try:
        pe, ip = t("rocpd_pmc_event"), t("rocpd_info_pmc")
        q = (
            f"select s.display_name{gcol}, p.name, count(*), sum(e.value), avg(e.value) from {pe} e "
            f"join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id "
            f"join {ks} s on d.kernel_id = s.id group by 1, p.name order by 4 desc"
        )
        prow = c.execute(q).fetchall()
        if prow:
            print("\n# PMC counters per kernel (sum over dispatches, mean per dispatch)")
            print(f"{'kernel':112s} {'counter':>12s} {'calls':>7s} {'sum':>16s} {'mean/dispatch':>16s}")
            for name, cn, n, s, a in prow[:top]:
                print(f"{short(name):112s} {cn:>12s} {n:7d} {s:16.1f} {a:16.1f}")
    except Exception as e:  # no counters collected
        print(f"\n# no PMC data ({e})")


if __name__ == "__main__":
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 15
    main(sys.argv[1], top)
