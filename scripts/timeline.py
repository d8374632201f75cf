#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 --kernel-trace rocpd database: every dispatch between the ends of the last two
`marker` kernels (default dotm_kernel = the closing dot product of a quadrant-tree contraction): start offset, duration,
queue, name.   python scripts/timeline.py results.db [marker]"""
import re, sqlite3, sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "gemmk_dot_finish"
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda n: [x for x in tabs if x.startswith(n)][0]
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select s.display_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
ends = [r[2] for r in rows if marker in r[0]]
if len(ends) < 2:
    sys.exit("marker kernel not found twice")
lo, hi = ends[-2], ends[-1]
sel = [r for r in rows if lo <= r[1] < hi]
print(f"# one step: {(hi - lo) / 1e6:.3f} ms, {len(sel)} dispatches   (columns: start_us dur_us queue name)")
busy = 0
for name, s, e, q in sel:
    nm = re.sub(r"\(.*$", "", name).replace("void ", "").replace("qamd::", "").replace("qamdk::", "")[:60]
    print(f"{(s - lo) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  q{q}  {nm}")
# union of busy intervals
iv = sorted((r[1], r[2]) for r in sel)
cur_s, cur_e, tot = None, None, 0
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            tot += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    tot += cur_e - cur_s
print(f"# GPU busy (union of dispatch intervals): {tot / 1e6:.3f} ms of {(hi - lo) / 1e6:.3f} ms")
