"""Per-kernel summary (count, average / total duration) and inter-kernel gaps from a rocprofv3
rocpd database; optional per-grid-size breakdown.  usage: rocpd_summary.py results.db [--by-grid KERNEL_SUBSTR]"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda s: [x for x in tabs if x.startswith(s)][0]
kd, ks = T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = 'display_name' if 'display_name' in cols else 'kernel_name'
names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from {ks}")}
rows = cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd} order by start").fetchall()
agg = {}
for k, s, e, g, w in rows:
    a = agg.setdefault(names.get(k, str(k)), [])
    a.append(e - s)
print(f"{'kernel':70s} {'calls':>8s} {'avg_us':>10s} {'total_ms':>10s}")
for n, a in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    a = np.array(a)
    print(f"{n[:70]:70s} {len(a):8d} {a.mean()/1e3:10.2f} {a.sum()/1e6:10.2f}")
st = np.array([r[1] for r in rows]); en = np.array([r[2] for r in rows])
gaps = st[1:] - en[:-1]
gaps = gaps[(gaps > -1e6) & (gaps < 1e5)]
print(f"gaps between consecutive dispatches (<100us): n={len(gaps)} median={np.median(gaps)/1e3:.2f} us mean={gaps.mean()/1e3:.2f} us p90={np.percentile(gaps,90)/1e3:.2f} us")
if len(sys.argv) > 3 and sys.argv[2] == '--by-grid':
    sub = sys.argv[3]
    sel = [(g // w, e - s) for k, s, e, g, w in rows if sub in names.get(k, '')]
    sel = sel[len(sel) * 2 // 3:]  # last repetition
    bins = {}
    for wg, dur in sel:
        bins.setdefault(wg // 256 * 256, []).append(dur)
    for b in sorted(bins):
        a = np.array(bins[b]); print(f"  workgroups {b:5d}+: n={len(a):5d} avg={a.mean()/1e3:8.2f} us  min={a.min()/1e3:8.2f}")
