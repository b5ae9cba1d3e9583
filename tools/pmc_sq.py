"""Sum of SQ counters per kernel from rocprofv3 --pmc result databases (one pass per counter group).
usage: python tools/pmc_sq.py <db> [<db> ...]   -> table kernel x counter (mean per dispatch of the largest grid)"""
import sqlite3, sys
from collections import defaultdict

tab = defaultdict(dict)
for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in names if "pmc_event" in t][0]
    kd = [t for t in names if "kernel_dispatch" in t][0]
    ks = [t for t in names if "kernel_symbol" in t][0]
    pi = [t for t in names if "info_pmc" in t][0]
    rows = con.execute(f"select s.kernel_name, d.grid_size_x, i.name, sum(e.value) from {ev} e join {kd} d on e.event_id = d.event_id "
                       f"join {ks} s on d.kernel_id = s.id join {pi} i on e.pmc_id = i.id group by d.id, i.name").fetchall()
    acc = defaultdict(list)
    for k, g, c, v in rows:
        acc[(k, c)].append((g, v))
    for (k, c), lst in acc.items():
        gmax = max(g for g, _ in lst)
        vals = [v for g, v in lst if g == gmax]
        tab[k][c] = sum(vals) / len(vals)
for k, d in tab.items():
    if "attn" not in k and "gemm" not in k:
        continue
    print(k[:80])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:16.0f}")
