"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately with --kernel-trace):
bytes = FETCH_SIZE * 1024 * 2 (gfx950 counts wide streaming reads at half their size, MI355X_MICROARCH.md HBM section)
      + WRITE_SIZE * 1024, per dispatch; only the encoder-sized launches of each kernel (the largest grid) are kept.
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <out.json>"""
import json, sqlite3, sys
from collections import defaultdict


def per_kernel(db):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if "pmc_event" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = con.execute(f"select s.kernel_name, d.grid_size_x, sum(e.value) from {ev} e join {kd} d on e.event_id = d.event_id "
                       f"join {ks} s on d.kernel_id = s.id group by d.id").fetchall()
    out = defaultdict(list)
    for name, grid, val in rows:
        out[name].append((grid, val))
    return out


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
res = {}
for name in fetch:
    gmax = max(g for g, _ in fetch[name])
    f = [v for g, v in fetch[name] if g == gmax]
    w = [v for g, v in write.get(name, []) if g == gmax]
    if not w:
        continue
    res[name] = {"launches": len(f), "read_bytes": int(sum(f) / len(f) * 1024 * 2), "write_bytes": int(sum(w) / len(w) * 1024)}
bwd = [k for k in res if "attn_bwd_dkdvw_asm" in k] + [k for k in res if "attn_bwd_dq_asm" in k]    # the kernels the benchmark step's encoder layers run
if len(bwd) != 2:
    bwd = [k for k in res if "attn_bwd_dkdv_asm" in k] + [k for k in res if "attn_bwd_dq_asm" in k]
if len(bwd) != 2:
    bwd = [k for k in res if "attn_bwd_dkdv_ps" in k] + [k for k in res if "attn_bwd_dq_ps" in k]
if not bwd:
    bwd = [k for k in res if "attn_bwd_dkdv" in k and "DF16bLi64" in k] + [k for k in res if "attn_bwd_dq" in k and "DF16bLi64" in k]
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py --steps 1 "
                 "--warmup 1 --no-cpu-baseline --no-timers --no-parity --no-other-configs`; bytes = FETCH_SIZE*1024*2 (gfx950 wide-stream correction, "
                 "MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024; mean over the encoder-sized launches of each kernel",
       "fk_attn_bwd_bytes_per_call": int(sum(res[k]["read_bytes"] + res[k]["write_bytes"] for k in bwd)),
       "kernels": res}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print("fk_attn_bwd bytes/call", doc["fk_attn_bwd_bytes_per_call"])
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))[:12]:
    print(f"{(v['read_bytes'] + v['write_bytes']) / 1e6:10.1f} MB  r {v['read_bytes'] / 1e6:9.1f}  w {v['write_bytes'] / 1e6:9.1f}  x{v['launches']:3d}  {k[:90]}")
