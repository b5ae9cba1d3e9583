"""Summarise a rocprofv3 --kernel-trace results database as the per-kernel table kept under profiles/.
usage: python tools/prof_summary.py <results.db> <steps_in_trace> <out_prefix> "<title line>" """
import sqlite3, sys, csv
db, steps, out, title = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
                   f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
# GPU-side idle between consecutive dispatches (gaps above 1 ms are host stalls: process start, the sync at the end): what the ~400 dependent
# launches of a step cost beyond their own durations
iv = sorted(con.execute(f"select start, end from {kd}").fetchall())
gap_ns, ngap, cur_end = 0, 0, None
for st, en in iv:
    if cur_end is not None and st > cur_end and st - cur_end < 1_000_000:
        gap_ns += st - cur_end
        ngap += 1
    cur_end = en if cur_end is None else max(cur_end, en)
with open(out + ".csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "total_ms", "ms_per_step", "avg_us", "min_us", "max_us", "pct"])
    for r in rows: w.writerow([r[0], r[1], round(r[2], 3), round(r[2] / steps, 3), round(r[3], 1), round(r[4], 1), round(r[5], 1), round(100 * r[2] / tot, 2)])
with open(out + ".md", "w") as f:
    f.write(f"# {title}\n\n{steps} steps in the trace; sum of kernel time {tot / steps:.2f} ms/step; idle between consecutive dispatches (gaps under 1 ms) {gap_ns / 1e6 / steps:.2f} ms/step over {ngap // steps} gaps/step, {len(iv) // steps} dispatches/step.\n\n| kernel | calls | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:24]: f.write(f"| `{r[0][:100]}` | {r[1]} | {r[2] / steps:.3f} | {r[3]:.1f} | {100 * r[2] / tot:.2f} |\n")
print(f"{out}.md / .csv written; {tot / steps:.2f} ms/step over {len(rows)} kernels; inter-dispatch idle {gap_ns / 1e6 / steps:.2f} ms/step ({len(iv) // steps} dispatches/step)")
