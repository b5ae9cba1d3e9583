"""MFMA-pipe busy fraction and executed MFMA FLOP/s per kernel from one rocprofv3 counter pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU --kernel-trace).
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * 1024 SIMDs) per dispatch (the gfx94x derived-counter formula);
MFMA FLOP = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512.
usage: python tools/pmc_mfma_util.py <results.db> <steps_in_trace> <out.json>"""
import json, sqlite3, sys
from collections import defaultdict

db, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
ev = [t for t in names if "pmc_event" in t][0]
kd = [t for t in names if "kernel_dispatch" in t][0]
ks = [t for t in names if "kernel_symbol" in t][0]
pi = [t for t in names if "info_pmc" in t][0]
rows = con.execute(f"select d.id, s.kernel_name, d.end - d.start, i.name, sum(e.value), max(e.value) from {ev} e "
                   f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {pi} i on e.pmc_id = i.id "
                   f"group by d.id, i.name").fetchall()
disp = defaultdict(dict)
for did, k, dur, c, sm, mx in rows:
    disp[did]["name"], disp[did]["ns"] = k, dur
    disp[did][c] = (sm, mx)
per = defaultdict(lambda: dict(launches=0, ns=0, busy=0.0, gui=0.0, mops=0.0))
tot = dict(ns=0, busy=0.0, gui=0.0, mops=0.0)
for d in disp.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    busy, gui = d["SQ_VALU_MFMA_BUSY_CYCLES"][0], d["GRBM_GUI_ACTIVE"][1]
    mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", (0, 0))[0]
    p = per[d["name"]]
    p["launches"] += 1; p["ns"] += d["ns"]; p["busy"] += busy; p["gui"] += gui; p["mops"] += mops
    tot["ns"] += d["ns"]; tot["busy"] += busy; tot["gui"] += gui; tot["mops"] += mops
doc = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU "
                 "--kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-timers --no-parity --no-other-configs; MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / "
                 "(max(GRBM_GUI_ACTIVE) * 1024 SIMDs) per dispatch (the gfx94x derived-counter formula); MFMA TF/s = "
                 "SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / duration  (tools/pmc_mfma_util.py)",
       "whole_trace": {"mfma_util_pct": 100 * tot["busy"] / (tot["gui"] * 1024), "mfma_tflops": tot["mops"] * 512 / tot["ns"] / 1e3,
                       "avg_clock_ghz": tot["gui"] / tot["ns"], "mfma_flop_per_step": tot["mops"] * 512 / steps},
       "kernels": {k: {"launches": p["launches"], "ms": p["ns"] / 1e6, "clock_ghz": p["gui"] / p["ns"],
                       "mfma_util_pct": 100 * p["busy"] / (p["gui"] * 1024), "mfma_tflops": p["mops"] * 512 / p["ns"] / 1e3}
                   for k, p in sorted(per.items(), key=lambda kv: -kv[1]["ns"]) if p["mops"] > 0}}
json.dump(doc, open(out, "w"), indent=1)
print("whole trace: MfmaUtil %.1f %%, executed %.0f TF/s" % (doc["whole_trace"]["mfma_util_pct"], doc["whole_trace"]["mfma_tflops"]))
for k, v in list(doc["kernels"].items())[:8]:
    print("%6.1f %%  %7.0f TF/s  %8.2f ms  %s" % (v["mfma_util_pct"], v["mfma_tflops"], v["ms"], k[:80]))
