#!/bin/bash
# per-kernel times of tools/attn_bench.py under rocprofv3 --kernel-trace for library variants: tools/kt_attn.sh name1 name2 ...  ("main" = in-tree)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$ROOT/frankenstein_amd/variants/lib_$v.so; fi
  rm -rf /tmp/kt_$v; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_$v -- python3 $ROOT/tools/attn_bench.py 3 > /tmp/kt_$v.log 2>&1)
  echo "== $v"
  python3 - "$(find /tmp/kt_$v -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
for name, n, avg, mn in con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%attn%' group by s.kernel_name order by 3 desc"):
    print(f"  {name[:70]:70s} n={n:3d} avg {avg:8.1f} us  min {mn:8.1f} us")
PY
done
