#!/bin/bash
# usage: tools/variant_prof.sh tag lib1.so lib2.so ...   (run on the GPU box): per-kernel averages of one validity step per build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for lib in "$@"; do
  name=$(basename "$lib" .so)
  out=gpurun_out/${tag}_${name}
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/variant_time.py "$lib" > "$out.log" 2>&1 < /dev/null
  echo "== $name rc $? $(grep 'thr 0:' "$out.log" | head -1)"
  f=$(find "$out" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if "k_broad" in n or "k_narrow" in n:
        print("   %-28s calls %3s avg %8.1f us  min %8.1f  max %8.1f" % (n[-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  else echo "   (no kernel_stats.csv)"; fi
done
