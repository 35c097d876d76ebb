#!/bin/bash
# usage (GPU box): bash tools/thr_prof.sh <scene> [sharp]  -- rocprofv3 kernel trace of tools/thr_time.py: per-kernel averages over the four thresholds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_thr_$1$2
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/thr_time.py "$@" > "$out.log" 2>&1 < /dev/null
echo "rc $?"; grep thr "$out.log"
f=$(find "$out" -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if "nbk" in n: print("   %-36s calls %3s avg %9.1f us  min %9.1f  max %9.1f" % (n[-36:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
