#!/bin/bash
# usage (GPU box): bash tools/scene_prof.sh   -- per-kernel averages of the validity step on the compound-mesh scene c5m
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_c5m
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/variant_time.py numbotics_amd/csrc/libnbk.so c5m > "$out.log" 2>&1 < /dev/null
echo "rc $?"; grep thr "$out.log"
f=$(find "$out" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if "nbk" in n: print("   %-32s calls %3s avg %8.1f us" % (n[-32:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
fi
