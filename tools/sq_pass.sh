#!/bin/bash
# usage (GPU box): bash tools/sq_pass.sh <tag> <script.py> [args...]  -- SQ counters per kernel of any tool script
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/${tag}_sq
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$out" -- python3 "$@" > "$out.log" 2>&1 < /dev/null
echo "rc $?"
f=$(find "$out" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
v = defaultdict(lambda: defaultdict(list)); dur = defaultdict(dict); grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "nbk::" not in k: continue
    v[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); grid[k] = int(r["Grid_Size"])
m = lambda x: sum(x) / len(x)
for k, c in v.items():
    waves = (grid[k] + 63) // 64; insts = m(c["SQ_INSTS_VALU"]); us = m(list(dur[k].values())) / 1e3; cyc = us * 1e-6 * 2.4e9
    print("%-34s %8.1f us  valu/wave %7.1f  lane_util %.2f  issue_util %.2f  waves/simd %.2f  salu/wave %6.1f lds/wave %5.1f" % (
        k[-34:], us, insts / waves, m(c["SQ_THREAD_CYCLES_VALU"]) / (insts * 64), insts * 4 / (1024 * cyc), m(c["SQ_WAVE_CYCLES"]) * 4 / (1024 * cyc),
        m(c["SQ_INSTS_SALU"]) / waves, m(c["SQ_INSTS_LDS"]) / waves))
PY
