#!/bin/bash
# usage (GPU box): bash tools/sq_more.sh <tag> <lib.so> [scene]   -- two SQ counter passes (8 counters each) over tools/variant_time.py
# with that build of the library; prints per-kernel figures normalised per wave.  Counters only ride with --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; lib=$2; scene=${3:-c2}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  out=gpurun_out/${tag}_sqm$i
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$out" -- python3 tools/variant_time.py "$lib" "$scene" > "$out.log" 2>&1 < /dev/null
  echo "pass $i rc $?"
done
python3 - gpurun_out/${tag}_sqm1 gpurun_out/${tag}_sqm2 <<'PY'
import csv, sys, glob, os
from collections import defaultdict
v = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list); grid = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "nbk::" not in k: continue
            v[k][r["Counter_Name"]].append(float(r["Counter_Value"])); grid[k] = int(r["Grid_Size"])
            if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = lambda x: sum(x) / max(1, len(x))
for k, c in v.items():
    waves = (grid[k] + 63) // 64; us = m(dur[k]) / 1e3
    g = lambda n: m(c[n]) if n in c else float("nan")
    wc = g("SQ_WAVE_CYCLES")
    print("%s: %.1f us, %d waves" % (k[-40:], us, waves))
    print("   per wave: valu %.0f salu %.0f smem %.0f branch %.0f ifetch %.0f | wave lifetime %.0f quad-cycles: wait_any %.2f wait_inst_any %.2f active_any %.2f active_valu %.2f active_sca %.2f | salu cycles/wave %.0f | mean outstanding smem %.2f vmem %.2f per wave-cycle"
          % (g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_SALU") / waves, g("SQ_INSTS_SMEM") / waves, g("SQ_INSTS_BRANCH") / waves, g("SQ_IFETCH") / waves,
             wc / waves, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_SCA") / wc,
             g("SQ_INST_CYCLES_SALU") / waves, g("SQ_INST_LEVEL_SMEM") / wc, g("SQ_INST_LEVEL_VMEM") / wc))
PY
