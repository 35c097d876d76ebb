#!/bin/bash
# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of the headline step only (bench.py --no-extras); prints the per-kernel table of nbk kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/${tag}_stats
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$out.log" 2>&1 < /dev/null
echo "rc $?"
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "nbk::" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us  min {float(r["MinNs"])/1e3:9.2f}  max {float(r["MaxNs"])/1e3:9.2f}')
PY
grep "^{" "$out.log" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], d['parity_vs_oracle'])
print({k:(v['narrowphase_build'], round(v['kernel_ms_median'],4), round(v['kernel_ms_min'],4), round(v['kernel_ms_max'],4)) for k,v in d['modes'].items() if isinstance(v,dict)})
"
