#!/bin/bash
# usage (on the GPU box): bash tools/pmc_passes.sh <tag>
# kernel trace + the three PMC passes of the headline step, each its own run (counters never share a run with a trace domain other
# than --kernel-trace), outputs under gpurun_out/<tag>_*; copy the CSVs named below into profiles/ and run tools/reduce_pmc.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
CMD=${NBK_PMC_CMD:-"python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"}   # NBK_PMC_CMD="python3 tools/fk_time.py" profiles the FK kernels
run() {  # name, extra rocprofv3 args...
  name=$1; shift
  out=gpurun_out/${tag}_$name
  timeout -k 5 200 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out" -- $CMD > "$out.log" 2>&1 < /dev/null
  echo "$name rc $? $(find "$out" -name '*.csv' | wc -l) csv"
}
run stats --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU
mkdir -p gpurun_out/${tag}_collect
cp $(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_collect/${tag}_bench_kernel_stats.csv
for k in fetch write sq; do
  f=$(find gpurun_out/${tag}_$k -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" gpurun_out/${tag}_collect/${tag}_pmc_${k/fetch/fetch_size}_counter_collection.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nbk::" in r["Kernel_Name"]]
if rows:
    with open(sys.argv[2].replace("pmc_write_", "pmc_write_size_"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
done
ls -la gpurun_out/${tag}_collect
