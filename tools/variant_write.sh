#!/bin/bash
# usage: tools/variant_write.sh tag lib1.so lib2.so ...   (GPU box): WRITE_SIZE / FETCH_SIZE per launch and the average duration of the validity kernels of each build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for lib in "$@"; do
  name=$(basename "$lib" .so)
  for c in WRITE_SIZE FETCH_SIZE; do
    out=gpurun_out/${tag}_${name}_$c
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out" -- python3 tools/variant_time.py "$lib" > "$out.log" 2>&1 < /dev/null
    f=$(find "$out" -name "*counter_collection.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" "$name" $c <<'PY'
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if "k_broad" in n or "k_narrow" in n:
        tot.setdefault(n[-30:], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n, v in tot.items():
    k = 2 if sys.argv[3] == "FETCH_SIZE" else 1
    print("%-10s %-11s %-30s %8.2f MB per launch  avg %7.1f us under PMC" % (sys.argv[2], sys.argv[3], n, k * sum(x for x, _ in v) / len(v) * 1024 / 1e6, sum(t for _, t in v) / len(v) / 1e3))
PY
  done
done
