#!/bin/bash
# usage (GPU box): bash tools/run_fuzz.sh <log name> <first seed> <n seeds> <configs per seed> [all]
cd "$GRAFT_REPO_ROOT"
log=gpurun_out/$1
if [ "$5" = "all" ]; then export NBK_FUZZ_ALL=1; fi
timeout -k 10 1000 python tools/fuzz_campaign.py $2 $3 $4 > "$log" 2>&1 < /dev/null
echo "rc $?"; grep -c "^seed" "$log"; tail -2 "$log"; echo "lines with MISMATCH: $(grep -c MISMATCH "$log" || true)"
