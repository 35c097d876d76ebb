#!/bin/bash
# usage (on the GPU box): bash tools/round_battery.sh <tag>
# the side measurements DESIGN.md §6 quotes, one log each under gpurun_out/<tag>_*.log (copy the ones to keep into profiles/);
# every step is bounded and the chain stops at the first one that fails or times out
cd "$GRAFT_REPO_ROOT"
tag=$1; L=numbotics_amd/csrc/libnbk.so
step() { name=$1; shift; timeout -k 10 240 "$@" > gpurun_out/${tag}_$name.log 2>&1 < /dev/null; rc=$?; echo "$name rc $rc"; return $rc; }
step thr_time_c2 python tools/thr_time.py c2 &&
step thr_time_c3 python tools/thr_time.py c3 &&
step thr_time_c2m python tools/thr_time.py c2m &&
step thr_time_c5m python tools/thr_time.py c5m &&
step seed_time_0 python tools/seed_time.py $L 0.0 20 &&
step seed_time_1e-6 python tools/seed_time.py $L 1e-6 20 &&
step dist_time_c2 python tools/dist_time.py c2 &&
step dist_time_c5m python tools/dist_time.py c5m &&
step iris_time_c2 python tools/iris_time.py c2 &&
step iris_time_c5m python tools/iris_time.py c5m &&
step big_batch_1e7 python tools/big_batch.py &&
step two_streams python tools/two_stream_time.py &&
step scalar_latency python tools/scalar_latency.py &&
step many_obstacles python tools/many_obstacles.py &&
step fk_big python tools/fk_big.py &&
step fk_time python tools/fk_time.py
