#!/bin/bash
# Round profile (run on the GPU box): kernel-trace stats + HBM traffic counters of the default bench command.
#   bash scripts/profile_bench.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
# same command on one stream (CCNET_LANES=0): per-kernel durations without overlap, the ones bench.py's serial event timing must agree with
export CCNET_LANES=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_serial -o t -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/trace_serial.log 2>&1 || { echo serial trace failed; tail -5 $out/trace_serial.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/fetch.log 2>&1 || { echo fetch failed; tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/write.log 2>&1 || { echo write failed; tail -5 $out/write.log; exit 1; }
unset CCNET_LANES
tail -1 $out/trace.log | cut -c1-300
echo profile done
