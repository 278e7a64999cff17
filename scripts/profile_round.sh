#!/bin/bash
# Round profile (run on the GPU box):  bash scripts/profile_round.sh <outdir under gpurun_out>
# headline config: kernel-trace stats (lanes and one stream) + HBM traffic counters (scripts/profile_bench.sh), then a bench
# line and a one-stream kernel trace for each of the other BASELINE configs.
set -o pipefail
out=$1
bash scripts/profile_bench.sh $out || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -u bench.py --breakdown > $out/bench_net.json 2> $out/bench_net.err || exit 1
export CCNET_LANES=0
for cfg in "net_deploy --batch 32 --precision bf16" "net_amff --batch 64 --precision fp32" "net_small_target --size 1280 --batch 16 --precision bf16 --conf 0.001"; do
  set -- $cfg; kind=$1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$kind -o t -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --kind $cfg > $out/trace_$kind.log 2>&1 || { echo trace $kind failed; tail -5 $out/trace_$kind.log; exit 1; }
done
unset CCNET_LANES
for cfg in "net_deploy --batch 32 --precision bf16" "net_amff --batch 64 --precision fp32" "net_small_target --size 1280 --batch 16 --precision bf16 --conf 0.001"; do
  set -- $cfg; kind=$1
  timeout -k 10 400 python3 -u bench.py --breakdown --no-cpu-baseline --kind $cfg > $out/bench_$kind.json 2> $out/bench_$kind.err || { echo bench $kind failed; tail -5 $out/bench_$kind.err; exit 1; }
done
echo round profile done
