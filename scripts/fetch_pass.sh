#!/bin/bash
# FETCH_SIZE pass only (run on the GPU box): bash scripts/fetch_pass.sh <outdir>; per-layer reads then come from
# scripts/traffic_from_pmc.py with an earlier WRITE_SIZE pass (writes do not depend on the tile order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1
mkdir -p $out
export CCNET_LANES=0
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/fetch.log 2>&1 || { echo fetch failed; tail -5 $out/fetch.log; exit 1; }
echo fetch done
