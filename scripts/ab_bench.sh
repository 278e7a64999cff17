#!/bin/bash
# same-box A/B of two builds of the library (run on the GPU box): bash scripts/ab_bench.sh <variant .so> <outdir> [bench args]
# runs bench.py --breakdown alternately (new, old, new, old) and prints value / ms per step / MFMA-group figures of each run
var=$1; out=$2; shift 2
mkdir -p $out
for k in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export CCNET_HIP_LIB=$var; else unset CCNET_HIP_LIB; fi
    timeout -k 10 300 python3 -u bench.py --breakdown --no-cpu-baseline "$@" > $out/$which$k.json 2> $out/$which$k.err || { echo "$which$k failed"; tail -3 $out/$which$k.err; exit 1; }
    python3 - $out/$which$k.json $which$k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], "mfma ms", r["ms_per_forward"], "hbm ms", r["hbm_bound_launches"]["ms_per_forward"], flush=True)
PY
  done
done
