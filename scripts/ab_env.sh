#!/bin/bash
# same-box A/B of one environment switch (run on the GPU box): bash scripts/ab_env.sh <VAR=value> <outdir> [bench args]
# runs bench.py --breakdown alternately without and with the switch, twice
kv=$1; out=$2; shift 2
mkdir -p $out
for k in 1 2; do
  for which in base with; do
    if [ $which = with ]; then export "$kv"; else unset "${kv%%=*}"; fi
    timeout -k 10 300 python3 -u bench.py --breakdown --no-cpu-baseline "$@" > $out/$which$k.json 2> $out/$which$k.err || { echo "$which$k failed"; tail -3 $out/$which$k.err; exit 1; }
    python3 - $out/$which$k.json $which$k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], "mfma ms", r["ms_per_forward"], "hbm ms", r["hbm_bound_launches"]["ms_per_forward"], flush=True)
PY
  done
done
