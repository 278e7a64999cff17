#!/bin/bash
# per-kernel split of the gate ops (run on the GPU box): bash scripts/gate_prof.sh <outdir>
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 scripts/gate_bench.py > $out/gate_bench.txt 2>&1 || { tail -5 $out/gate_bench.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 scripts/gate_bench.py 5 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/t_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-90s %5s %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
cat $out/gate_bench.txt
