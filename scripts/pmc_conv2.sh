#!/bin/bash
# SQ counters of one conv shape for two builds of the library (run on the GPU box): bash scripts/pmc_conv2.sh <shape> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; out=$2; mkdir -p $out
for lib in NEW OLD; do
  if [ $lib = OLD ]; then export CCNET_HIP_LIB=$GRAFT_REPO_ROOT/ccnet_amd/csrc/abl/lib_OLD.so; else unset CCNET_HIP_LIB; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/$lib -o p -- python3 scripts/conv_bench.py $shape > $out/$lib.log 2>&1 || { echo pmc $lib failed; tail -5 $out/$lib.log; exit 1; }
done
python3 - $out <<'PY'
import csv, sys, collections, glob
out=sys.argv[1]
for lib in ("NEW","OLD"):
    f=glob.glob(out+"/"+lib+"/**/p_counter_collection.csv", recursive=True)
    agg=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f[0])):
        if "conv_patch" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    d={k:v for k,v in agg.items()}
    wc=d.get("SQ_WAVE_CYCLES",1)
    print(lib, {k: round(v/wc,4) for k,v in d.items()}, "wave_cycles", wc)
PY
