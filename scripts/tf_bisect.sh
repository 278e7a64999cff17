#!/bin/bash
# Run-to-run determinism of the fused Transformer for build variants:  bash scripts/tf_bisect.sh <outdir> "<variant names>" [enc,dec ...]
# (build_variants/lib_<name>.so; "main" = the shipped library).  GPU box.
set -o pipefail
out=$1; shift; names=$1; shift; mkdir -p $out
for m in $names; do
  lib=build_variants/lib_$m.so; [ $m = main ] && lib=ccnet_amd/libccnet_hip.so
  echo "=== variant $m" >> $out/bisect.log
  CCNET_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 -u scripts/tf_det_probe.py "$@" 2>&1 | grep -v -i "warn\|TransformerEncoder" >> $out/bisect.log || { echo "variant $m failed" >> $out/bisect.log; exit 1; }
done
echo done >> $out/bisect.log
