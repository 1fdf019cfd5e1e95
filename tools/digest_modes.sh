#!/bin/bash
# Which tile selection disagrees?  tools/fullsize_digest.py of one bench configuration under several SMI_GEMM settings;
# prints the two step losses of each (bit-identical GEMM generations give identical lines).
cd "$(dirname "$0")/.."
CFG=${1:-sd14_512_b1_r4_c3lier}; shift
MODES=${@:-default 128 v1 160 5ph no5ph 8ph no8ph 64 notune nosplit}
for m in $MODES; do
  unset SMI_GEMM SMI_GEMM_TUNE SMI_GEMM_SPLITK
  case $m in
    default) ;;
    notune) export SMI_GEMM_TUNE=0;;
    nosplit) export SMI_GEMM_SPLITK=0;;
    *) export SMI_GEMM=$m;;
  esac
  r=$(python3 tools/fullsize_digest.py --config $CFG --out /tmp/digest_$m.pt 2>&1 | grep "digest written" | sed 's/.*\.pt//')
  echo "$CFG SMI_GEMM=$m: $r"
done
