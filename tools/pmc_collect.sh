#!/bin/bash
# Hardware-counter passes over tools/pmc_targets.py (one group per pass: FETCH_SIZE and WRITE_SIZE do not fit together,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters are collected with --kernel-trace only).
# Usage: pmc_collect.sh OUTDIR [summary.txt] [hemv_traffic.json]
set -u
OUT=${1:-gpurun_out/r03_pmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SUM=${2:-$OUT/summary.txt}
TRAF=${3:-$OUT/hemv_traffic.json}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
case "$SUM" in /*) ;; *) SUM="$PWD/$SUM";; esac
case "$TRAF" in /*) ;; *) TRAF="$PWD/$TRAF";; esac
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o p -- python "$R/tools/pmc_targets.py" > "$OUT/pass$i.log" 2>&1
done
python "$R/tools/pmc_summary.py" "$OUT" "$TRAF" > "$SUM" 2>&1
