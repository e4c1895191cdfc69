#!/bin/bash
# Dev: issue / memory-pipe counters of one Winograd layer (conv3_2 shape), one --pmc pass per counter group
# (never combined with API tracing).  Run on the GPU box from the repo root; summaries: scripts/pmc_summary.py.
set -e
OUT=${1:-gpurun_out/r02/wino_pmc}
ALGO=${2:-wino}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_MISC" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_TA_BUSY TCP_TCP_TA_DATA_STALL_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 scripts/dev_wino_one.py 640 $ALGO > "$OUT/p$i.log" 2>&1
  echo "pass $i done"
done
python3 scripts/pmc_summary.py "$OUT" --match conv3x3_wino --out "$OUT/summary.csv"
cat "$OUT/summary.csv"
