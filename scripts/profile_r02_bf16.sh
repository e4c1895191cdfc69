#!/bin/bash
# Round-2 profile of config 5's bf16 trunk (run on the GPU box from the repo root): kernel-trace stats, then separate PMC
# passes for HBM traffic and MFMA busy cycles.  Outputs under gpurun_out/r02b/.
set -e
OUT=gpurun_out/r02b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/dev_trunk_pass.py 640 bf16 > $OUT/trunk_bf16.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 bf16 > $OUT/pmc_p$i.log 2>&1
  echo "bf16 pmc pass $i done"
done
python3 scripts/pmc_summary.py $OUT/pmc --match conv --out $OUT/trunk_bf16_pmc_summary.csv
cat $OUT/trunk_bf16_pmc_summary.csv
tail -n 1 $OUT/trunk_bf16.log
