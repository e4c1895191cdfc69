#!/bin/bash
# Dev: LDS bank-conflict counters of the F(4x4) kernel on one layer shape per tile-block shape (product library)
set -e
OUT=gpurun_out/r02x/ldschk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shp in 224,64,64 112,128,128 56,256,256 28,512,512; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/s$shp -- python3 scripts/dev_wino_one.py 640 wino43 $shp > $OUT/s$shp.log 2>&1
  python3 scripts/pmc_summary.py $OUT/s$shp --match conv3x3_wino43 --out $OUT/s$shp.csv
  tail -n 1 $OUT/s$shp.csv
done
