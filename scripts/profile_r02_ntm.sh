#!/bin/bash
# Round-2 HBM-side traffic of the NTM sequence kernels (run on the GPU box from the repo root): two separate PMC passes.
set -e
OUT=gpurun_out/r02n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc/$grp -- python3 scripts/dev_ntm_timing.py 32 20 > $OUT/pmc_$grp.log 2>&1
done
python3 scripts/pmc_summary.py $OUT/pmc --match ntm_seq --out $OUT/ntm_pmc_summary.csv
cat $OUT/ntm_pmc_summary.csv
