#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the trunk (scripts/profile_r03b.sh's first two PMC passes) into gpurun_out/r03d
set -e
OUT=gpurun_out/r03d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NTK_TRUNK_SPLIT=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/trunk_pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 winograd > $OUT/trunk_pmc_p$i.log 2>&1
  echo "trunk pmc pass $i done"
done
