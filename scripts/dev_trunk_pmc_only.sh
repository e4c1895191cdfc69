set -e
OUT=gpurun_out/r02p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NTK_TRUNK_SPLIT=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/trunk_pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 winograd > $OUT/trunk_pmc_p$i.log 2>&1
done
python3 scripts/pmc_summary.py $OUT/trunk_pmc --match conv --out $OUT/trunk_pmc_summary.csv
tail -n 3 $OUT/trunk_pmc_summary.csv | cut -c1-200
