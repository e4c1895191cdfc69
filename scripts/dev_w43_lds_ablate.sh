set -e
# usage: dev_w43_lds_ablate.sh [H,cin,cout]   (default conv3_2: 56,256,256); variants build_abl/libntmtrack_w43abl{2,4,16}.so
SHAPE=${1:-56,256,256}
OUT=gpurun_out/r02x/ldsabl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in prod w43abl2 w43abl4 w43abl16; do
  if [ $v = prod ]; then unset NTK_LIB_PATH; else export NTK_LIB_PATH=build_abl/libntmtrack_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$v -- python3 scripts/dev_wino_one.py 640 wino43 $SHAPE > $OUT/$v.log 2>&1
  python3 scripts/pmc_summary.py $OUT/$v --match conv3x3_wino43 --out $OUT/$v.csv
  echo $v; cat $OUT/$v.csv
done
