#!/bin/bash
# Round-2 profile pack (run on the GPU box from the repo root): kernel-trace stats of the headline bench and of the DNC
# bench, then PMC passes (each on its own, --kernel-trace + --pmc only) for the HBM traffic and MFMA busy cycles of the
# default trunk.  Outputs under gpurun_out/r02p/; the summaries are copied into profiles/ by hand.
set -e
OUT=gpurun_out/r02p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/prof_c2.err
echo "c2 profile done"
timeout -k 10 400 python3 bench.py --model dnc > $OUT/bench_dnc_c3.json 2> $OUT/bench_dnc_c3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dnc -- python3 bench.py --model dnc --no-cpu-baseline > $OUT/bench_dnc_c3_under_rocprof.json 2> $OUT/prof_dnc.err
echo "dnc profile done"
python3 scripts/trace_union.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_trunk_intervals.txt
cat $OUT/bench_c2_trunk_intervals.txt
# per-layer counters: one launch per layer and pass (the product splits a pass over two streams: two launches per layer)
export NTK_TRUNK_SPLIT=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/trunk_pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 winograd > $OUT/trunk_pmc_p$i.log 2>&1
  echo "trunk pmc pass $i done"
done
python3 scripts/pmc_summary.py $OUT/trunk_pmc --match conv --out $OUT/trunk_pmc_summary.csv
cat $OUT/trunk_pmc_summary.csv
tail -n 1 $OUT/bench_c2.json | cut -c1-300
