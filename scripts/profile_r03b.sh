#!/bin/bash
# Round-3 profile pack, part b (run on the GPU box from the repo root): the headline bench and the fp32 trunk after the F(4x4)
# kernel went to eight waves per workgroup.  Kernel-trace stats of the headline bench, the union of the trunk's kernel intervals,
# PMC passes for the trunk's per-layer HBM-side traffic / MFMA busy cycles (each pass on its own: --kernel-trace + --pmc only),
# the per-layer timing of both kernel forms, the per-section stamps and the ablation builds of the eight-wave kernel.
# Outputs under gpurun_out/r03b/; the summaries are copied to profiles/ (scripts/trunk_pmc_table.py writes the PMC table).
set -e
OUT=gpurun_out/r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/prof_c2.err
python3 scripts/trace_union.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_trunk_intervals.txt
echo "c2 done"
timeout -k 10 300 python3 bench.py --model dnc > $OUT/bench_dnc_c3.json 2> $OUT/bench_dnc_c3.err
echo "c3 done"
export NTK_TRUNK_SPLIT=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/trunk_pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 winograd > $OUT/trunk_pmc_p$i.log 2>&1
  echo "trunk pmc pass $i done"
done
unset NTK_TRUNK_SPLIT
timeout -k 10 300 python3 scripts/dev_wino43d.py 640 > $OUT/wino43d_layers.txt 2>&1
timeout -k 10 100 python3 scripts/dev_c11_time.py >> $OUT/wino43d_layers.txt 2>&1
for shp in "640 56 256 256" "640 224 64 64" "640 28 512 512"; do
  NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so timeout -k 10 120 python3 scripts/dev_wino43d_prof.py $shp >> $OUT/wino43d_stamps.txt 2>&1
done
timeout -k 10 100 python3 scripts/dev_wino43d_time.py 1 > $OUT/wino43d_ablation.txt 2>&1
timeout -k 10 100 python3 scripts/dev_wino43d_time.py 0 >> $OUT/wino43d_ablation.txt 2>&1
for a in 1 2 4 8 16 6 31; do
  NTK_LIB_PATH=build_abl/libntmtrack_d$a.so timeout -k 10 100 python3 scripts/dev_wino43d_time.py 1 >> $OUT/wino43d_ablation.txt 2>&1
done
for extra in "--batch 64 --seq-len 20" "--mode infer" "--features-roi"; do timeout -k 10 300 python3 bench.py $extra --no-cpu-baseline 2> /dev/null | tail -n 1 | cut -c1-260; done > $OUT/bench_extra.txt
cat $OUT/bench_extra.txt
tail -n 1 $OUT/bench_c2.json | cut -c1-200
tail -n 1 $OUT/bench_dnc_c3.json | cut -c1-200
cat $OUT/bench_c2_trunk_intervals.txt | tail -5
