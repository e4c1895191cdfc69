#!/bin/bash
# Round-4 profile pack (run on the GPU box from the repo root): the headline bench alone and under --kernel-trace --stats (+ the
# union of the trunk's kernel intervals and the core stream's timeline from the same trace), per-layer times of the blocked and the
# NHWC trunk, five separate --pmc passes over one trunk pass, two --pmc passes over the NTM sequence kernels, and the other bench
# lines.  Outputs under gpurun_out/r04p/; the summaries are copied to profiles/ by hand.
OUT=gpurun_out/r04p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEP=${1:-all}
ALGO=${ALGO:-split3}        # the trunk form the --pmc passes profile: split3 (the default trunk) | winograd
if [ $STEP = all ] || [ $STEP = bench ]; then
timeout -k 10 300 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/prof_c2.err || exit 1
python3 scripts/trace_union.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_trunk_intervals.txt
python3 scripts/core_timeline.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_core_timeline.txt
cp $(ls -t $OUT/prof_c2/*/*_kernel_stats.csv | head -1) $OUT/bench_c2_kernel_stats.csv
echo "bench done"; tail -n 1 $OUT/bench_c2.json | cut -c1-220; cat $OUT/bench_c2_core_timeline.txt | tail -12
timeout -k 10 300 python3 scripts/r04/trunk_layout.py 640 > $OUT/wino43d_layers.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/r04/split3_layers.py 640 > $OUT/split3_layers.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/r04/split3_trunk.py 640 > $OUT/split3_trunk.txt 2>&1 || exit 1
grep -v amdgpu $OUT/split3_trunk.txt | tail -4
fi
if [ $STEP = all ] || [ $STEP = pmc ]; then
export NTK_TRUNK_SPLIT=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/trunk_pmc/p$i -- python3 scripts/dev_trunk_pass.py 640 $ALGO > $OUT/trunk_pmc_p$i.log 2>&1 || exit 1
  echo "trunk pmc pass $i done"
done
unset NTK_TRUNK_SPLIT
if [ $ALGO = split3 ]; then
  NOTE="fp32 trunk of round 4, split form: conv1_1 row kernel (fp32 NHWC) + conv1_2 .. conv4_3 on conv3x3_relu_bf16p_kernel<X3> (fp16 hi/lo parts, three MFMA products per fp32 product; mfma_busy = the fp16 pipe); collected by scripts/profile_r04.sh"
  TAB=$OUT/vgg_trunk_split3_hbm_traffic_pmc.csv
else
  NOTE="fp32 trunk of round 4, Winograd form: conv1_1 row kernel (NHWC) + nine fused Winograd F(4x4,3x3) layers on the eight-wave kernel with channel-blocked maps between them; collected by ALGO=winograd scripts/profile_r04.sh"
  TAB=$OUT/vgg_trunk_blocked_hbm_traffic_pmc.csv
fi
python3 scripts/trunk_pmc_table.py $OUT $TAB "$NOTE" || exit 1
tail -n 4 $TAB | cut -c1-220
bash scripts/r04/split3_pmc.sh > $OUT/split3_pmc.txt 2>&1 || exit 1
tail -n 12 $OUT/split3_pmc.txt
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/ntm_pmc/$grp -- python3 scripts/dev_ntm_timing.py 32 20 > $OUT/ntm_pmc_$grp.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py $OUT/ntm_pmc --match ntm_seq --out $OUT/ntm_pmc_summary.csv
cat $OUT/ntm_pmc_summary.csv
fi
if [ $STEP = all ] || [ $STEP = extra ]; then
C5="--model dnc --mem-size 512 --mem-dim 128 --batch 64 --seq-len 50 --conv-dtype bf16 --steps 3 --warmup 1"
timeout -k 10 300 python3 bench.py --model dnc > $OUT/bench_dnc_c3.json 2> $OUT/bench_dnc_c3.err || exit 1
timeout -k 10 400 python3 bench.py $C5 > $OUT/bench_dnc_c5.json 2> $OUT/bench_dnc_c5.err || exit 1
for extra in "--batch 64 --seq-len 20" "--mode infer" "--model dnc --mode infer" "--batch 64 --seq-len 30"; do timeout -k 10 300 python3 bench.py $extra --no-cpu-baseline 2> /dev/null | tail -n 1 | cut -c1-260; done > $OUT/bench_extra.txt
cat $OUT/bench_extra.txt
for f in dnc_c3 dnc_c5; do tail -n 1 $OUT/bench_$f.json | cut -c1-200; done
fi
