#!/bin/bash
# Round-3 final refresh of the three bench lines (run on the GPU box from the repo root); outputs under gpurun_out/r03c/.
set -e
OUT=gpurun_out/r03c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C5="--model dnc --mem-size 512 --mem-dim 128 --batch 64 --seq-len 50 --conv-dtype bf16 --steps 3 --warmup 1"
timeout -k 10 300 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/prof_c2.err
python3 scripts/trace_union.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_trunk_intervals.txt
python3 scripts/core_timeline.py $(ls -t $OUT/prof_c2/*/*_kernel_trace.csv | head -1) > $OUT/bench_c2_core_timeline.txt
timeout -k 10 300 python3 bench.py --model dnc > $OUT/bench_dnc_c3.json 2> $OUT/bench_dnc_c3.err
timeout -k 10 400 python3 bench.py $C5 > $OUT/bench_dnc_c5.json 2> $OUT/bench_dnc_c5.err
for extra in "--batch 64 --seq-len 20" "--mode infer" "--model dnc --mode infer" "--batch 64 --seq-len 30"; do timeout -k 10 300 python3 bench.py $extra --no-cpu-baseline 2> /dev/null | tail -n 1 | cut -c1-260; done > $OUT/bench_extra.txt
cat $OUT/bench_extra.txt
for f in c2 dnc_c3 dnc_c5; do tail -n 1 $OUT/bench_$f.json | cut -c1-200; done
cat $OUT/bench_c2_core_timeline.txt $OUT/bench_c2_trunk_intervals.txt | tail -16
