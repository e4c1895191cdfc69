#!/bin/bash
# Round-3 profile pack (run on the GPU box from the repo root).  Kernel-trace stats of the headline bench (configs[1]) and of
# BASELINE configs[4] (DNC 512 x 128 on the memory-partitioned cluster kernels), then PMC passes (each on its own, --kernel-trace
# + --pmc only) for the HBM-side traffic of the mp kernels.  Outputs under gpurun_out/r03p/; the summaries are copied to profiles/.
set -e
OUT=gpurun_out/r03p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C5="--model dnc --mem-size 512 --mem-dim 128 --batch 64 --seq-len 50 --conv-dtype bf16 --steps 3 --warmup 1"
timeout -k 10 300 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/prof_c2.err
echo "c2 done"
timeout -k 10 300 python3 bench.py --model dnc > $OUT/bench_dnc_c3.json 2> $OUT/bench_dnc_c3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 bench.py --model dnc --no-cpu-baseline > $OUT/bench_dnc_c3_under_rocprof.json 2> $OUT/prof_c3.err
echo "c3 done"
timeout -k 10 400 python3 bench.py $C5 > $OUT/bench_dnc_c5.json 2> $OUT/bench_dnc_c5.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 bench.py $C5 --no-cpu-baseline > $OUT/bench_dnc_c5_under_rocprof.json 2> $OUT/prof_c5.err
echo "c5 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mp -- python3 scripts/dev_mp_pmc.py 200 > $OUT/mp_stats.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/mp_pmc/p$i -- python3 scripts/dev_mp_pmc.py 200 > $OUT/mp_pmc_p$i.log 2>&1
  echo "mp pmc pass $i done"
done
python3 scripts/pmc_summary.py $OUT/mp_pmc --match dnc_mp --out $OUT/mp_pmc_summary.csv
cat $OUT/mp_pmc_summary.csv
NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so timeout -k 10 300 python3 scripts/dev_mp_prof.py 512 128 64 300 4 > $OUT/mp_stamps_c5.txt 2>&1
NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so timeout -k 10 300 python3 scripts/dev_mp_prof.py 256 64 32 400 4 > $OUT/mp_stamps_c3.txt 2>&1
for extra in "--batch 64 --seq-len 20" "--mode infer" "--model dnc --mode infer"; do timeout -k 10 300 python3 bench.py $extra --no-cpu-baseline 2> /dev/null | tail -n 1 | cut -c1-260; done > $OUT/bench_extra.txt
cat $OUT/bench_extra.txt
for f in c2 dnc_c3 dnc_c5; do tail -n 1 $OUT/bench_$f.json | cut -c1-200; done
ls $OUT/prof_c2/*/ $OUT/prof_c5/*/ $OUT/prof_mp/*/ | head -40
