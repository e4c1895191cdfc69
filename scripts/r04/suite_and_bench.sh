#!/bin/bash
# GPU box: full -m gpu suite, then the headline bench (skipped when the suite was killed at its limit)
OUT=gpurun_out/${1:-r04_a}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.txt 2>&1
rc=$?
echo "pytest rc=$rc" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed at its limit: no bench"; exit $rc; fi
timeout -k 10 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "bench rc=$?"
tail -c 600 $OUT/bench_c2.json
exit $rc
