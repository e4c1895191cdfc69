#!/bin/bash
# GPU box: parity of the wave-specialised NTM kernels, then timing of both forms alone and inside the bench
OUT=gpurun_out/${1:-r04_ws}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_ntm_gpu.py tests/test_ntm_train_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py tests/test_copy_task_gpu.py -m gpu -x -q -k "not dnc and not config5" > $OUT/pytest.txt 2>&1
rc=$?
echo "pytest rc=$rc" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== ws forms" > $OUT/timing.txt
timeout -k 10 120 python scripts/dev_ntm_timing.py 32 20 >> $OUT/timing.txt 2>&1 || exit 1
echo "== res forms" >> $OUT/timing.txt
NTK_NTM_FWD_FORM=res NTK_NTM_BWD_FORM=res timeout -k 10 120 python scripts/dev_ntm_timing.py 32 20 >> $OUT/timing.txt 2>&1 || exit 1
grep -v amdgpu.ids $OUT/timing.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_ws.json 2> $OUT/bench_ws.err || exit 1
NTK_NTM_FWD_FORM=res NTK_NTM_BWD_FORM=res timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_res.json 2> $OUT/bench_res.err || exit 1
python - <<PY
import json
for n in ("ws","res"):
    d=json.loads(open("$OUT/bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["breakdown_ms"]["steady_state_step"], d["breakdown_ms"]["vgg_trunk_stream"], d["breakdown_ms"]["ntm_fwd_bwd_opt_stream"], d["memory_step"]["us_per_step"], d["memory_step_bptt"]["us_per_step"], d["breakdown_ms"]["cu_seconds"]["cu_time_bound_ms"])
PY
