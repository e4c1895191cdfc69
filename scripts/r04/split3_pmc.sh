#!/bin/bash
# GPU box: MFMA-pipe busy share and effective clock of the split-form layers (one --pmc pass, kernel trace only)
# usage: split3_pmc.sh [bf16]      (bf16: the same for config 5's bf16 patch-form layers, scripts/r04/bf16_trunk.py)
OUT=gpurun_out/r04_s3${1:+_$1}
if [ "$1" = bf16 ]; then TARGET="scripts/r04/bf16_trunk.py 640"; else TARGET="scripts/r04/split3_layers.py 640 only3"; fi
export PMC_ROOT=$OUT/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $TARGET > $OUT/pmc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["PMC_ROOT"]
cc = max(glob.glob(root + "/**/*counter_collection.csv", recursive=True))
kt = max(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))
dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
d = collections.defaultdict(dict)
names = {}
for r in csv.DictReader(open(cc)):
    if "bf16p_kernel" not in r["Kernel_Name"]:
        continue
    i = int(r["Dispatch_Id"]); names[i] = r["Kernel_Name"]
    d[i][r["Counter_Name"]] = d[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("dispatch,grid,duration_us,clock_GHz(GRBM_GUI_ACTIVE/8/duration),mfma_busy(SQ_VALU_MFMA_BUSY_CYCLES/(SQ_BUSY_CYCLES/32*1024))")
for i in sorted(d):
    c = d[i]
    if dur[i] < 500000:
        continue
    print("%d,%s,%.1f,%.3f,%.3f" % (i, names[i][60:110], dur[i] / 1e3, c["GRBM_GUI_ACTIVE"] / 8 / dur[i], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024)))
PY
