#!/bin/bash
# GPU box: LDS bank-conflict share per trunk layer (one --pmc pass over one trunk pass; kernel trace only)
OUT=gpurun_out/r04_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export NTK_TRUNK_SPLIT=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc -- python3 scripts/dev_trunk_pass.py 640 ${1:-split3} > $OUT/pmc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
cc = max(glob.glob("gpurun_out/r04_lds/pmc/**/*counter_collection.csv", recursive=True))
d = collections.defaultdict(dict); names = {}
for r in csv.DictReader(open(cc)):
    if "conv" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]:
        continue
    i = int(r["Dispatch_Id"]); names[i] = r["Kernel_Name"]
    d[i][r["Counter_Name"]] = d[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(d)[-10:]
for i in ids:
    c = d[i]
    print("%d %-52s conflict/active %.3f  lds insts %.3e" % (i, names[i][30:82], c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_INSTS_LDS"]))
PY
