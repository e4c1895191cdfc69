"""Wall-clock time covered by the trunk's conv kernels per trunk pass, from a rocprofv3 kernel trace of bench.py.

With the trunk pass split over two streams (ntmtrack.vgg.VGG16Conv43.split_streams) its kernels overlap pairwise, so the
SUM of their durations (what --stats reports) is about twice the time the pass takes; the UNION of their intervals is the
figure that compares with bench.py's breakdown_ms.vgg_trunk_stream (HIP events on the trunk stream).
usage: python scripts/trace_union.py <kernel_trace.csv> [parts per pass = 2]"""
import csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
conv = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
               if "conv3x3_wino43" in r["Kernel_Name"] or "conv_c3_rows_kernel" in r["Kernel_Name"]
               or "conv3x3_relu_bf16p_kernel" in r["Kernel_Name"]), key=lambda x: x[0])          # (the split / bf16 patch-form layers)
# a pass = one conv1_1 launch per stream part + nine layers each: consecutive passes never overlap (the trunk stream joins its
# side streams at the end of a pass), so the start-sorted kernels split into equal groups
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # stream parts of a trunk pass (VGG16Conv43.split_streams)
per = 10 * nparts
passes = [[(s, e) for s, e, _n in conv[i:i + per]] for i in range(0, len(conv) - per + 1, per)]

def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs

us = [union(p) / 1e6 for p in passes]; ss = [sum(e - s for s, e in p) / 1e6 for p in passes]; ks = [len(p) for p in passes]
full = list(range(len(passes)))
mid = full[4:-3] if len(full) > 10 else full
print("trunk passes found: %d (kernels per pass: %d)" % (len(passes), max(ks)))
print("steady-state passes %d..%d: union of the conv kernels' intervals %.2f ms per pass (min %.2f, max %.2f); sum of their durations %.2f ms per pass"
      % (mid[0], mid[-1], sum(us[i] for i in mid) / len(mid), min(us[i] for i in mid), max(us[i] for i in mid), sum(ss[i] for i in mid) / len(mid)))
