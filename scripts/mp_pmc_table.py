"""Tabulate the HBM-side traffic of the memory-partitioned DNC cluster kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and one
--kernel-trace run of scripts/dev_mp_pmc.py (dispatch order per repetition: inference forward, recording forward, BPTT).

usage: python scripts/mp_pmc_table.py <pmc_dir> <trace_dir> <S> <B> <algorithmic_bytes_per_sequence_step>"""
import collections
import csv
import glob
import sys

pmc_dir, trace_dir, S, B, alg = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
per = collections.defaultdict(list)
for f in sorted(glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "dnc_mp" not in r["Kernel_Name"]:
            continue
        key = (int(r["Dispatch_Id"]), "bwd" if "bwd" in r["Kernel_Name"] else "fwd", r["Counter_Name"])
        acc[key] = acc.get(key, 0) + float(r["Counter_Value"])
    for (d, kern, ctr), v in acc.items():
        per[(kern, ctr)].append((d, v))


def pick(kern, ctr, which):
    xs = sorted(per[(kern, ctr)])
    if kern == "fwd":
        xs = [xs[i] for i in range(len(xs)) if i % 2 == which]
    return sum(v for _, v in xs) / len(xs)


dur = {"fwd": [], "bwd": []}
name = {}
for f in glob.glob(trace_dir + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dnc_mp" in r["Kernel_Name"]:
            k = "bwd" if "bwd" in r["Kernel_Name"] else "fwd"
            dur[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
            name[k] = r["Kernel_Name"].split("(")[0].replace("void ", "").strip() + r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">") + 1] if "<" in r["Kernel_Name"] else k
fw = [d for _, d in sorted(dur["fwd"])]
bw = [d for _, d in sorted(dur["bwd"])]
print("kernel,fetch_bytes(FETCH_SIZE*1024*2),write_bytes(WRITE_SIZE*1024),traffic_bytes,traffic_per_sequence_step,algorithmic_per_sequence_step,traffic/algorithmic,ms,hbm_side_GBps,frac_of_8TBps")
for label, kern, which, ms in (("forward, inference (link updated in place)", "fwd", 0, sum(fw[0::2]) / len(fw[0::2])),
                               ("forward, recording for BPTT", "fwd", 1, sum(fw[1::2]) / len(fw[1::2])),
                               ("BPTT", "bwd", 0, sum(bw) / len(bw))):
    f = pick(kern, "FETCH_SIZE", which) * 1024 * 2
    w = pick(kern, "WRITE_SIZE", which) * 1024
    t = f + w
    print('"%s",%.4e,%.4e,%.4e,%.0f,%d,%.2f,%.3f,%.0f,%.3f' % (label, f, w, t, t / (S * B), alg, t / (S * B) / alg, ms, t / (ms * 1e-3) / 1e9, t / (ms * 1e-3) / 8e12))
