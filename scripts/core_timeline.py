"""Dev: timeline of one steady-state step on the core stream (and the trunk stream's idle gaps) from a rocprofv3 kernel trace.
usage: python scripts/core_timeline.py <kernel_trace.csv> [fwd-kernel-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "ntm_seq_fwd"
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
q = [r['Queue_Id'] for r in rows if key in r['Kernel_Name']][0]
core = [r for r in rows if r['Queue_Id'] == q]
fw = [i for i, r in enumerate(core) if key in r['Kernel_Name']]
a, b = fw[12], fw[13]
t0 = core[a]['s']
prev_e = core[a - 1]['e']
tot_gap = tot_k = 0.0
for r in core[a:b]:
    gap = (r['s'] - prev_e) / 1e3
    tot_gap += max(gap, 0); tot_k += (r['e'] - r['s']) / 1e3
    if (r['e'] - r['s']) > 100e3 or gap > 20:
        print("%-64s start %9.1f us dur %9.1f us gap-before %8.1f us" % (r['Kernel_Name'][:64], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap))
    prev_e = r['e']
print("core stream: period %.1f us, kernels %.1f us, gaps %.1f us" % ((core[b]['s'] - core[a]['s']) / 1e3, tot_k, tot_gap))
conv = [r for r in rows if "conv" in r['Kernel_Name'] and core[a]['s'] <= r['s'] < core[b]['s']]
conv.sort(key=lambda r: r['s'])
end = None
for r in conv:
    if end is not None and r['s'] - end > 200e3:
        print("trunk idle %.1f us before %s at %.1f us" % ((r['s'] - end) / 1e3, r['Kernel_Name'][:50], (r['s'] - t0) / 1e3))
    end = r['e'] if end is None else max(end, r['e'])
