"""Per-layer table of the trunk's PMC passes collected by scripts/profile_r02.sh / profile_r03b.sh (<root>/trunk_pmc/p1..p5) ->
profiles/r02_vgg_trunk_wino43_hbm_traffic_pmc.csv (default).  usage: python scripts/trunk_pmc_table.py [root] [out.csv] [round-note]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02p"
outp = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_vgg_trunk_wino43_hbm_traffic_pmc.csv"


def load(pass_dir):
    f = max(glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run
    return list(csv.DictReader(open(f)))


def per_layer(rows, counter):
    d = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "conv" not in k or "pack" in k:
            continue
        d.setdefault(int(r["Dispatch_Id"]), 0.0)
        d[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = [d[k] for k in sorted(d)]
    assert len(vals) % 10 == 0, len(vals)
    n = len(vals) // 10
    return [sum(vals[i + 10 * p] for p in range(n)) / n for i in range(10)]


fetch = per_layer(load(root + "/trunk_pmc/p1"), "FETCH_SIZE")
write = per_layer(load(root + "/trunk_pmc/p2"), "WRITE_SIZE")
p3 = load(root + "/trunk_pmc/p3")
busy, mfma = per_layer(p3, "SQ_BUSY_CYCLES"), per_layer(p3, "SQ_VALU_MFMA_BUSY_CYCLES")
nm, nv = per_layer(p3, "SQ_INSTS_MFMA"), per_layer(p3, "SQ_INSTS_VALU")
p4 = load(root + "/trunk_pmc/p4")
conf, lact = per_layer(p4, "SQ_LDS_BANK_CONFLICT"), per_layer(p4, "SQ_LDS_IDX_ACTIVE")
p5 = load(root + "/trunk_pmc/p5")
wc, wa, wi, ai = (per_layer(p5, c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
F = 640
spec = [("conv1_1", 224, 3, 64, False), ("conv1_2", 224, 64, 64, True), ("conv2_1", 112, 64, 128, False), ("conv2_2", 112, 128, 128, True),
        ("conv3_1", 56, 128, 256, False), ("conv3_2", 56, 256, 256, False), ("conv3_3", 56, 256, 256, True), ("conv4_1", 28, 256, 512, False),
        ("conv4_2", 28, 512, 512, False), ("conv4_3", 28, 512, 512, False)]
out = open(outp, "w")
out.write("# rocprofv3 --kernel-trace --pmc <group> (five separate passes: FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU |\n")
out.write("# SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY) on `python3 scripts/dev_trunk_pass.py 640 winograd`\n")
note = sys.argv[3] if len(sys.argv) > 3 else "default fp32 trunk of round 2: conv1_1 row kernel + nine fused Winograd F(4x4,3x3) layers, conv_wino43.hip; collected by scripts/profile_r02.sh"
out.write("# (%s); mean of 6 launches per layer;\n" % note)
out.write("# tabulated by scripts/trunk_pmc_table.py.\n")
out.write("# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B -> doubled; WRITE_SIZE as read. Counter unit KB.\n")
out.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines * 1024 SIMDs): share of SIMD cycles with the fp32 MFMA pipe busy.\n")
out.write("# lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; parked / issue_stall / issuing = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.\n")
out.write("layer,fetch_bytes(FETCH_SIZE*1024*2),write_bytes(WRITE_SIZE*1024),algorithmic_bytes(in+weights+out),traffic/algorithmic,mfma_busy,valu_per_mfma,lds_conflict,parked,issue_stall,issuing\n")
tf = tw = ta = 0
for i, (n, H, ci, co, pool) in enumerate(spec):
    fb, wb = fetch[i] * 1024 * 2, write[i] * 1024
    oh = H // 2 if pool else H
    alg = (F * H * H * ci + 9 * ci * co + F * oh * oh * co) * 4
    tf += fb; tw += wb; ta += alg
    mb = mfma[i] / (busy[i] / 32 * 1024) if busy[i] else 0
    out.write("%s,%.4e,%.4e,%.4e,%.2f,%.3f,%.2f,%.3f,%.3f,%.3f,%.3f\n" % (n, fb, wb, alg, (fb + wb) / alg, mb, (nv[i] - nm[i]) / max(nm[i], 1),
                                                                           conf[i] / max(lact[i], 1), wa[i] / wc[i], wi[i] / wc[i], ai[i] / wc[i]))
out.write("total,%.4e,%.4e,%.4e,%.2f,,,,,,\n" % (tf, tw, ta, (tf + tw) / ta))
out.close()
print(open(outp).read())
print("fetch %.4e write %.4e" % (tf, tw))
