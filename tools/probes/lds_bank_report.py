import csv, glob, collections, os, sys
acc = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("void k"):
            acc[(int(r["Grid_Size"]) // 64 - 256)][r["Counter_Name"]] = float(r["Counter_Value"])
strides = [16, 256, 258, 260, 264, 272, 288, 320]
ins_n = ["read_b64", "read2_b64", "write_b64", "write2_b64", "read_b128", "read_b32"]
pat_n = ["rows8x8B", "transposeW", "secondR", "rows32x8B", "column16"]
print("cycles per wave-instruction (LDS_IDX_ACTIVE / INSTS_LDS-1-init), conflicts in brackets; columns = slot stride in dwords")
print("%-11s %-11s " % ("instr", "pattern") + " ".join("%10d" % s for s in strides))
for ins in range(6):
    for pat in range(5):
        row = []
        for s in range(8):
            c = acc.get((ins * 5 + pat) * 8 + s)
            if not c: row.append("     -    "); continue
            n = c["Grid_Size"] if "Grid_Size" in c else None
            wg = 256 + (ins * 5 + pat) * 8 + s
            per = 256.0 * wg            # REP wave-instructions per workgroup
            row.append("%5.1f(%4.1f)" % (c["SQ_LDS_IDX_ACTIVE"] / per, c["SQ_LDS_BANK_CONFLICT"] / per))
        print("%-11s %-11s " % (ins_n[ins], pat_n[pat]) + " ".join(row))
