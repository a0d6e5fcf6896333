# usage (on the GPU box): bash tools/probes/run_probes.sh   -> gpurun_out/r02_lds_probe.txt, r02_lds_bank.txt, r02_valu_rate.txt
# (binaries are built in the build container: see the hipcc line at the top of each tools/probes/*.hip)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lds_probe_f -- $R/tools/probes/_bin/lds_probe > /dev/null 2>&1
python3 - <<PY > $R/gpurun_out/r02_lds_probe.txt
import csv,glob,collections,os
acc=collections.defaultdict(dict)
for f in glob.glob("$R/gpurun_out/lds_probe_f/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(int(r["Dispatch_Id"]),r["Kernel_Name"])][r["Counter_Name"]]=float(r["Counter_Value"])
names={2:"fold writes (thread = sample position), plain layout",3:"fold writes, XOR-swizzled layout",4:"overlap-add reads, plain",5:"overlap-add reads, swizzled",
       6:"128-point FFT of 16 slots, twiddles [p][j] (new)",7:"128-point FFT, twiddles [j][p] (round 1), swizzled slots",8:"inverse FFT, twiddles [p][j]",
       9:"bin pairs, lane = bin of one slot (equaliser kernel)",10:"bin pairs, 16 lanes = one bin of 16 slots (analysis kernel), plain",11:"the same, XOR-swizzled (round 1 layout)"}
print("LDS probe (tools/probes/lds_probe.hip): one phase of the filterbank kernels per kernel, 1024 workgroups x 64 repetitions; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS")
for (d,k) in sorted(acc):
    if d in names:
        c=acc[(d,k)]; print("%-72s LDS instr %10d  active cycles %11d  conflict cycles %11d  (%.1f %%)  cycles/instr %.1f" % (names[d], c["SQ_INSTS_LDS"], c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"], 100*c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1), c["SQ_LDS_IDX_ACTIVE"]/max(c["SQ_INSTS_LDS"],1)))
PY
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lds_bank_f -- $R/tools/probes/_bin/lds_bank > /dev/null 2>&1
python3 $R/tools/probes/lds_bank_report.py $R/gpurun_out/lds_bank_f > $R/gpurun_out/r02_lds_bank.txt
$R/tools/probes/_bin/valu_rate > $R/gpurun_out/r02_valu_rate.txt
$R/tools/probes/_bin/issue_mix | grep -v SALU >> $R/gpurun_out/r02_valu_rate.txt
cat $R/gpurun_out/r02_lds_probe.txt
