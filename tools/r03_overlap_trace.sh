# usage (on the GPU box): bash tools/r03_overlap_trace.sh   -> gpurun_out/r03_overlap_trace.txt
# rocprofv3 kernel trace of the optional decode-beside-equaliser path (SAF_HIP_AMBI_DEC_OVERLAP=1): start / end of every dispatch of
# the publishing equaliser kernel and of the persistent decode kernel; the intervals of a step overlap.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SAF_HIP_AMBI_DEC_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_ov_trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-extra-paths --no-profile > $R/gpurun_out/r03_ov_bench.json 2> $R/gpurun_out/r03_ov_trace.err
python3 - <<PY > $R/gpurun_out/r03_overlap_trace.txt
import csv, glob
f = glob.glob("$R/gpurun_out/r03_ov_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "afstft_eq_kernel" in r["Kernel_Name"] or "dec_stream_kernel" in r["Kernel_Name"] or "band_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("rocprofv3 --kernel-trace of bench.py --steps 6 --warmup 2 with SAF_HIP_AMBI_DEC_OVERLAP=1 (tools/r03_overlap_trace.sh); times in us from the first dispatch")
print("%-34s %6s %12s %12s %10s" % ("kernel", "queue", "start", "end", "duration"))
last = {}
for r in rows[-24:]:
    n = r["Kernel_Name"].split("(")[0].replace("void saf::", "")[:34]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-34s %6s %12.1f %12.1f %10.1f" % (n, r.get("Queue_Id", "?"), s, e, e - s))
eq = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "afstft_eq_kernel" in r["Kernel_Name"]]
dc = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "dec_stream_kernel" in r["Kernel_Name"]]
ov = 0
for (a, b), (c, d) in zip(eq, dc):
    ov += max(0, min(b, d) - max(a, c))
print("equaliser launches %d, decode launches %d; summed overlap of the paired intervals %.1f us = %.0f %% of the equaliser kernels' time" % (len(eq), len(dc), ov / 1e3, 100.0 * ov / max(1, sum(b - a for a, b in eq))))
PY
cat $R/gpurun_out/r03_overlap_trace.txt
