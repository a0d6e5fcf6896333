# usage (on the GPU box):  bash tools/gpu_profile.sh <tag>      e.g. r01_v2
# 1. rocprofv3 --kernel-trace --stats of the bench command (value region alone, and all regions) -> gpurun_out/<tag>_kernel_stats*.csv
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes, no trace domains) -> gpurun_out/<tag>_traffic.json
# 3. the plain bench line (with cpu_baseline)                        -> gpurun_out/<tag>_bench.json
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# the timed region of `value` alone (the averages of afstft_eq / band_gemm are those of the bench line's kernels_ms) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-extra-paths > $R/gpurun_out/${TAG}_bench_traced.json 2> $R/gpurun_out/${TAG}_trace.err
cp $(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
# ... and every region of the default run (uniform / general equaliser launches and the transform path mixed per kernel name)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_all -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2> $R/gpurun_out/${TAG}_trace_all.err
cp $(find $R/gpurun_out/${TAG}_trace_all -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_all_regions.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_pmc_$C -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-extra-paths > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$C.err
done
TRAFFIC_SOURCE="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes of bench.py --steps 6 --no-extra-paths, tools/gpu_profile.sh ${TAG}" python3 $R/tools/traffic_summary.py --frames-per-launch=16384 $R/gpurun_out/${TAG}_pmc_FETCH_SIZE $R/gpurun_out/${TAG}_pmc_WRITE_SIZE > $R/gpurun_out/${TAG}_traffic.json
cat $R/gpurun_out/${TAG}_traffic.json
cd $R
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json
head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
