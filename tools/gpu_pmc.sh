# usage: tools/gpu_pmc.sh <tag> "<counter list 1>" "<counter list 2>" ...   (each list = one rocprofv3 --pmc pass over bench.py)
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-extra-paths ${BENCH_EXTRA:-} > $R/gpurun_out/pmc_${TAG}_$i.json 2> $R/gpurun_out/pmc_${TAG}_$i.err || { tail -5 $R/gpurun_out/pmc_${TAG}_$i.err; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_* 
