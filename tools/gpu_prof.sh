set -e
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cat $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head
