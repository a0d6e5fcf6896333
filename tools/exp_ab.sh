# usage (GPU box): bash tools/exp_ab.sh   — A/B of tools/_ab/A.hip vs B.hip as csrc/afstft_kernels.hip on the SAME box (boxes differ by +-3 %)
R=$GRAFT_REPO_ROOT
cd $R
for V in A B A B; do
  cp tools/_ab/$V.hip spatial_audio_framework_amd/csrc/afstft_kernels.hip
  python -m spatial_audio_framework_amd.build > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V |', d['value'], d['roofline']['kernels_ms'], d['band_independent_path']['value'])"
done
