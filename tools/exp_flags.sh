# usage (GPU box): bash tools/exp_flags.sh "<flags A>" "<flags B>" ...   — rebuilds afstft_kernels.hip with each flag set and prints the bench line's kernel times (two runs each)
R=$GRAFT_REPO_ROOT
cd $R
for F in "$@"; do
  touch spatial_audio_framework_amd/csrc/afstft_kernels.hip
  SAF_HIP_FLAGS_afstft_kernels="$F" python -m spatial_audio_framework_amd.build > /dev/null 2>&1 || { echo "build failed: $F"; continue; }
  for r in 1 2; do
    python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$F |', d['value'], d['roofline']['kernels_ms'], d['band_independent_path']['value'])"
  done
done
