#!/bin/bash
# Kernel A/B on one GPU box.  In the build container:   tools/ab_variants.sh build name1="<file stem>:<flags>" name2=...
#   -> tools/probes/_bin/variants/libsaf_hip_<name>.so (git-ignored, travels with gpurun)
# On the GPU box:  tools/ab_variants.sh run "<bench.py arguments>"  -> one line per variant
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/tools/probes/_bin/variants
if [ "$1" = build ]; then
    shift; mkdir -p $V
    for spec in "$@"; do
        name=${spec%%=*}; rest=${spec#*=}; stem=${rest%%:*}; flags=${rest#*:}
        ( cd $R && env "SAF_HIP_FLAGS_$stem=$flags" python -m spatial_audio_framework_amd.build > /dev/null && cp spatial_audio_framework_amd/libsaf_hip.so $V/libsaf_hip_$name.so && touch spatial_audio_framework_amd/csrc/$stem.hip )
        echo "built $name ($stem: $flags)"
    done
    ( cd $R && python -m spatial_audio_framework_amd.build > /dev/null )
else
    shift
    for so in $V/libsaf_hip_*.so; do
        n=$(basename $so .so); n=${n#libsaf_hip_}
        SAF_HIP_LIB=$so python $R/bench.py $1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-24s' % '$n', j['value'], j['ms_per_step'], j['roofline'].get('kernels_ms'))
"
    done
fi
