#!/bin/bash
# Same-box A/B runs of bench.py's headline region (round 3).  On the GPU box:
#   tools/r03_ab_overlap.sh <tag> "ENV=.. ENV=.. [-- bench args]" "..."   -> gpurun_out/r03_ab_<tag>.txt
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_ab_$1.txt; shift
: > $O
for cfg in "$@"; do
    echo "== $cfg" >> $O
    envs=${cfg%%--*}; extra=""; case "$cfg" in *--*) extra="--${cfg#*--}";; esac
    env $envs python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-paths --no-other-configs $extra 2>>$O.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('kernels_ms'))
" >> $O
done
cat $O
