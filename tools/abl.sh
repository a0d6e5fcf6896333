for d in 0 1 2 4 8 16 7 31; do
  for c in 0 64; do
    SAF_DBG=$d SAF_CHUNK=$c python bench.py --no-cpu-baseline --steps 20 > gpurun_out/abl.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/abl.json'));print('dbg=$d chunk=$c', d['roofline']['kernels_ms'])"
  done
done
