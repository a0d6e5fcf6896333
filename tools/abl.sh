for d in 0 1 2 4 8 16 3 6 7 12; do
    SAF_DBG=$d python bench.py --no-cpu-baseline --steps 20 > gpurun_out/abl.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/abl.json'));print('dbg=$d', d['roofline']['kernels_ms']['afstft_analysis'])"
done
