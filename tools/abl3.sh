for d in 0 1 2 4 8 3 6 7; do
    SAF_SDBG=$d python bench.py --no-cpu-baseline --steps 20 > gpurun_out/abl.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/abl.json'));print('sdbg=$d', d['roofline']['kernels_ms']['afstft_synthesis'])"
done
