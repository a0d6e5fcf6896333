for d in 0 1 2 3; do
    SAF_GDBG=$d python bench.py --no-cpu-baseline --steps 20 --instances 32 > gpurun_out/abl.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/abl.json'));print('gdbg=$d', d['roofline']['kernels_ms'])"
done
