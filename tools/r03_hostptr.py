#!/usr/bin/env python3
"""The host_pointer section of bench.py alone (one handle / 8 / 32 handles on as many host threads through the unchanged
ambi_dec_process).   [GPU_MAX_HW_QUEUES=N] python tools/r03_hostptr.py"""
import json
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load
r = bench.host_pointer_section(load(), api)
r["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
print(json.dumps(r))
