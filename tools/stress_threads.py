#!/usr/bin/env python3
"""initCodec on a second thread while the audio thread keeps calling process (SURVEY 8b "Threading"): the reference mutes
the output while the codec initialises; nothing may crash or hang, and once the init is done the output must be the new
configuration's."""
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from spatial_audio_framework_amd import api
from util import frames

F = 128
a = api.AmbiDec(F)
a.setMasterDecOrder(3); a.setOutputConfigPreset(21); a.initCodec(); a.init(48000)
x = frames(1, 16, 400 * F)
stop = False
stats = {"blocks": 0, "muted": 0}


def audio():
    i = 0
    while not stop:
        y = a.process(np.ascontiguousarray(x[:, (i % 400) * F:(i % 400 + 1) * F]), 24)
        stats["blocks"] += 1
        if i > 20 and not y.any():
            stats["muted"] += 1
        assert np.isfinite(y).all()
        i += 1


t = threading.Thread(target=audio)
t.start()
for k in range(12):
    time.sleep(0.05)
    a.setOutputConfigPreset(21 if k % 2 else 29)         # structural change -> codec not initialised
    a.setDecMethod(0, 1 + k % 4)
    a.initCodec()                                        # on this (second) thread
stop = True
t.join(timeout=30)
assert not t.is_alive()
y = a.process(np.ascontiguousarray(x[:, :F]), 64)
print("blocks", stats["blocks"], "muted while initialising", stats["muted"], "final loudspeakers", a.getNumLoudspeakers(), "ok")
