import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p_ in (ROOT, ROOT / "tests", ROOT / "tests" / "golden"):
    if str(p_) not in sys.path:
        sys.path.insert(0, str(p_))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def saf():
    """The product API (libsaf_hip.so through ctypes). Fails loudly if the library is missing.

    On a GPU box torch's current stream becomes a stream of its own for the session: the tests hand
    `torch.cuda.current_stream().cuda_stream` to `saf.set_stream`, and the legacy default stream is 0 — which the library
    reads as "use your own (non-blocking) stream", i.e. NOT ordered against torch's kernels (a `torch.rand` input could still
    be in flight when the library's kernel reads it).  With a real stream, torch's and the library's work are one queue."""
    from spatial_audio_framework_amd import api
    from spatial_audio_framework_amd._lib import load
    load()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_stream(torch.cuda.Stream())
    except ImportError:
        pass
    return api
