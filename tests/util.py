"""Shared helpers for the tests: seeded synthetic input and error metrics."""
import numpy as np


def xorshift_noise(seed, n):
    """Uniform noise in [-1, 1) from xorshift32 (SURVEY §8d synthetic input), float32."""
    out = np.empty(n, np.float32)
    s = np.uint32(seed if seed != 0 else 1)
    # vectorised in chunks: run 4096 independent streams seeded by a scalar xorshift
    lanes = min(n, 4096)
    st = np.empty(lanes, np.uint32)
    x = int(s)
    for i in range(lanes):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        st[i] = x
    pos = 0
    while pos < n:
        st ^= (st << np.uint32(13))
        st ^= (st >> np.uint32(17))
        st ^= (st << np.uint32(5))
        m = min(lanes, n - pos)
        out[pos:pos + m] = (st[:m] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 / (1 << 24)) - np.float32(1.0)
        pos += m
    return out


def frames(seed, nCh, nSamples):
    return xorshift_noise(seed, nCh * nSamples).reshape(nCh, nSamples)


def relrms(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.sqrt((np.abs(a - b) ** 2).sum() / max(float((np.abs(b) ** 2).sum()), 1e-30)))


def maxabs(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max())


def canon_faces(F):
    """Canonical form of a face list: each triangle's vertices sorted, rows sorted."""
    F = np.sort(np.asarray(F, np.int64), axis=1)
    return F[np.lexsort(F.T[::-1])]
