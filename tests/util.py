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


def ulp_perturb(a, seed, ulps=1.0):
    """`a` (float32 / complex64) with every component moved by up to `ulps` units in its last place, random size and sign: the
    change any reordering of float32 operations upstream may cause"""
    rng = np.random.default_rng(seed)
    a = np.asarray(a)
    eps = np.float64(2.0 ** -23) * ulps
    if np.iscomplexobj(a):
        return (a.real * (1 + eps * rng.uniform(-1, 1, a.shape)) + 1j * a.imag * (1 + eps * rng.uniform(-1, 1, a.shape))).astype(a.dtype)
    return (a * (1 + eps * rng.uniform(-1, 1, a.shape))).astype(a.dtype)


def oracle_sensitivity(run, ref, metric, seeds=(11, 12, 13)):
    """How far the ORACLE's own output moves (in `metric(out, ref)`) when its input moves by one unit in the last place:
    run(seed) returns the oracle's output on an input perturbed with that seed, `ref` is its output on the exact input.
    A bound on |library - oracle| for an ill-conditioned function is a multiple of this, not a constant picked by hand."""
    return max(metric(run(sd), ref) for sd in seeds)


def canon_faces(F):
    """Canonical form of a face list: each triangle's vertices sorted, rows sorted."""
    F = np.sort(np.asarray(F, np.int64), axis=1)
    return F[np.lexsort(F.T[::-1])]


def fibonacci_dirs_deg(N):
    """N roughly uniform directions [azimuth 0..360, elevation] in degrees (Fibonacci sphere)."""
    i = np.arange(N) + 0.5
    el = np.degrees(np.arcsin(1.0 - 2.0 * i / N))
    az = np.mod(np.degrees(np.pi * (1.0 + 5.0 ** 0.5) * i), 360.0)
    return np.stack([az, el], 1).astype(np.float32)


def synth_hrirs(N=836, L=256, fs=48000, seed=7):
    """Synthetic stand-in for the reference's default HRIR set (absent from the checkout): same shapes
    ([N][2][L] at 48 kHz, 836 directions) with a spherical-head delay/level model and a direction-dependent decaying
    tail, deterministic.  Returns (hrirs float32 [N][2][L], dirs_deg float32 [N][2] with azimuth in 0..360)."""
    rng = np.random.default_rng(seed)
    dirs = fibonacci_dirs_deg(N)
    az, el = np.radians(dirs[:, 0]), np.radians(dirs[:, 1])
    lat = np.sin(az) * np.cos(el)                     # +1: source at the left ear side (az = +90)
    h = np.zeros((N, 2, L), np.float64)
    n = np.arange(L)
    for ear, sgn in ((0, 1.0), (1, -1.0)):
        delay = 40.0 - sgn * lat * 0.00035 * fs                                  # samples; near ear earlier
        gain = 10.0 ** (sgn * lat * 6.0 / 20.0)
        for d in range(N):
            k = int(np.floor(delay[d])); fr = delay[d] - k
            h[d, ear, k] += (1 - fr) * gain[d]; h[d, ear, k + 1] += fr * gain[d]
            tail = rng.normal(size=L) * np.exp(-(n - k) / (12.0 + 6.0 * (1 + sgn * lat[d]))) * (n > k + 1) * 0.15 * gain[d]
            h[d, ear] += tail
    # gentle low-pass (gain 0.4 at Nyquist: measured HRIRs keep energy up there; a null would be blown up by the
    # diffuse-field equalisation together with its rounding noise)
    ker = np.array([0.15, 0.7, 0.15])
    h = np.apply_along_axis(lambda v: np.convolve(v, ker, mode="same"), 2, h)
    return h.astype(np.float32), dirs
