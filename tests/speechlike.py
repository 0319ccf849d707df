"""Seeded speech-shaped test signals for the accuracy guard of the spectral route (tests/test_gpu_spectral.py,
tests/diag/guard_search.py). Test infrastructure only.

The guard (f2cnn_amd/csrc/f2_spectral.hip) has to catch rows whose ringing beyond the last sample dwarfs what lies
inside the row, and to leave ordinary level structure alone. `make(seed, n)` draws one utterance from a family of such
structures with random parameters; the same (seed, n, family) always gives the same int16 samples, so a worst case found by
the search can be pinned in a test by its seed."""
import numpy as np

FAMILIES = ("am_steps", "silence_then_speech", "syllables", "voiced", "late_burst", "clicks", "tone_bursts",
            "level_ramp", "speech_then_silence", "dc_noise", "one_sample", "end_onset")
ORDINARY = ("am_steps", "silence_then_speech", "syllables", "voiced")     # level structure every real recording has


def _tilt(rng, n, pole):
    """noise with a low-pass tilt (what speech looks like to the high channels): y[t] = x[t] + pole y[t-1]"""
    from scipy.signal import lfilter
    return lfilter([1.0], [1.0, -pole], rng.standard_normal(n))


def make(seed, n, family=None):
    """-> (int16 wave of n samples, family name, dict of the drawn parameters)"""
    rng = np.random.default_rng([int(seed), int(n)])
    drawn = FAMILIES[int(rng.integers(len(FAMILIES)))]       # (always drawn: the same samples with or without `family`)
    if family is None:
        family = drawn
    noise = rng.standard_normal(n)
    par = {}
    if family == "am_steps":
        # noise whose level steps by up to 70 dB every 25-200 ms (a talker between loud vowels and room noise)
        depth = par["depth_db"] = float(rng.uniform(20.0, 70.0))
        gain = np.empty(n)
        pos = 0
        while pos < n:
            seg = int(rng.integers(400, 3200))
            gain[pos:pos + seg] = 10.0 ** (-rng.uniform(0.0, depth) / 20.0)
            pos += seg
        w = (noise if rng.random() < 0.5 else _tilt(rng, n, 0.95) * 0.3) * gain * 9000.0
    elif family == "silence_then_speech":
        start = par["start"] = int(rng.uniform(0.05, 0.98) * n)
        floor = par["floor"] = float(rng.choice([0.0, 1.0, 10.0]))
        w = noise * floor
        w[start:] += rng.standard_normal(n - start) * 6000.0
    elif family == "syllables":
        # tilted noise under a 3-8 Hz syllable rhythm with pauses down to the noise floor
        rate = par["rate_hz"] = float(rng.uniform(3.0, 8.0))
        floor = par["floor_db"] = float(rng.uniform(-80.0, -30.0))
        t = np.arange(n) / 16000.0
        am = np.clip(np.sin(2 * np.pi * rate * t + rng.uniform(0, 6.28)), 0.0, None) ** float(rng.uniform(1.0, 6.0))
        w = _tilt(rng, n, float(rng.uniform(0.7, 0.99)))
        w = w / np.abs(w).max() * 25000.0 * (am + 10.0 ** (floor / 20.0))
    elif family == "voiced":
        # harmonic series of a gliding pitch, formant-like weighting, gated on and off
        f0 = rng.uniform(80.0, 250.0) * (1.0 + rng.uniform(-0.2, 0.2) * np.arange(n) / n)
        ph = 2 * np.pi * np.cumsum(f0) / 16000.0
        w = np.zeros(n)
        fc = rng.uniform(300.0, 3000.0, 3)
        for h in range(1, int(7000.0 / f0.max())):
            amp = sum(1.0 / (1.0 + ((h * f0.mean() - f) / 150.0) ** 2) for f in fc)
            w += amp * np.sin(h * ph + rng.uniform(0, 6.28))
        gate = np.ones(n)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n))
            gate[a:a + int(rng.integers(400, 6000))] = 10.0 ** (-rng.uniform(1.0, 4.0))
        par["gates"] = int((gate < 1).sum())
        w = w / np.abs(w).max() * 20000.0 * gate + noise * float(rng.choice([0.0, 2.0]))
    elif family == "late_burst":
        # low-level noise with a loud 5-50 ms burst that ENDS d samples before the end of the utterance
        d = par["d"] = min(n - 1, int(np.exp(rng.uniform(0.0, np.log(4000.0)))))
        length = par["length"] = int(rng.integers(80, 800))
        floor = par["floor"] = float(rng.choice([0.0, 1.0, 9.0, 90.0]))
        w = noise * floor
        stop = n - d
        w[max(0, stop - length):stop] += rng.standard_normal(stop - max(0, stop - length)) * 9000.0
    elif family == "clicks":
        k = par["k"] = int(rng.integers(1, 8))
        floor = par["floor"] = float(rng.choice([0.0, 0.0, 3.0, 30.0]))
        w = noise * floor
        pos = rng.integers(0, n, k)
        if rng.random() < 0.5:
            pos[0] = max(0, n - 1 - int(np.exp(rng.uniform(0.0, np.log(3000.0)))))      # one of them close to the end
        par["pos"] = [int(p) for p in pos]
        w[pos] += rng.choice([-1.0, 1.0], k) * rng.uniform(2000.0, 32000.0, k)
    elif family == "tone_bursts":
        f = par["f"] = float(np.exp(rng.uniform(np.log(100.0), np.log(7500.0))))
        t = np.arange(n) / 16000.0
        gate = (np.sin(2 * np.pi * rng.uniform(2.0, 20.0) * t + rng.uniform(0, 6.28)) > rng.uniform(-0.5, 0.9)).astype(float)
        w = 15000.0 * np.sin(2 * np.pi * f * t) * gate + noise * float(rng.choice([0.0, 1.0, 5.0]))
    elif family == "level_ramp":
        db = par["db"] = float(rng.uniform(-90.0, 90.0))                    # level change across the utterance
        w = noise * 10.0 ** (db * (np.arange(n) / n - (1.0 if db > 0 else 0.0)) / 20.0) * 20000.0
    elif family == "speech_then_silence":
        stop = par["stop"] = int(rng.uniform(0.02, 0.9) * n)
        w = noise * float(rng.choice([0.0, 1.0]))
        w[:stop] += rng.standard_normal(stop) * 8000.0
    elif family == "dc_noise":
        w = float(rng.uniform(-20000.0, 20000.0)) + noise * float(np.exp(rng.uniform(0.0, np.log(3000.0))))
    elif family == "one_sample":
        pos = par["pos"] = int(rng.integers(0, n)) if rng.random() < 0.5 else max(0, n - 1 - int(np.exp(rng.uniform(0.0, np.log(6000.0)))))
        w = np.zeros(n)
        w[pos] = float(rng.choice([-1.0, 1.0]) * rng.uniform(100.0, 32000.0))
    elif family == "end_onset":
        # near-silence, then speech-level sound that STARTS d samples before the end (cut mid-syllable)
        d = par["d"] = min(n - 1, int(np.exp(rng.uniform(np.log(2.0), np.log(5000.0)))))
        floor = par["floor"] = float(rng.choice([0.0, 1.0, 10.0]))
        w = noise * floor
        w[n - d:] += rng.standard_normal(d) * 8000.0
    else:
        raise ValueError(family)
    return np.clip(np.round(w), -32768, 32767).astype(np.int16), family, par
