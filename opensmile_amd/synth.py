"""Synthetic 16 kHz mono int16 corpus -- the measurement contract of SURVEY.md §8(d).

Utterance u (0-based) uses rng = np.random.default_rng(1234 + u):
  f0 ~ U[80, 300] Hz with 5 Hz +-3 % vibrato, 10 harmonics (amplitude 1/h),
  4 Hz raised-cosine syllabic envelope, peak 0.3 FS, plus white Gaussian noise
  sigma = 0.01 FS; every 10th utterance is noise only (sigma = 0.1 FS);
  utterance 0 is all zeros (exercises melfloor / log floors) and utterance 1 is
  a +-0.9 FS 100 Hz square wave (clipping / harmonic edge).
Rounded to nearest int16.
"""
import numpy as np

FS = 16000


def utterance(u, n_samples, fs=FS):
    rng = np.random.default_rng(1234 + u)
    t = np.arange(n_samples, dtype=np.float64) / fs
    if u == 0:
        x = np.zeros(n_samples)
    elif u == 1:
        x = 0.9 * np.sign(np.sin(2 * np.pi * 100.0 * t))
    elif u % 10 == 0:
        x = rng.normal(0.0, 0.1, n_samples)
    else:
        f0 = rng.uniform(80.0, 300.0)
        inst = f0 * (1.0 + 0.03 * np.sin(2 * np.pi * 5.0 * t))
        phase = 2 * np.pi * np.cumsum(inst) / fs
        x = np.zeros(n_samples)
        for h in range(1, 11):
            x += np.sin(h * phase) / h
        env = 0.5 * (1.0 - np.cos(2 * np.pi * 4.0 * t))
        x *= env
        x *= 0.3 / max(np.max(np.abs(x)), 1e-12)
        x += rng.normal(0.0, 0.01, n_samples)
    y = np.rint(np.clip(x, -1.0, 1.0) * 32767.0)
    return y.astype(np.int16)


def corpus(n_utt, n_samples, first=0, fs=FS):
    """Packed corpus: (pcm int16 [n_utt*n_samples], offsets int64 [n_utt+1])."""
    pcm = np.empty(n_utt * n_samples, dtype=np.int16)
    for i in range(n_utt):
        pcm[i * n_samples:(i + 1) * n_samples] = utterance(first + i, n_samples, fs)
    offs = np.arange(n_utt + 1, dtype=np.int64) * n_samples
    return pcm, offs


def corpus_tiled(n_utt, n_samples, n_unique=32, fs=FS):
    """Bench-sized corpus: n_unique distinct utterances generated per the
    contract, tiled to n_utt (generation of 1000 x 10 s in numpy would take
    minutes; content repeats, work per frame does not change)."""
    base, _ = corpus(min(n_unique, n_utt), n_samples, 0, fs)
    reps = -(-n_utt // min(n_unique, n_utt))
    pcm = np.tile(base, reps)[: n_utt * n_samples].copy()
    offs = np.arange(n_utt + 1, dtype=np.int64) * n_samples
    return pcm, offs
