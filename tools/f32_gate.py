"""Go / no-go gate for an f32 first pass of the STFT.  CPU only, numpy.

For every episode of a corpus the chromaprint pipeline is evaluated twice: the STFT
(window, 4096-point real FFT, |X|^2, pitch-class fold) in f64 and in f32; everything behind the fold (FIR,
L2 norm, 16 area-ratio classifiers) in f64 from either chroma.  Reported:
  * the observed error of every classifier input, e = |log v32 - log v64|  (v = (1+a)/(1+b));
  * for radius r: the fraction of KEPT items (every step-th) whose f32 value has any of the 16 x 3 threshold
    comparisons inside r -- those items must be recomputed by the f64 kernel -- and the fraction of frame
    chunks (8 frame pairs = 16 frames) an f64 recompute of those items' 20 frames would touch;
  * the calibration of a data-dependent bound  |d chroma_c| <= k1 u chroma_c + k2 u sqrt(chroma_c E) + k3 u^2 E
    with E = sum |X|^2 of the frame (all bins), u = 2^-24.

usage: python tools/f32_gate.py [episodes=28] [minutes=24] [out.json]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import scipy.fft as sfft

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import synth  # noqa: E402
from tests.np_chromaprint import CLASSIFIERS, note_table  # noqa: E402

U = 2.0 ** -24
W64 = (1.0 / 32767.0) * (0.54 - 0.46 * np.cos(np.arange(4096) * 2.0 * np.pi / 4095))
W32 = W64.astype(np.float32)
K, NOTE = note_table()
ORDER = np.argsort(NOTE, kind="stable")
STARTS = np.searchsorted(NOTE[ORDER], np.arange(12))


def frames_of(pcm, dtype):
    n = len(pcm)
    nf = 0 if n < 4096 else (n - 4096) // 1365 + 1
    idx = np.arange(nf)[:, None] * 1365 + np.arange(4096)[None, :]
    return pcm.astype(dtype)[idx]


def chroma_of(pcm, dtype):
    """chroma [frames, 12] and E = sum over ALL bins of |X|^2 (two-sided), arithmetic in `dtype`."""
    out, energy = [], []
    w = W64 if dtype == np.float64 else W32
    n = len(pcm)
    nf = 0 if n < 4096 else (n - 4096) // 1365 + 1
    for f0 in range(0, nf, 1024):
        f1 = min(nf, f0 + 1024)
        idx = np.arange(f0, f1)[:, None] * 1365 + np.arange(4096)[None, :]
        x = pcm[idx].astype(dtype) * w
        spec = sfft.rfft(x, axis=1)
        assert spec.dtype == (np.complex128 if dtype == np.float64 else np.complex64)
        power = spec.real * spec.real + spec.imag * spec.imag
        out.append(np.add.reduceat(power[:, K[ORDER]], STARTS, axis=1))
        energy.append((x * x).sum(axis=1).astype(np.float64) * 4096.0)
    return np.concatenate(out).astype(np.float64), np.concatenate(energy)


def classifier_values(chroma):
    """log v for every raw item and classifier: [items, 16]; also the feature norms [rows]."""
    coef = np.array([0.25, 0.75, 1.0, 0.75, 0.25])
    rows = len(chroma) - 4
    fir = sum(coef[j] * chroma[j:j + rows] for j in range(5))
    norm = np.sqrt((fir ** 2).sum(axis=1))
    feat = np.where(norm[:, None] < 0.01, 0.0, fir / np.where(norm[:, None] == 0, 1, norm[:, None]))
    items = rows - 15
    integ = np.zeros((rows + 1, 13), dtype=np.longdouble)
    integ[1:, 1:] = feat.astype(np.longdouble).cumsum(axis=0).cumsum(axis=1)
    x = np.arange(items)

    def area(r1, c1, r2, c2):
        return (integ[x + r2, c2] - integ[x + r1, c2] - integ[x + r2, c1] + integ[x + r1, c1]).astype(np.float64)
    vals = np.empty((items, 16))
    for c, (t, y, h, wd, *_thr) in enumerate(CLASSIFIERS):
        if t == 0:
            a, b = area(0, y, wd, y + h), 0.0
        elif t == 1:
            a, b = area(0, y + h // 2, wd, y + h), area(0, y, wd, y + h // 2)
        elif t == 2:
            a, b = area(wd // 2, y, wd, y + h), area(0, y, wd // 2, y + h)
        elif t == 3:
            a = area(0, y + h // 2, wd // 2, y + h) + area(wd // 2, y, wd, y + h // 2)
            b = area(0, y, wd // 2, y + h // 2) + area(wd // 2, y + h // 2, wd, y + h)
        elif t == 4:
            h3 = h // 3
            a = area(0, y + h3, wd, y + 2 * h3)
            b = area(0, y, wd, y + h3) + area(0, y + 2 * h3, wd, y + h)
        else:
            w3 = wd // 3
            a = area(w3, y, 2 * w3, y + h)
            b = area(0, y, w3, y + h) + area(2 * w3, y, wd, y + h)
        vals[:, c] = np.log((1.0 + a) / (1.0 + b))
    return vals, norm


THR = np.array([c[4:7] for c in CLASSIFIERS])            # [16, 3]
RADII = [1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3]


def quantise(vals):
    q = (vals[:, :, None] >= THR[None]).sum(axis=2)
    return q


def analyse(pcm, step=2, chunk_frames=16):
    c64, e64 = chroma_of(pcm, np.float64)
    c32, _ = chroma_of(pcm, np.float32)
    v64, n64 = classifier_values(c64)
    v32, n32 = classifier_values(c32)
    err = np.abs(v32 - v64)
    kept = np.arange(0, len(v64), step)
    flips = int((quantise(v32[kept]) != quantise(v64[kept])).any(axis=1).sum())
    margin32 = np.abs(v32[:, :, None] - THR[None]).min(axis=(1, 2))          # per raw item, f32 pass
    nframes = len(c64)
    nchunks = (nframes + chunk_frames - 1) // chunk_frames
    table = {}
    for r in RADII:
        bad = kept[margin32[kept] <= r]
        touched = np.zeros(nframes, dtype=bool)
        for i in bad:
            touched[i:i + 20] = True
        chunks = np.zeros(nchunks, dtype=bool)
        chunks[np.nonzero(touched)[0] // chunk_frames] = True
        table[r] = (len(bad), len(kept), int(touched.sum()), nframes, int(chunks.sum()), nchunks)
    # calibration of the chroma error model
    dc = np.abs(c32 - c64)
    e = e64[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        k2 = np.where(c64 > 0, dc / (U * np.sqrt(c64 * e)), 0.0)
        k1 = np.where(c64 > 0, dc / (U * c64), 0.0)
    norm_cut = int((np.abs(n64 - 0.01) < 1e-6).sum())
    return dict(err_max=float(err.max()), err_p999=float(np.quantile(err, 0.999)), err_mean=float(err.mean()),
                flips=flips, table=table, k2_max=float(k2.max()), k2_p999=float(np.quantile(k2, 0.999)),
                k1_max=float(k1.max()), norm_near_cut=norm_cut,
                rel_chroma_err_max=float((dc.max(axis=1) / np.maximum(c64.max(axis=1), 1e-300)).max()))


def zoo():
    rng = np.random.default_rng(11)
    n = 14 * 11025
    t = np.arange(n) / 11025.0
    z = {}
    for amp in (0.6, 1, 2, 5, 20, 100, 1000, 12000, 60000):
        z[f"noise{amp}"] = np.clip(np.rint(rng.standard_normal(n) * amp), -32768, 32767)
    z["dc"] = np.full(n, 1234.0)
    z["dc+lsb"] = 20000 + (rng.random(n) < 0.5)
    z["chirp"] = 9000 * np.sin(2 * np.pi * (40 * t + 0.5 * 380 * t * t))
    z["impulses"] = np.where(np.arange(n) % 997 == 0, 30000.0, 0.0)
    z["nyquist"] = np.where(np.arange(n) % 2 == 0, 32767.0, -32768.0)
    z["fade-in"] = 8000 * np.sin(2 * np.pi * 440 * t) * np.clip((t - 5.0) / 6.0, 0, 1) ** 4
    z["two-tones"] = 3000 * np.sin(2 * np.pi * 261.63 * t) + 3000 * np.sin(2 * np.pi * 2093.0 * t)
    z["weak-inband+strong-5k"] = 30 * np.sin(2 * np.pi * 440 * t) + 30000 * np.sin(2 * np.pi * 5000.0 * t)
    z["square-fullscale"] = np.where(np.sin(2 * np.pi * 220 * t) >= 0, 32767.0, -32768.0)
    return {k: np.asarray(v).astype(np.int16) for k, v in z.items()}


def main():
    n_eps = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
    out = sys.argv[3] if len(sys.argv) > 3 else None
    report = {"corpus": f"{n_eps} x {minutes} min synthetic episodes (opening halves), step 2, chunks of 16 frames",
              "episodes": [], "zoo": {}}
    agg = {r: [0, 0, 0, 0, 0, 0] for r in RADII}
    worst = 0.0
    flips = 0
    for k in range(n_eps):
        ep = synth.make_episode(k, minutes * 60.0, 90.0 if minutes >= 10 else 20.0, 60.0 if minutes >= 10 else 0.0)
        res = analyse(ep.pcm[: len(ep.pcm) // 2])
        worst = max(worst, res["err_max"])
        flips += res["flips"]
        for r in RADII:
            for i, v in enumerate(res["table"][r]):
                agg[r][i] += v
        res["table"] = {str(r): v for r, v in res["table"].items()}
        report["episodes"].append(res)
        print(f"ep {k}: err max {res['err_max']:.3e} p99.9 {res['err_p999']:.3e} mean {res['err_mean']:.3e} "
              f"flips {res['flips']} k2 max {res['k2_max']:.2f} k1 max {res['k1_max']:.2f}", flush=True)
    for name, pcm in zoo().items():
        res = analyse(pcm, step=1)
        res["table"] = {str(r): v for r, v in res["table"].items()}
        report["zoo"][name] = res
        print(f"zoo {name}: err max {res['err_max']:.3e} flips {res['flips']} k2 max {res['k2_max']:.2f} "
              f"k1 max {res['k1_max']:.2f} relchroma {res['rel_chroma_err_max']:.2e} "
              f"uncertain@1e-5 {res['table']['1e-05'][0]}/{res['table']['1e-05'][1]}", flush=True)
    print(f"\ncorpus: worst |log v32 - log v64| = {worst:.3e}; items whose hash would flip without a fallback: {flips}")
    print("radius   items uncertain      frames touched     chunks(16 frames) touched   headroom over worst")
    summary = {}
    for r in RADII:
        a = agg[r]
        summary[str(r)] = dict(items_frac=a[0] / a[1], frames_frac=a[2] / a[3], chunks_frac=a[4] / a[5],
                               headroom=r / worst if worst else None)
        print(f"{r:7.0e}  {a[0]:8d}/{a[1]} = {a[0] / a[1]:.4%}   {a[2] / a[3]:.4%}   {a[4] / a[5]:.4%}"
              f"   {r / worst if worst else float('inf'):.1f}x")
    report["summary"] = summary
    report["worst_err"] = worst
    if out:
        with open(out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
