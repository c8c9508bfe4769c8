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

The f32 STFT is the kernel's own arithmetic (radix 16 x 3 in place, correctly rounded window and twiddle tables,
explicit FMAs), stepped on the CPU by tests/cpu_emu (`emu`, default), or scipy's pocketfft in f32 (`scipy`).
Second table: the DATA-DEPENDENT radius the product uses, r = K * S with S = max over the item's 16 feature rows of
u * sqrt(E_row / norm_row) (E_row, norm_row: the FIR'd frame energy and chroma norm) -- the observed error divided by S
is what K has to cover.

usage: python tools/f32_gate.py [episodes=28] [minutes=24] [out.json] [emu|scipy]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import synth  # noqa: E402
from tests.np_chromaprint import CLASSIFIERS, chroma_of, classifier_values  # noqa: E402

U = 2.0 ** -24


_EMU = None


def chroma_emu32(pcm):
    """f32 pass exactly as stft_chroma32_kernel computes it (tests/cpu_emu/emu.cpp): chroma, E = N sum x^2."""
    global _EMU
    import ctypes as C
    if _EMU is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        _EMU = C.CDLL(os.path.join(root, "tests", "cpu_emu", "libemu.so"))
        _EMU.emu_stft_chroma_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    nf = 0 if len(pcm) < 4096 else (len(pcm) - 4096) // 1365 + 1
    ch, en = np.zeros((nf, 12)), np.zeros((nf, 4), dtype=np.float32)
    _EMU.emu_stft_chroma_stream(pcm.ctypes.data, len(pcm), 1, ch.ctypes.data, en.ctypes.data)
    return ch, en.astype(np.float64).sum(axis=1) * 16384.0     # the kernel's samples carry a factor 1/2


THR = np.array([c[4:7] for c in CLASSIFIERS])            # [16, 3]
RADII = [1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3]
KS = [4, 8, 16, 32, 64, 128, 256]
MODE = "emu"


def quantise(vals):
    q = (vals[:, :, None] >= THR[None]).sum(axis=2)
    return q


def analyse(pcm, step=2, chunk_frames=16):
    c64, e64 = chroma_of(pcm, np.float64)
    c32, e32 = chroma_emu32(pcm) if MODE == "emu" else (chroma_of(pcm, np.float32)[0], e64)
    v64, n64 = classifier_values(c64)
    v32, n32, scale = classifier_values(c32, e32)
    err = np.abs(v32 - v64)
    kept = np.arange(0, len(v64), step)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(scale > 0, err.max(axis=1) / np.maximum(scale, 1e-300), 0.0)
    err_unscaled = float(err.max(axis=1)[scale == 0].max()) if (scale == 0).any() else 0.0
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
    ktable = {}
    for kk in KS:
        bad = kept[margin32[kept] <= kk * scale[kept]]
        touched = np.zeros(nframes, dtype=bool)
        for i in bad:
            touched[i:i + 20] = True
        c8 = np.zeros((nframes + 7) // 8, dtype=bool)            # the product's fallback chunks: 4 frame pairs
        c8[np.nonzero(touched)[0] // 8] = True
        ktable[kk] = (len(bad), len(kept), int(touched.sum()), nframes, int(c8.sum()), len(c8))
    # calibration of the chroma error model
    dc = np.abs(c32 - c64)
    e = e64[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        k2 = np.where(c64 > 0, dc / (U * np.sqrt(c64 * e)), 0.0)
        k1 = np.where(c64 > 0, dc / (U * c64), 0.0)
    norm_cut = int((np.abs(n64 - 0.01) < 1e-6).sum())
    return dict(ratio_max=float(ratio.max()), scale_min=float(scale[scale > 0].min()) if (scale > 0).any() else 0.0,
                scale_max=float(scale.max()), err_where_scale_is_zero=err_unscaled, ktable=ktable,
                err_max=float(err.max()), err_p999=float(np.quantile(err, 0.999)), err_mean=float(err.mean()),
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
    z["weakest-inband+strong-5k"] = 3 * np.sin(2 * np.pi * 440 * t) + 30000 * np.sin(2 * np.pi * 5000.0 * t)
    z["weak-inband+strong-dc"] = 300 * np.sin(2 * np.pi * 440 * t) + 30000
    z["noise3+strong-5k"] = np.clip(np.rint(rng.standard_normal(n) * 3 + 30000 * np.sin(2 * np.pi * 5100 * t)), -32768, 32767)
    z["weak-inband+strong-12hz"] = 100 * np.sin(2 * np.pi * 440 * t) + 30000 * np.sin(2 * np.pi * 12 * t)
    z["chirp-fullscale"] = 30000 * np.sin(2 * np.pi * (20 * t + 0.5 * 390 * t * t))
    z["quiet-tone"] = 20 * np.sin(2 * np.pi * 440 * t)
    z["tone+lsb-noise"] = np.clip(np.rint(12000 * np.sin(2 * np.pi * 523.25 * t) + rng.standard_normal(n) * 0.7), -32768, 32767)
    return {k: np.asarray(v).astype(np.int16) for k, v in z.items()}


def main():
    global MODE
    if len(sys.argv) > 4:
        MODE = sys.argv[4]
    n_eps = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
    out = sys.argv[3] if len(sys.argv) > 3 else None
    report = {"corpus": f"{n_eps} x {minutes} min synthetic episodes (opening halves), step 2, chunks of 16 frames",
              "episodes": [], "zoo": {}}
    agg = {r: [0, 0, 0, 0, 0, 0] for r in RADII}
    kagg = {k: [0, 0, 0, 0, 0, 0] for k in KS}
    worst_ratio = zoo_ratio = 0.0
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
        for kk in KS:
            for i, v in enumerate(res["ktable"][kk]):
                kagg[kk][i] += v
        worst_ratio = max(worst_ratio, res["ratio_max"])
        res["table"] = {str(r): v for r, v in res["table"].items()}
        res["ktable"] = {str(r): v for r, v in res["ktable"].items()}
        report["episodes"].append(res)
        print(f"ep {k}: err max {res['err_max']:.3e} p99.9 {res['err_p999']:.3e} mean {res['err_mean']:.3e} "
              f"flips {res['flips']} err/S max {res['ratio_max']:.3f} S [{res['scale_min']:.2e}, {res['scale_max']:.2e}]",
              flush=True)
    for name, pcm in zoo().items():
        res = analyse(pcm, step=1)
        res["table"] = {str(r): v for r, v in res["table"].items()}
        zoo_ratio = max(zoo_ratio, res["ratio_max"])
        k32 = res["ktable"][32]
        res["ktable"] = {str(r): v for r, v in res["ktable"].items()}
        report["zoo"][name] = res
        print(f"zoo {name}: err max {res['err_max']:.3e} flips {res['flips']} err/S max {res['ratio_max']:.3f} "
              f"S [{res['scale_min']:.2e}, {res['scale_max']:.2e}] err where S = 0: {res['err_where_scale_is_zero']:.1e} "
              f"uncertain at K = 32: {k32[0]}/{k32[1]}", flush=True)
    print(f"\ncorpus: worst |log v32 - log v64| = {worst:.3e}; items whose hash would flip without a fallback: {flips}")
    print("radius   items uncertain      frames touched     chunks(16 frames) touched   headroom over worst")
    summary = {}
    for r in RADII:
        a = agg[r]
        summary[str(r)] = dict(items_frac=a[0] / a[1], frames_frac=a[2] / a[3], chunks_frac=a[4] / a[5],
                               headroom=r / worst if worst else None)
        print(f"{r:7.0e}  {a[0]:8d}/{a[1]} = {a[0] / a[1]:.4%}   {a[2] / a[3]:.4%}   {a[4] / a[5]:.4%}"
              f"   {r / worst if worst else float('inf'):.1f}x")
    print(f"\ndata-dependent radius r = K * S: worst err / S on the corpus {worst_ratio:.3f}, on the signal zoo {zoo_ratio:.3f}")
    print("   K   items uncertain      frames touched     chunks (4 pairs) touched   headroom over worst err/S (corpus, zoo)")
    ksummary = {}
    top = max(worst_ratio, zoo_ratio)
    for kk in KS:
        a = kagg[kk]
        ksummary[str(kk)] = dict(items_frac=a[0] / a[1], frames_frac=a[2] / a[3], chunks_frac=a[4] / a[5],
                                 headroom=kk / top if top else None)
        print(f"{kk:4d}  {a[0]:8d}/{a[1]} = {a[0] / a[1]:.4%}   {a[2] / a[3]:.4%}   {a[4] / a[5]:.4%}   {kk / top:.1f}x")
    report["summary"] = summary
    report["summary_data_dependent"] = ksummary
    report["worst_err_over_scale"] = {"corpus": worst_ratio, "zoo": zoo_ratio}
    report["f32_arithmetic"] = MODE
    report["worst_err"] = worst
    if out:
        with open(out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
