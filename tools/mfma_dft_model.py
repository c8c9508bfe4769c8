"""CPU model (numpy) of the f32 first pass's three radix-16 DFT stages as SPLIT-f16 MATRIX PRODUCTS (round 6's design
study): what error does a first pass on v_mfma_f32_32x32x16_f16 carry, in the radius' own unit S?

One 4096-point complex transform per frame pair (z = a + i b), n = 256 n0 + 16 n1 + n2, k = k0 + 16 k1 + 256 k2:
  stage 0 over n0, twiddle W^((16 n1 + n2) k0), stage 1 over n1, twiddle W^(16 n2 k1), stage 2 over n2.
A stage is Y[32 x 256] = F[32 x 32] X[32 x 256] with F = [[C, S], [-S, C]] of the DFT-16 (re rows, im rows): operands
split x = hi + lo (two f16), F = Fh + Fl, products Fh.lo + Fl.hi + Fh.hi (+ Fl.lo with `products=4`), every product of
two f16 exact in f32, accumulated in f32.  `acc`: "seq" = one f32 rounding per term (pessimistic), "dot" = one rounding
per 16-term instruction (a fused dot product), Samples enter as s16 x window (no 1/32767: |x| <= 32768 stays inside
f16's range), the twiddles carry 2^-4 after stages 0 and 1.

usage: python tools/mfma_dft_model.py [episodes=3] [minutes=24] [products=3] [acc=seq|dot]
Prints the same tables as tools/f32_gate.py (error of every classifier input, err / S, share to recompute by K)."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.np_chromaprint import K, ORDER, STARTS  # noqa: E402

N = 4096
WIN = (0.54 - 0.46 * np.cos(np.arange(N) * 2.0 * np.pi / (N - 1)))          # f64; x 1/32767 at the end
W32 = WIN.astype(np.float32)
ang = -2.0 * np.pi * np.outer(np.arange(16), np.arange(16)) / 16.0           # [k][n]
C16, S16 = np.cos(ang), -np.sin(ang)                                          # W16^(nk) = C - i S
F64 = np.block([[C16, S16], [-S16, C16]])                                     # [yr; yi] = F [xr; xi]
F64[np.abs(F64) < 1e-15] = 0.0
FH = F64.astype(np.float16)
FL = (F64 - FH.astype(np.float64)).astype(np.float16)
TW = np.exp(-2j * np.pi * np.arange(N) / N)


def split(x32):
    hi = x32.astype(np.float16)
    lo = (x32 - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def stage(x, products, acc_mode):
    """x: complex64 [16, M] -> DFT over axis 0, [16, M] complex64, arithmetic as the matrix pipe's."""
    X = np.concatenate([x.real, x.imag]).astype(np.float32)                 # [32, M]
    hi, lo = split(X)
    terms = [(FH, lo), (FL, hi), (FH, hi)]
    if products == 4:
        terms.insert(0, (FL, lo))
    acc = np.zeros((32, X.shape[1]), dtype=np.float32)
    for A, B in terms:
        A32, B32 = A.astype(np.float32), B.astype(np.float32)
        if acc_mode == "seq":
            for k in range(32):
                acc = acc + A32[:, k:k + 1] * B32[k:k + 1, :]               # exact product, one f32 rounding per add
        else:
            for k0 in (0, 16):                                              # one instruction = 16 terms, one rounding
                acc = (acc.astype(np.float64) + A32[:, k0:k0 + 16].astype(np.float64) @ B32[k0:k0 + 16].astype(np.float64)).astype(np.float32)
    return (acc[:16] + 1j * acc[16:]).astype(np.complex64)


def cmul32(a, b):
    """complex64 product with f32 operations (4 mul + 2 add, no fma)."""
    ar, ai, br, bi = a.real.astype(np.float32), a.imag.astype(np.float32), b.real.astype(np.float32), b.imag.astype(np.float32)
    return ((ar * br - ai * bi) + 1j * (ar * bi + ai * br)).astype(np.complex64)


def fft_pairs(z, products=3, acc_mode="seq"):
    """z: complex64 [P, 4096] (unscaled) -> spectrum [P, 4096] complex64, scaled by 2^-8."""
    P = z.shape[0]
    x = z.reshape(P, 16, 16, 16).transpose(1, 0, 2, 3).reshape(16, -1)       # [n0, (P, n1, n2)]
    y = stage(x, products, acc_mode).reshape(16, P, 16, 16)                  # [k0, P, n1, n2]
    k0 = np.arange(16)[:, None, None, None]
    b = (16 * np.arange(16)[:, None] + np.arange(16)[None, :])[None, None]
    y = cmul32(y, (TW[(k0 * b) % N] * 0.0625).astype(np.complex64))
    x = y.transpose(2, 0, 1, 3).reshape(16, -1)                              # [n1, (k0, P, n2)]
    y = stage(x, products, acc_mode).reshape(16, 16, P, 16)                  # [k1, k0, P, n2]
    k1 = np.arange(16)[:, None, None, None]
    n2 = np.arange(16)[None, None, None, :]
    y = cmul32(y, (TW[(16 * k1 * n2) % N] * 0.0625).astype(np.complex64))
    x = y.transpose(3, 0, 1, 2).reshape(16, -1)                              # [n2, (k1, k0, P)]
    y = stage(x, products, acc_mode).reshape(16, 16, 16, P)                  # [k2, k1, k0, P]
    return y.transpose(3, 0, 1, 2).reshape(P, N)                             # k = 256 k2 + 16 k1 + k0


def chroma_mfma(pcm, products=3, acc_mode="seq", batch=256):
    """chroma [frames, 12] (f64 values of f32 sums) and the PAIR energy E [frames] from the modelled first pass."""
    pcm = np.asarray(pcm)
    n = len(pcm)
    nf = 0 if n < N else (n - N) // 1365 + 1
    chroma, energy = np.zeros((nf, 12)), np.zeros(nf)
    scale = np.float32((256.0 / 32767.0) ** 2)
    for f0 in range(0, nf, 2 * batch):
        f1 = min(nf, f0 + 2 * batch)
        fa = np.arange(f0, f1, 2)
        fb = np.minimum(fa + 1, nf - 1)
        has_b = (fa + 1) < nf
        idx = np.arange(N)[None, :]
        a = pcm[fa[:, None] * 1365 + idx].astype(np.float32) * W32
        b = pcm[fb[:, None] * 1365 + idx].astype(np.float32) * W32 * has_b[:, None].astype(np.float32)
        Z = fft_pairs((a + 1j * b).astype(np.complex64), products, acc_mode)
        Zc = np.conj(Z[:, (-np.arange(N)) % N])
        Xa = ((Z + Zc) * np.complex64(0.5)).astype(np.complex64)
        Xb = ((Z - Zc) * np.complex64(-0.5j)).astype(np.complex64)
        for X, fr, ok in ((Xa, fa, np.ones(len(fa), bool)), (Xb, fa + 1, has_b)):
            p = (X.real.astype(np.float32) ** 2 + X.imag.astype(np.float32) ** 2) * scale
            ch = np.add.reduceat(p[:, K[ORDER]], STARTS, axis=1)
            chroma[fr[ok]] = ch[ok].astype(np.float64)
        e = ((a.astype(np.float64) ** 2).sum(axis=1) + (b.astype(np.float64) ** 2).sum(axis=1)) * N / 32767.0 ** 2
        energy[fa] = e
        energy[(fa + 1)[has_b]] = e[has_b]
    return chroma, energy


def main():
    import tools.f32_gate as gate
    n_eps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
    products = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    acc_mode = sys.argv[4] if len(sys.argv) > 4 else "seq"
    which = sys.argv[5] if len(sys.argv) > 5 else "both"
    from needle_amd import synth

    def first_pass(pcm):
        return chroma_mfma(pcm, products, acc_mode)

    real_emu = gate.chroma_emu32

    def emu_pair_energy(pcm):                    # the product's f32 kernel, E of the frame PAIR as since round 5
        ch, e = real_emu(pcm)
        ep = e.copy()
        ep[0:len(e) - len(e) % 2:2] += e[1::2][: len(e) // 2]
        ep[1::2] = ep[0:len(e) - len(e) % 2:2][: len(e) // 2]
        return ch, ep

    print(f"split-f16 matrix-pipe model: products {products}, accumulation '{acc_mode}'; beside it the f32 kernel's own arithmetic (emu)")
    for label, fn in (("mfma", first_pass), ("emu32", emu_pair_energy)):
        if which not in ("both", label):
            continue
        gate.MODE = "emu"
        gate.chroma_emu32 = fn
        agg = {k: [0, 0, 0, 0, 0, 0] for k in gate.KS}
        worst = 0.0
        for k in range(n_eps):
            ep = synth.make_episode(k, minutes * 60.0, 90.0 if minutes >= 10 else 20.0, 60.0 if minutes >= 10 else 0.0)
            res = gate.analyse(ep.pcm[: len(ep.pcm) // 2])
            worst = max(worst, res["ratio_max"])
            for kk in gate.KS:
                for i, v in enumerate(res["ktable"][kk]):
                    agg[kk][i] += v
            print(f"[{label}] ep {k}: err max {res['err_max']:.3e} mean {res['err_mean']:.3e} flips {res['flips']} err/S max {res['ratio_max']:.3f} "
                  f"rel chroma err max {res['rel_chroma_err_max']:.2e}", flush=True)
        zworst = 0.0
        for name, pcm in gate.zoo().items():
            res = gate.analyse(pcm, step=1)
            zworst = max(zworst, res["ratio_max"])
            print(f"[{label}] zoo {name}: err max {res['err_max']:.3e} flips {res['flips']} err/S max {res['ratio_max']:.3f} "
                  f"S [{res['scale_min']:.2e}, {res['scale_max']:.2e}]", flush=True)
        print(f"[{label}] worst err / S: corpus {worst:.3f}, zoo {zworst:.3f} (S with u = 2^-24 and the PAIR's energy)")
        for kk in gate.KS:
            a = agg[kk]
            print(f"[{label}]   K {kk:4d}: items {a[0]}/{a[1]} = {a[0] / max(a[1], 1):.4%}  chunks (4 pairs) {a[4] / max(a[5], 1):.4%}")
        gate.chroma_emu32 = real_emu


if __name__ == "__main__":
    main()
