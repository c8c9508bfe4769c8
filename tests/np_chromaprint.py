"""Independent numpy statement of chromaprint's default fingerprinter, used only by
the tests to cross-check the C oracle: numpy's FFT, cumulative sums for the integral image, no shared code."""
import numpy as np
import scipy.fft as sfft

CLASSIFIERS = [  # (type, y, height, width, t0, t1, t2)
    (0, 4, 3, 15, 1.98215, 2.35817, 2.63523), (4, 4, 6, 15, -1.03809, -0.651211, -0.282167),
    (1, 0, 4, 16, -0.298702, 0.119262, 0.558497), (3, 8, 2, 12, -0.105439, 0.0153946, 0.135898),
    (3, 4, 4, 8, -0.142891, 0.0258736, 0.200632), (4, 0, 3, 5, -0.826319, -0.590612, -0.368214),
    (1, 2, 2, 9, -0.557409, -0.233035, 0.0534525), (2, 7, 3, 4, -0.0646826, 0.00620476, 0.0784847),
    (2, 6, 2, 16, -0.192387, -0.029699, 0.215855), (2, 1, 3, 2, -0.0397818, -0.00568076, 0.0292026),
    (5, 10, 1, 15, -0.53823, -0.369934, -0.190235), (3, 6, 2, 10, -0.124877, 0.0296483, 0.139239),
    (2, 1, 1, 14, -0.101475, 0.0225617, 0.231971), (3, 5, 6, 4, -0.0799915, -0.00729616, 0.063262),
    (1, 9, 2, 12, -0.272556, 0.019424, 0.302559), (3, 4, 2, 14, -0.164292, -0.0321188, 0.0846339)]
GRAY = [0, 1, 3, 2]


def note_table():
    k = np.arange(10, 1308)
    freq = k * 11025.0 / 4096.0
    octave = np.log(freq / 27.5) / np.log(2.0)
    return k, (12 * (octave - np.floor(octave))).astype(int)


def chroma_features(pcm):
    pcm = np.asarray(pcm, dtype=np.float64)
    n = len(pcm)
    frames = 0 if n < 4096 else (n - 4096) // 1365 + 1
    w = (1.0 / 32767.0) * (0.54 - 0.46 * np.cos(np.arange(4096) * 2.0 * np.pi / 4095))
    k, note = note_table()
    chroma = np.zeros((frames, 12))
    for f in range(frames):
        spec = np.fft.rfft(pcm[f * 1365:f * 1365 + 4096] * w)
        power = spec.real ** 2 + spec.imag ** 2
        np.add.at(chroma[f], note, power[k])
    return chroma


def fingerprint(pcm):
    chroma = chroma_features(pcm)
    frames = len(chroma)
    if frames < 5:
        return np.zeros(0, dtype=np.uint32)
    coef = np.array([0.25, 0.75, 1.0, 0.75, 0.25])
    rows = frames - 4
    fir = sum(coef[j] * chroma[j:j + rows] for j in range(5))
    norm = np.sqrt((fir ** 2).sum(axis=1))
    feat = np.where(norm[:, None] < 0.01, 0.0, fir / np.where(norm[:, None] == 0, 1, norm[:, None]))
    items = []
    for x in range(rows - 15):
        win = feat[x:x + 16]
        integ = np.zeros((17, 13))
        integ[1:, 1:] = win.cumsum(axis=0).cumsum(axis=1)

        def area(r1, c1, r2, c2):
            return integ[r2, c2] - integ[r1, c2] - integ[r2, c1] + integ[r1, c1]
        bits = 0
        for (t, y, h, wd, t0, t1, t2) in CLASSIFIERS:
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
            v = np.log((1.0 + a) / (1.0 + b))
            q = (0 if v < t0 else 1) if v < t1 else (2 if v < t2 else 3)
            bits = (bits << 2) | GRAY[q]
        items.append(bits)
    return np.array(items, dtype=np.uint32)


# ---- vectorised forms over a whole stream, in f64 or f32 (tools/f32_gate.py, the adversarial certification tests) ----
U = 2.0 ** -24
W64 = (1.0 / 32767.0) * (0.54 - 0.46 * np.cos(np.arange(4096) * 2.0 * np.pi / 4095))
W32 = W64.astype(np.float32)
K, NOTE = note_table()
ORDER = np.argsort(NOTE, kind="stable")
STARTS = np.searchsorted(NOTE[ORDER], np.arange(12))


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


def classifier_values(chroma, energy=None):
    """log v for every raw item and classifier: [items, 16]; the feature norms [rows]; with `energy` [frames] also
    S [items] = max over the item's 16 rows of u sqrt(E_row / norm_row) (0 for rows under the 0.01 cut)."""
    coef = np.array([0.25, 0.75, 1.0, 0.75, 0.25])
    rows = len(chroma) - 4
    fir = sum(coef[j] * chroma[j:j + rows] for j in range(5))
    norm = np.sqrt((fir ** 2).sum(axis=1))
    scale = None
    if energy is not None:
        e_row = sum(coef[j] * energy[j:j + rows] for j in range(5))
        with np.errstate(divide="ignore", invalid="ignore"):
            sig = np.where(norm >= 0.01, U * np.sqrt(e_row / np.maximum(norm, 1e-300)), 0.0)
        items_ = rows - 15
        scale = sig[np.arange(items_)[:, None] + np.arange(16)[None, :]].max(axis=1)
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
    if energy is not None:
        return vals, norm, scale
    return vals, norm


