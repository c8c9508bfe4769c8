"""GPU tests (-m gpu) of the matrix-core resampler (needle_amd/csrc/resample_mfma.h): the kernel the row-layout rates
(48, 32, 24, 16, 8 kHz ...) take.  Bit-exact against oracle/ora_resample.c like every resampler kernel; what is specific
here: a workgroup is persistent and walks over many tiles (NEEDLE_HIP_RESAMPLE_GRID makes three workgroups do the work of
hundreds, so that the steady state -- a tile written to LDS while the next one's loads replace it in the registers, rows
longer than the staging threads' own share, the tail row -- is what the short test inputs exercise), tiles at either end
of a stream take the sample-by-sample path, streams of very different lengths share a launch, and the result must not depend
on the wave layout or on which kernel ran."""
import numpy as np
import pytest

from needle_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert capi.device_count() > 0, "GPU tests need a HIP device (the product has no CPU fallback)"


def _signal(n, ch, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 48000.0
    x = 9000 * np.sin(2 * np.pi * (220 + 35 * seed) * t) + 5000 * rng.standard_normal(n)
    x = np.clip(x, -32768, 32767).astype(np.int16)
    if ch == 2:
        y = np.clip(x.astype(np.int32) // 3 + rng.integers(-2000, 2000, n), -32768, 32767).astype(np.int16)
        x = np.stack([x, y], axis=1).reshape(-1)          # L + R odd about half the time: the truncating down-mix matters
    return x


@pytest.mark.parametrize("grid", [None, "3", "1"])
@pytest.mark.parametrize("rate,ch", [(48000, 2), (48000, 1), (32000, 2), (16000, 1), (8000, 2), (24000, 1)])
def test_persistent_workgroups_equal_the_oracle(rate, ch, grid, monkeypatch):
    if grid is None:
        monkeypatch.delenv("NEEDLE_HIP_RESAMPLE_GRID", raising=False)
    else:
        monkeypatch.setenv("NEEDLE_HIP_RESAMPLE_GRID", grid)
    # lengths: empty, shorter than one window, one tile and a bit, many tiles, a length that ends a tile exactly
    g = np.gcd(11025, rate)
    L, M = 11025 // g, rate // g
    exact = 16 * M * 5                                          # five whole tiles of sixteen rows
    pcms = [_signal(n, ch, k) for k, n in enumerate((0, 37, 16 * M + 991, rate * 2 + 123, exact, rate // 3))]
    got = capi.resample(pcms, ch, rate)
    for k, (a, p) in enumerate(zip(got, pcms)):
        assert a.tolist() == O.resample(p, ch, rate).tolist(), (rate, ch, grid, k)


def test_wave_layouts_and_the_dpp_kernel_agree(monkeypatch):
    pcm = [_signal(48000 * 3 + 5, 2, 7), _signal(48000 + 77, 2, 8)]
    want = [O.resample(p, 2, 48000).tolist() for p in pcm]
    for env in ({}, {"NEEDLE_HIP_RESAMPLE_LAYOUT": "0"}, {"NEEDLE_HIP_RESAMPLE_QUAD": "1"},
                {"NEEDLE_HIP_RESAMPLE_GRID": "2", "NEEDLE_HIP_RESAMPLE_LAYOUT": "0"}):
        for k in ("NEEDLE_HIP_RESAMPLE_LAYOUT", "NEEDLE_HIP_RESAMPLE_QUAD", "NEEDLE_HIP_RESAMPLE_GRID"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert [a.tolist() for a in capi.resample(pcm, 2, 48000)] == want, env


def test_full_scale_and_alternating_sign_inputs_clamp_and_truncate_alike(monkeypatch):
    monkeypatch.setenv("NEEDLE_HIP_RESAMPLE_GRID", "2")
    n = 48000 * 2
    sq = np.where((np.arange(n) // 11) % 2 == 0, 32767, -32768).astype(np.int16)
    odd = np.stack([np.full(n, -32768, np.int16), np.full(n, 32767, np.int16)], axis=1).reshape(-1)   # L + R = -1 everywhere
    neg = np.stack([(-(np.arange(n) % 7) - 1).astype(np.int16), np.zeros(n, np.int16)], axis=1).reshape(-1)
    assert capi.resample([sq], 1, 48000)[0].tolist() == O.resample(sq, 1, 48000).tolist()
    for p in (odd, neg):
        assert capi.resample([p], 2, 48000)[0].tolist() == O.resample(p, 2, 48000).tolist()
