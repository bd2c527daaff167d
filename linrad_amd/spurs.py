"""Control-plane tables of the spur subtraction that depend only on the configuration: the reference line shapes lrh_spur_config takes.

init_spur_spectra (spursub.c:824-940) transforms, for 256 fractional frequencies f = 3 + i/256 bins, a windowed complex exponential
of 256 points with the window in front of the spectra the spurs live in (the fft2 window, or the fft1 window with the second fft off),
turns each result to phase zero by its own alternating-sign power-weighted sum, keeps the real parts of the first eight bins and
normalises them to unit sum of squares.  The table depends on the window's sine power only -- not on the transform size."""
import numpy as np

SPUR_SIZE = 8
NO_OF_SPUR_SPECTRA = 256


def _window(size, sinpow):
    """make_window mode 4 (fft0.c:812-905): sin^n over the first half, unit mean square, mirrored; float32 like the reference"""
    if sinpow == 0:
        return np.ones(size, np.float32)
    if sinpow > 7:
        raise ValueError("line shapes for the Gaussian / erfc windows are not built here")
    half = np.sin(np.arange(size // 2 + 1) * (np.pi / size)) ** float(sinpow)
    half = half.astype(np.float32)
    sumsq = np.sum(half.astype(np.float64) ** 2)
    half = (half * np.float32(1 / np.sqrt(2 * sumsq / size))).astype(np.float32)
    win = np.empty(size, np.float32)
    win[:size // 2 + 1] = half
    win[size // 2 + 1:] = half[1:size // 2][::-1]
    return win


def spur_spectra(sinpow=2):
    """float32 [256 * 8]: the table lrh_spur_config / lro_spur_config take"""
    n = 256
    win = _window(n, sinpow).astype(np.float64)
    out = np.empty(NO_OF_SPUR_SPECTRA * SPUR_SIZE, np.float32)
    sign = np.where(np.arange(SPUR_SIZE) % 2 == 0, 1.0, -1.0)
    for i in range(NO_OF_SPUR_SPECTRA):
        step = 2 * np.pi * (1 + SPUR_SIZE / 4 + i / NO_OF_SPUR_SPECTRA) / n
        ph = step * np.arange(n)
        z = win * (np.sin(ph) + 1j * np.cos(ph))
        spec = (np.fft.ifft(z) * n)[:SPUR_SIZE]          # fftforward (fft0.c:481ff): e^{+j} kernel, natural order -- the line sits at bin +3
        wgt = sign * np.abs(spec) ** 2
        d = np.sum(wgt * spec)
        d /= abs(d)
        re = (spec * np.conj(d)).real
        out[i * SPUR_SIZE:(i + 1) * SPUR_SIZE] = re / np.sqrt(np.sum(re * re))
    return out
