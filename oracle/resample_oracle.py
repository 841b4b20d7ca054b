"""CPU restatement (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline may import
this) of the resampling step of the reference's evaluation sweep:

    x = librosa.resample(x, fsog, fs, res_type='kaiser_fast', scale=True)      Code/pceval.py:74

librosa 0.8.0 (environment.yml:82) hands this to resampy 0.2.2 (environment.yml:148) and divides the
result by sqrt(ratio) (``scale=True``).  Neither package is vendored in /root/reference nor installed
here, and the reference holds no fixture of a resampled signal: **parity unpinned** (SURVEY.md 8c).
What is restated is resampy's published algorithm - J. O. Smith's band-limited interpolation: a
Kaiser-windowed sinc sampled ``2**precision`` times per zero crossing, linearly interpolated between
table entries, both wings summed per output sample - with the ``kaiser_fast`` design parameters as the
resampy documentation gives them (16 zero crossings, precision 9, roll-off 0.85, Kaiser beta
8.555504641634386).  The shipped ``kaiser_fast`` table was produced by an optimisation run of the
authors; these are its published inputs, not a byte copy of the table.

Sanity (tests/test_resample.py): against scipy.signal.resample_poly (another windowed-sinc design:
agreement to ~1e-2 in the pass band, not a parity claim) and on sinusoids below the new Nyquist
frequency (amplitude gain 1 / sqrt(ratio) -> after librosa's scaling).
"""
import numpy as np

KAISER_FAST = dict(num_zeros=16, precision=9, rolloff=0.85, beta=8.555504641634386)


def sinc_window(num_zeros=16, precision=9, rolloff=0.85, beta=8.555504641634386):
    """Right wing of the interpolation filter (resampy.filters.sinc_window): rolloff * sinc(rolloff t)
    on t in [0, num_zeros], 2**precision samples per zero crossing, times the right half of a Kaiser
    window."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return (taper * sinc_win).astype(np.float64), num_bits


def filter_tables(ratio, **design):
    """(win, delta, num_table): the table the kernel / the loop below index, scaled for down-sampling
    (resampy.core.resample: ``interp_win *= sample_ratio`` when sample_ratio < 1)."""
    d = dict(KAISER_FAST)
    d.update(design)
    win, num_table = sinc_window(**d)
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return win, delta, num_table


def output_length(n_in, ratio):
    """librosa.resample(fix=True): ceil(n * ratio) samples (resampy itself yields int(n * ratio))."""
    return int(np.ceil(n_in * ratio))


def resample(x, fs_old, fs_new, scale=True, **design):
    """float64 restatement of librosa.resample(x, fs_old, fs_new, 'kaiser_fast', fix=True, scale=...)."""
    x = np.asarray(x, dtype=np.float64)
    ratio = float(fs_new) / float(fs_old)
    if ratio == 1.0:
        return x.copy()
    win, delta, num_table = filter_tables(ratio, **design)
    n_in = x.shape[0]
    n_res = int(n_in * ratio)
    sc = min(1.0, ratio)
    index_step = int(sc * num_table)
    nwin = win.shape[0]
    y = np.zeros(n_res)
    for t in range(n_res):                     # resampy.interpn.resample_f
        time_register = t / ratio
        n = int(time_register)
        frac = sc * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        if i_max > 0:
            k = offset + np.arange(i_max) * index_step
            y[t] += np.dot(win[k] + eta * delta[k], x[n - np.arange(i_max)])
        frac = sc - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_in - n - 1, (nwin - offset) // index_step)
        if k_max > 0:
            k = offset + np.arange(k_max) * index_step
            y[t] += np.dot(win[k] + eta * delta[k], x[n + 1 + np.arange(k_max)])
    n_out = output_length(n_in, ratio)         # librosa.util.fix_length
    if n_out > n_res:
        y = np.concatenate([y, np.zeros(n_out - n_res)])
    else:
        y = y[:n_out]
    if scale:
        y = y / np.sqrt(ratio)
    return y
