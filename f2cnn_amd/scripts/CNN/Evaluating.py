"""Drop-in for the hot-path part of the reference's ``scripts/CNN/Evaluating.py``: WAV -> filterbank ->
envelope -> every-sample 11xC windows -> normalise -> CNN -> rising/falling label per sample
(``EvaluateOneWavArray`` :42-87, ``EvaluateOneWavFile`` :116-135). Accuracy against VTR labels and the
plots (:89-113) need the TIMIT/VTR side files and are outside this path; the scores and labels are
returned and saved next to the WAV instead."""
import os

import numpy

from ... import _lib
from ...config import F2Config
from ...gammatone import filters
from ...model import F2CNNModel, load_model
from ..processing.EnvelopeExtraction import FFT_PRECISION
from ..processing.GammatoneFiltering import GetArrayFromWAV


def EvaluateOneWavArray(wavArray, framerate, wavFileName=None, model='last_trained_model', LPF=False, CUTOFF=100,
                        CENTER_FREQUENCIES=None, FILTERBANK_COEFFICIENTS=None, ctx=None, return_envelopes=False):
    """Returns (scores (nb,2) float32, labels (nb,) uint8 [, envelopes (C,N) float64]); nb = N - 11*STEP."""
    ctx = ctx or _lib.default_context()
    cfg = F2Config()
    if FILTERBANK_COEFFICIENTS is None:
        CENTER_FREQUENCIES = filters.centre_freqs(framerate, cfg.nchannels, cfg.low_freq)
        FILTERBANK_COEFFICIENTS = filters.make_erb_filters(framerate, CENTER_FREQUENCIES)
    coefs = numpy.ascontiguousarray(FILTERBANK_COEFFICIENTS, dtype=numpy.float64)
    Cn = coefs.shape[0]
    if not isinstance(model, F2CNNModel):
        model = load_model(model)
    print("Applying filterbank...")
    if not LPF:
        print("Extracting Envelope...")
    else:
        print("Extraction Envelope with {}Hz Low Pass Filter...".format(CUTOFF))
    wave, dt = filters._wave_args(wavArray)
    N = wave.shape[0]
    STEP = int(framerate * cfg.sampling_period * (1 / 1000000.))
    nb = max(int(N - cfg.dots_per_input * STEP), 0)
    print("Generating input data for CNN...")
    print("INPUT SHAPE:", (nb, cfg.dots_per_input, Cn))
    scores = numpy.empty((nb, 2), numpy.float32)
    labels = numpy.empty(nb, numpy.uint8)
    env = numpy.empty((Cn, N), numpy.float64) if return_envelopes else None
    print("Evaluating the data with the pretrained model...")
    try:
        got = ctx.eval_utterance(model.handle(ctx), wave, dt, N, coefs, Cn, bool(LPF), CUTOFF if LPF else 0.0,
                                 FFT_PRECISION, cfg.radius, STEP, env, scores, labels, _lib.MEM_HOST)
    except _lib.F2Error as e:
        if e.code == _lib.F2_ERR_NONPOSITIVE:
            raise ValueError("values must all be positive")
        raise
    assert got == nb
    return (scores, labels, env) if return_envelopes else (scores, labels)


def EvaluateOneWavFile(file, LPF=False, CUTOFF=50, model='last_trained_model', CENTER_FREQUENCIES=None,
                       FILTERBANK_COEFFICIENTS=None):
    """`cnn eval --file X.WAV`: writes <base>.F2CNN.npz (scores, labels) and returns (scores, labels)."""
    print('Using model', model if not isinstance(model, F2CNNModel) else '<in-memory model>')
    print("File:\t\t{}".format(file))
    framerate, wavArray = GetArrayFromWAV(file)
    scores, labels = EvaluateOneWavArray(wavArray, framerate, file, model=model, LPF=LPF, CUTOFF=CUTOFF,
                                         CENTER_FREQUENCIES=CENTER_FREQUENCIES,
                                         FILTERBANK_COEFFICIENTS=FILTERBANK_COEFFICIENTS)
    out = os.path.splitext(file)[0] + '.F2CNN.npz'
    numpy.savez(out, scores=scores, labels=labels)
    rising = int(labels.sum())
    print("\t\t{}\tdone ! {} windows: {} rising, {} falling -> {}".format(file, len(labels), rising,
                                                                          len(labels) - rising, out))
    return scores, labels


# ---- batch / noise evaluation (reference scripts/CNN/Evaluating.py:138-221) -------------
def EvaluateWavArrays(wavArrays, framerate, model='last_trained_model', LPF=False, CUTOFF=100,
                      FILTERBANK_COEFFICIENTS=None, ctx=None):
    """EvaluateOneWavArray for a list of utterances of one sample type in one device pass (f2_eval_batch): the
    filterbank and envelope kernels see the whole batch, windows and CNN run utterance by utterance.
    Returns a list of (scores (nb,2) float32, labels (nb,) uint8)."""
    ctx = ctx or _lib.default_context()
    cfg = F2Config()
    if FILTERBANK_COEFFICIENTS is None:
        FILTERBANK_COEFFICIENTS = filters.make_erb_filters(framerate, filters.centre_freqs(framerate, cfg.nchannels,
                                                                                           cfg.low_freq))
    coefs = numpy.ascontiguousarray(FILTERBANK_COEFFICIENTS, dtype=numpy.float64)
    Cn = coefs.shape[0]
    if not isinstance(model, F2CNNModel):
        model = load_model(model)
    if not len(wavArrays):
        return []
    waves, dts = zip(*[filters._wave_args(w) for w in wavArrays])
    if len(set(dts)) != 1:
        raise ValueError("the utterances of one batch must share a sample type (int16 or float64)")
    offsets = numpy.zeros(len(waves) + 1, numpy.int64)
    offsets[1:] = numpy.cumsum([w.shape[0] for w in waves])
    flat = numpy.concatenate(waves)
    STEP = int(framerate * cfg.sampling_period * (1 / 1000000.))
    nbs = [max(int(w.shape[0] - cfg.dots_per_input * STEP), 0) for w in waves]
    scores = numpy.empty((sum(nbs), 2), numpy.float32)
    labels = numpy.empty(sum(nbs), numpy.uint8)
    try:
        ctx.eval_batch(model.handle(ctx), flat, dts[0], offsets, coefs, len(waves), Cn, bool(LPF), CUTOFF if LPF else 0.0,
                       FFT_PRECISION, cfg.radius, STEP, scores, labels, _lib.MEM_HOST)
    except _lib.F2Error as e:
        if e.code == _lib.F2_ERR_NONPOSITIVE:
            raise ValueError("values must all be positive")
        raise
    out, pos = [], 0
    for nb in nbs:
        out.append((scores[pos:pos + nb], labels[pos:pos + nb]))
        pos += nb
    return out


def EvaluateRandom(count=None, LPF=False, CUTOFF=50, model='last_trained_model'):
    """`cnn evalrand`: evaluate the WAV files under resources/f2cnn/*/ in random order (all of them, or `count`
    drawn with replacement like numpy.random.choice in the reference). The filterbank is designed once and the model
    is uploaded once (the reference reloads the Keras model for every file)."""
    import glob
    import time
    TotalTime = time.time()
    wavFiles = sorted(glob.glob(os.path.join('resources', 'f2cnn', '*', '*.WAV')))
    if not wavFiles:
        print("NO WAV FILES FOUND")
        exit(-1)
    print("\n###############################\nEvaluating network on {} WAV files in '{}'.".format(
        len(wavFiles), os.path.split(wavFiles[0])[0]))
    cfg = F2Config()
    CENTER_FREQUENCIES = filters.centre_freqs(cfg.framerate, cfg.nchannels, cfg.low_freq)
    FILTERBANK_COEFFICIENTS = filters.make_erb_filters(cfg.framerate, CENTER_FREQUENCIES)
    if not isinstance(model, F2CNNModel):
        model = load_model(model)
    if count is None:
        wavFiles = list(numpy.random.permutation(wavFiles))
    elif count > 1:
        wavFiles = list(numpy.random.choice(wavFiles, count))
    results = {}
    BATCH = 16                                    # files per device pass
    for s0 in range(0, len(wavFiles), BATCH):
        group = wavFiles[s0:s0 + BATCH]
        loaded = [GetArrayFromWAV(file) for file in group]
        rates = {fr for fr, _ in loaded}
        if len(rates) == 1 and len({numpy.asarray(w).dtype for _, w in loaded}) == 1:
            outs = EvaluateWavArrays([w for _, w in loaded], loaded[0][0], model=model, LPF=LPF, CUTOFF=CUTOFF,
                                     FILTERBANK_COEFFICIENTS=FILTERBANK_COEFFICIENTS)
        else:                                     # mixed files: one at a time
            outs = [EvaluateOneWavArray(w, fr, file, model=model, LPF=LPF, CUTOFF=CUTOFF,
                                        FILTERBANK_COEFFICIENTS=FILTERBANK_COEFFICIENTS if fr == cfg.framerate else None)
                    for file, (fr, w) in zip(group, loaded)]
        for file, (scores, labels) in zip(group, outs):
            out = os.path.splitext(file)[0] + '.F2CNN.npz'
            numpy.savez(out, scores=scores, labels=labels)
            rising = int(labels.sum())
            print("\t\t{}\tdone ! {} windows: {} rising, {} falling -> {}".format(file, len(labels), rising,
                                                                                  len(labels) - rising, out))
            results[file] = (scores, labels)
    print("Evaluating network on all files.")
    print('              Total time:', time.time() - TotalTime)
    print('')
    return results


def SNRdbToSNRlinear(SNRdb):
    return 10 ** (SNRdb / 10.0)


def RMS(signal):
    """Root mean square of a signal (computed in float64: int16 squares would overflow)."""
    return numpy.sqrt(numpy.mean(numpy.square(numpy.asarray(signal, dtype=numpy.float64))))


def _noisy_copy_paths(file, SNRdB):
    """Where `cnn evalnoise` puts its outputs (reference layout, Evaluating.py:203-206): OutputWavFiles/addedNoise/<stem><SNR>dB.*"""
    stem = os.path.basename(os.path.splitext(file)[0])
    target = os.path.join('OutputWavFiles', 'addedNoise', '{}{}dB'.format(stem, SNRdB))
    return os.path.splitext(file)[0], target


def add_gaussian_noise(wave, SNRdB, rng=None):
    """wave + N(0, sigma^2), sigma = RMS(wave) / 10^(SNRdB / 10) - the reference's scaling (Evaluating.py:199), which divides
    by the POWER ratio where an amplitude ratio would be 10^(SNRdB / 20); reproduced, not corrected. float64 out."""
    sigma = RMS(wave) / SNRdbToSNRlinear(SNRdB)
    draw = (rng or numpy.random).normal
    return numpy.asarray(wave, dtype=numpy.float64) + draw(scale=sigma, size=len(wave))


def EvaluateWithNoise(file, LPF=False, CUTOFF=100, model='last_trained_model', CENTER_FREQUENCIES=None,
                      FILTERBANK_COEFFICIENTS=None, SNRdB=-3, rng=None):
    """`cnn evalnoise` (reference: scripts/CNN/Evaluating.py:193-221): the file plus Gaussian noise at the requested level is
    written next to copies of its annotation files under OutputWavFiles/addedNoise/ and the float64 waveform is evaluated by
    the device pipeline. Returns (scores, labels) and also leaves them in <target>.F2CNN.npz; `rng` (a numpy Generator or
    RandomState) makes the noise reproducible - the reference draws from the global numpy state."""
    import shutil
    from scipy.io import wavfile
    print("File:\t\t{}".format(file))
    print("Appyling gaussian noise, new SNR is {SNR}dB".format(SNR=SNRdB))      # (the reference's wording, kept for log parsers)
    framerate, clean = GetArrayFromWAV(file)
    noisy = add_gaussian_noise(clean, SNRdB, rng)
    source, target = _noisy_copy_paths(file, SNRdB)
    os.makedirs(os.path.dirname(target), exist_ok=True)
    wavfile.write(target + '.WAV', framerate, noisy)
    # annotation files travel with the audio where they exist (the reference gives up on all three at the first missing one)
    for ext in ('.FB', '.PHN', '.WRD'):
        if os.path.exists(source + ext):
            shutil.copyfile(source + ext, target + ext)
    print('New noisy WAVE file saved as', target + '.WAV')
    scores, labels = EvaluateOneWavArray(noisy, framerate, target + '.WAV', model=model, LPF=LPF, CUTOFF=CUTOFF,
                                         CENTER_FREQUENCIES=CENTER_FREQUENCIES,
                                         FILTERBANK_COEFFICIENTS=FILTERBANK_COEFFICIENTS)
    numpy.savez(target + '.F2CNN.npz', scores=scores, labels=labels)
    print("\t\t{}\tdone !".format(file))
    return scores, labels
