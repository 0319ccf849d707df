"""Label CSV producer: `prepare label` (reference scripts/processing/LabelDataGenerator.py:22-125).

For every 10 ms step of an organised WAV file that has VTR formants, the slope of the chosen formant over the
2*RADIUS+1 surrounding frames and how sure that slope is. A row is kept when the step is not inside a silent
phoneme and the Pearson p-value is below RISK:

    set,region,speaker,sentence,phoneme,timepoint,slope,p,sign      ->  trainingData/label_data.csv

Written from that definition, all steps of a file at once:
  * the frame windows are one strided view of the formant track (rows = steps, 2*RADIUS+1 columns);
  * the abscissae of a window are symmetric about the step (x_k = step + (k - RADIUS) * STEP), so the least-squares
    line has slope a = <y, d> / <d, d> with d = (k - RADIUS) * STEP and intercept b = mean(y) - a * step
    (the reference solves each 11 x 2 system with numpy.linalg.lstsq: same solution up to the last bits, which
    only matter on exact ties of the 5-decimal rounding - those rows go through lstsq here too);
  * the reference's p-value is scipy.stats.pearsonr(y, a x + b): one call over all rows (axis=1);
  * slope and p are rounded to 5 decimals with NumPy's rule (the reference calls round() on numpy.float64 scalars,
    which is numpy.round), sign = 1 for a > 0.
Host-side NumPy/SciPy; the windows x regression is a few thousand flops per second of audio.
"""
import csv
import glob
import os
import time
from configparser import ConfigParser

import numpy
from scipy.stats import pearsonr

from .FBFileReader import GetFormantFrequencies, frame_windows
from .GammatoneFiltering import GetArrayFromWAV
from .PHNFileReader import SILENTS, ExtractPhonemes

CSV_COLUMNS = ("set", "region", "speaker", "sentence", "phoneme", "timepoint", "slope", "p", "sign")


def phonemes_at(phonemes, timepoints):
    """Phoneme of every timepoint: the first segment (file order) with start <= t <= end, 'h#' when none
    (the lookup rule of PHNFileReader.GetPhonemeFromArrayAt, for an array of timepoints)."""
    timepoints = numpy.asarray(timepoints, dtype=numpy.int64)
    names = numpy.full(timepoints.shape, 'h#', dtype=object)
    todo = numpy.ones(timepoints.shape, dtype=bool)
    for name, start, end in phonemes or ():
        hit = todo & (timepoints >= start) & (timepoints <= end)
        names[hit] = name
        todo &= ~hit
    return names


def slopes_and_pvalues(windows, steps, step_len):
    """Least-squares slope of every window against x_k = step + (k - R) * step_len, and the p-value of
    pearsonr(window, fitted line)."""
    dots = windows.shape[1]
    d = (numpy.arange(dots) - dots // 2) * float(step_len)
    a = windows @ d / (d @ d)
    b = windows.mean(axis=1) - a * steps
    # Formant values carry two decimals and the abscissae are multiples of step_len, so a * 1e5 is a rational with a
    # small denominator: exact ties of the 5-decimal rounding are common (about one row in twenty) and are decided
    # by the last bits of the solver. Those rows are re-solved the way the reference solves every row.
    scaled = numpy.abs(a) * 1e5
    for i in numpy.flatnonzero(numpy.abs(scaled - numpy.floor(scaled) - 0.5) < 1e-4):
        x = steps[i] + d
        (a[i], b[i]), _, _, _ = numpy.linalg.lstsq(numpy.vstack([x, numpy.ones(dots)]).T, windows[i], rcond=None)
    fitted = a[:, None] * (steps[:, None] + d[None, :]) + b[:, None]
    p = pearsonr(windows, fitted, axis=1).pvalue if len(a) else numpy.zeros(0)
    return a, p


def ExtractLabel(wavFile, config):
    """Rows [set, region, speaker, sentence, phoneme, timepoint, slope, p, sign] of one file, or None."""
    base = os.path.splitext(wavFile)[0]
    radius = config.getint('CNN', 'RADIUS')
    risk = config.getfloat('CNN', 'RISK')
    period_us = config.getint('CNN', 'SAMPLING_PERIOD')
    track, _ = GetFormantFrequencies(base + '.FB', config.getint('CNN', 'FORMANT'))
    if track is None:
        return None
    phonemes = ExtractPhonemes(base + '.PHN')
    framerate, samples = GetArrayFromWAV(wavFile)

    samples_per_frame = framerate * period_us * (1.0 / 1000000)     # 160.0 at 16 kHz / 10 ms
    step_len = int(samples_per_frame)
    count = max(int(len(samples) / samples_per_frame - (2 * radius + 1) - 1), 0)
    steps = step_len * radius + step_len * numpy.arange(count, dtype=numpy.int64)

    names = phonemes_at(phonemes, steps)
    voiced = ~numpy.isin(names, SILENTS)
    steps, names = steps[voiced], names[voiced]
    windows = frame_windows(track, steps, radius, samples_per_frame)
    slope, p = slopes_and_pvalues(windows, steps.astype(numpy.float64), step_len)

    region, speaker, sentence = os.path.basename(base).split(".")
    subset = os.path.basename(os.path.dirname(base))
    keep = p < risk
    # (the sign column follows the slope as fitted, not its 5-decimal rounding: reference :70-75)
    rows = [[subset, region, speaker, sentence, str(ph), int(t), a, pv, 1 if a_raw > 0 else 0]
            for ph, t, a, pv, a_raw in zip(names[keep], steps[keep], numpy.round(slope[keep], 5), numpy.round(p[keep], 5),
                                           slope[keep])]
    return rows or None


def GenerateLabelData():
    """`prepare label`: every resources/f2cnn/*/*.WAV in sorted order -> trainingData/label_data.csv."""
    started = time.time()
    config = ConfigParser()
    config.read('configF2CNN.conf')
    wavs = sorted(glob.glob(os.path.join("resources", "f2cnn", "*", "*.WAV")))
    if not wavs:
        print("NO FILES FOUND")
        exit(-1)
    print("\n###############################\nGenerating Label Data from files in '{}' into 2 classes.".format(
        os.path.dirname(os.path.dirname(wavs[0]))))
    print(len(wavs), "files found")
    target = os.path.join("trainingData", "label_data.csv")
    os.makedirs(os.path.dirname(target), exist_ok=True)
    written = 0
    with open(target, "w") as out:
        writer = csv.writer(out, lineterminator='\n')
        for i, wav in enumerate(wavs):
            rows = ExtractLabel(wav, config) or []
            writer.writerows(rows)
            written += len(rows)
            print("{:>6}/{}\t{:<50}\t{} rows".format(i + 1, len(wavs), wav, len(rows)))
    print("Generated Label Data CSV of", written, "lines in '{}'.".format(target))
    print('                Total time:', time.time() - started)
    print('')
