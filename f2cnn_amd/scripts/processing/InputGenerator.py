"""Drop-in for the reference's ``scripts/processing/InputGenerator.py``: labelled timepoints ->
``input_data*.npy`` ((Ntp, 11, C) float32), gathered by HIP kernel K3 instead of Python list comprehensions."""
import csv
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy

from ... import _lib
from ...config import F2Config


def GetListOfEnvelopeFilesAndTimepoints(labelFilename):
    """dict {"<TEST|TRAIN>/<region>.<speaker>.<sentence>.ENV1.npy": [timepoints in CSV order]} (reference :9-25)."""
    output = dict()
    with open(labelFilename, 'r') as labelFile:
        for row in csv.reader(labelFile):
            if not row:
                continue
            testOrTrain, region, speaker, sentence, _phoneme, timepoint = row[:6]
            key = os.path.join(testOrTrain, '.'.join((region, speaker, sentence, 'ENV1.npy')))
            output.setdefault(key, []).append(int(timepoint))
    return output


def gather_windows(envelopes, centers, radius, step, normalize=False, ctx=None):
    """(len(centers), 2*radius+1, C) float32 windows of one (C,N) envelope matrix."""
    ctx = ctx or _lib.default_context()
    env = numpy.ascontiguousarray(envelopes, dtype=numpy.float64)
    centers = numpy.ascontiguousarray(centers, dtype=numpy.int64)
    Cn, N = env.shape
    out = numpy.empty((centers.shape[0], 2 * radius + 1, Cn), numpy.float32)
    ctx.gather_windows(env, Cn, N, centers, centers.shape[0], radius, step, normalize, out, _lib.MEM_HOST)
    return out


def GenerateInputData(labelFile=None, inputFile=None, LPF=False, CUTOFF=100):
    """`prepare input` (reference :28-93): one (2*RADIUS+1, C) window per labelled timepoint -> input_data*.npy.

    Row order is part of the file format: envelope files in sorted path order, timepoints of a file in CSV order
    (`cnn train` pairs row i of this matrix with CSV line i, which holds when the CSV itself is path-sorted, as
    `prepare label` writes it). The rows of every file are therefore known before anything is read: each file's block
    is gathered by K3 straight into its place, while reader threads fetch the next envelope files."""
    started = time.time()
    if not os.path.isdir("trainingData"):
        print("LABEL GENERATION SHOULD BE DONE PRIOR TO INPUT...")
        exit(-1)
    label_csv = labelFile or os.path.join("trainingData", "label_data.csv")
    per_file = GetListOfEnvelopeFilesAndTimepoints(label_csv)
    print("\n###############################\nGenerating Input Data from files with '{}'.".format(label_csv))
    print("Using Low Pass Filtering with a cutoff at {}Hz".format(CUTOFF) if LPF else "Not using Low Pass Filtering")
    if not per_file:
        print("NO ENV1.npy FILES FOUND, PLEASE GENERATE ENVELOPES")
        exit(-1)
    order = sorted(per_file)
    first_row = numpy.concatenate([[0], numpy.cumsum([len(per_file[name]) for name in order])])
    print(len(order), "files found along with their", int(first_row[-1]), "entry timepoints.")
    cfg = F2Config()
    windows = numpy.empty((int(first_row[-1]), cfg.dots_per_input, cfg.nchannels), dtype=numpy.float32)
    print("Output shape:", windows.shape)

    def fetch(name):
        return numpy.load(os.path.join('resources', 'f2cnn', name))
    with ThreadPoolExecutor(4) as readers:
        ahead = [readers.submit(fetch, name) for name in order[:3]]
        for k, name in enumerate(order):
            envelopes = ahead.pop(0).result()
            if k + 3 < len(order):
                ahead.append(readers.submit(fetch, order[k + 3]))
            windows[first_row[k]:first_row[k + 1]] = gather_windows(envelopes, per_file[name], cfg.radius, cfg.step)
            print("\t\t{:<50} done !  {}/{} Files".format(os.path.join('resources', 'f2cnn', name), k + 1, len(order)))
    print('Generated Input Matrix of shape {}.'.format(windows.shape))
    target = inputFile or os.path.join('trainingData',
                                       'input_data_LPF{}.npy'.format(CUTOFF) if LPF else 'input_data_NOLPF.npy')
    print("Saving as {}...".format(target))
    os.makedirs(os.path.dirname(target) or '.', exist_ok=True)
    numpy.save(target, windows)
    numpy.save(os.path.join('trainingData', 'last_input_data.npy'), windows)    # the reference's backup copy (:90)
    print('                Total time:', time.time() - started)
    print('')
