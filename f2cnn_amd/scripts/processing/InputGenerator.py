"""Drop-in for the reference's ``scripts/processing/InputGenerator.py``: labelled timepoints ->
``input_data*.npy`` ((Ntp, 11, C) float32), gathered by HIP kernel K3 instead of Python list comprehensions."""
import csv
import os
import time

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
    """`prepare input` (reference :28-93): entries ordered by sorted(file) then CSV order within a file."""
    TotalTime = time.time()
    if not os.path.isdir("trainingData"):
        print("LABEL GENERATION SHOULD BE DONE PRIOR TO INPUT...")
        exit(-1)
    csvFilename = labelFile or os.path.join("trainingData", "label_data.csv")
    filesAndTimepointsDict = GetListOfEnvelopeFilesAndTimepoints(csvFilename)
    print("\n###############################\nGenerating Input Data from files with '{}'.".format(csvFilename))
    if LPF:
        print("Using Low Pass Filtering with a cutoff at {}Hz".format(CUTOFF))
    else:
        print("Not using Low Pass Filtering")
    if not filesAndTimepointsDict:
        print("NO ENV1.npy FILES FOUND, PLEASE GENERATE ENVELOPES")
        exit(-1)
    files = sorted(filesAndTimepointsDict.keys())
    totalTimePoints = sum(len(v) for v in filesAndTimepointsDict.values())
    print(len(files), "files found along with their", totalTimePoints, "entry timepoints.")
    cfg = F2Config()
    inputData = numpy.zeros((totalTimePoints, cfg.dots_per_input, cfg.nchannels), dtype=numpy.float32)
    print("Output shape:", inputData.shape)
    currentEntry = 0
    for currentFileIndex, file in enumerate(files):
        timepoints = filesAndTimepointsDict[file]
        path = os.path.join('resources', 'f2cnn', file)
        print("Reading:\t{}".format(path))
        envelopes = numpy.load(path)
        block = gather_windows(envelopes, timepoints, cfg.radius, cfg.step)
        inputData[currentEntry:currentEntry + len(timepoints)] = block
        currentEntry += len(timepoints)
        print("\t\t{:<50} done !  {}/{} Files".format(path, currentFileIndex + 1, len(files)))
    print('Generated Input Matrix of shape {}.'.format(inputData.shape))
    savePath = inputFile or os.path.join(
        'trainingData', 'input_data_LPF{}.npy'.format(CUTOFF) if LPF else 'input_data_NOLPF.npy')
    print("Saving as {}...".format(savePath))
    os.makedirs(os.path.split(savePath)[0] or '.', exist_ok=True)
    numpy.save(savePath, inputData)
    numpy.save(os.path.join('trainingData', 'last_input_data.npy'), inputData)
    print('                Total time:', time.time() - TotalTime)
    print('')
