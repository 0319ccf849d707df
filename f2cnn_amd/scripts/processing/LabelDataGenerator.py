"""Label CSV producer (reference scripts/processing/LabelDataGenerator.py): for every
10 ms step of every organised WAV file with VTR formants, the least-squares slope of F2 over the 11 surrounding
frames, the p-value of its Pearson correlation, and the rising/falling class when p < RISK. Output columns:
set,region,speaker,sentence,phoneme,timepoint,slope,p,sign  ->  trainingData/label_data.csv (consumed by
`prepare input`). Host-side NumPy/SciPy: a few hundred 11-point regressions per second of audio."""
import csv
import glob
import os
import time
from configparser import ConfigParser

import numpy
from scipy.stats import pearsonr

from .FBFileReader import GetFormantFrequencies, GetFromantFrequenciesAround
from .GammatoneFiltering import GetArrayFromWAV
from .PHNFileReader import SILENTS, ExtractPhonemes, GetPhonemeFromArrayAt


def ExtractLabel(wavFile, config):
    """Rows [set, region, speaker, sentence, phoneme, timepoint, slope, p, sign] of one file, or None."""
    fileBase = os.path.splitext(wavFile)[0]
    RADIUS = config.getint('CNN', 'RADIUS')
    RISK = config.getfloat('CNN', 'RISK')
    FORMANT = config.getint('CNN', 'FORMANT')
    SAMPPERIOD = config.getint('CNN', 'SAMPLING_PERIOD')
    DOTSPERINPUT = RADIUS * 2 + 1
    USTOS = 1.0 / 1000000

    FormantArray, _ = GetFormantFrequencies(fileBase + '.FB', FORMANT)
    if FormantArray is None:
        return None
    phonemes = ExtractPhonemes(fileBase + '.PHN')
    framerate, wavList = GetArrayFromWAV(wavFile)
    wavToFormant = framerate * SAMPPERIOD * USTOS
    nb = int(len(wavList) / wavToFormant - DOTSPERINPUT - 1)
    region, speaker, sentence = os.path.split(fileBase)[1].split(".")
    testOrTrain = os.path.split(os.path.split(fileBase)[0])[1]
    STEP = int(framerate * SAMPPERIOD * USTOS)
    START = int(STEP * RADIUS)
    offsets = numpy.array([(k - RADIUS) * STEP for k in range(DOTSPERINPUT)])

    output = []
    for step in (START + k * STEP for k in range(nb)):
        phoneme = GetPhonemeFromArrayAt(phonemes, step)
        if phoneme in SILENTS:
            continue
        FormantValues = numpy.array(GetFromantFrequenciesAround(FormantArray, step, RADIUS, wavToFormant))
        x = step + offsets
        A = numpy.vstack([x, numpy.ones(len(x))]).T
        [a, b], _, _, _ = numpy.linalg.lstsq(A, FormantValues, rcond=None)
        _r, p = pearsonr(FormantValues, a * x + b)
        if p < RISK:
            output.append([testOrTrain, region, speaker, sentence, phoneme, step, round(a, 5), round(p, 5),
                           1 if a > 0 else 0])
    return output if len(output) > 0 else None


def GenerateLabelData():
    """`prepare label`: every resources/f2cnn/*/*.WAV (sorted) -> trainingData/label_data.csv."""
    TotalTime = time.time()
    config = ConfigParser()
    config.read('configF2CNN.conf')
    filenames = sorted(glob.glob(os.path.join("resources", "f2cnn", "*", "*.WAV")))
    if not filenames:
        print("NO FILES FOUND")
        exit(-1)
    print("\n###############################\nGenerating Label Data from files in '{}' into 2 classes.".format(
        os.path.split(os.path.split(filenames[0])[0])[0]))
    print(len(filenames), "files found")
    csvLines = []
    for i, file in enumerate(filenames):
        print("Reading:\t{:<50}\t{}/{}".format(file, i, len(filenames)))
        fileEntry = ExtractLabel(file, config)
        if fileEntry is not None:
            csvLines.extend(fileEntry)
        print("\t\t{:<50}\tdone !".format(file))
    filePath = os.path.join("trainingData", "label_data.csv")
    print("Saving {} lines in '{}'.".format(len(csvLines), filePath))
    os.makedirs(os.path.split(filePath)[0], exist_ok=True)
    with open(filePath, "w") as outputFile:
        writer = csv.writer(outputFile, lineterminator='\n')
        for line in csvLines:
            writer.writerow(line)
    print("Generated Label Data CSV of", len(csvLines), "lines.")
    print('                Total time:', time.time() - TotalTime)
    print('')
