"""Drop-in for the reference's ``scripts/processing/GammatoneFiltering.py``: WAV -> gammatone filterbank
-> ``<base>.GFB.npy`` (NPY v1.0, '<f8', C-order (C,N), row 0 = highest centre frequency).

The per-file process pool of the reference (:122-125) becomes batches of files per GPU launch; with several
ranks (torchrun or F2CNN_RANK/F2CNN_WORLD) rank r handles files[r::world] of the sorted list.
"""
import glob
import os
import time

import numpy

from ... import _lib
from ...config import F2Config
from ...gammatone import filters
from ...iopipe import JobReport, Unreadable, host_pool, run_batches
from ...runtime import shard_for_rank
from ...wavio import read_audio


def GetArrayFromWAV(filename):
    """(framerate, samples) of a RIFF or NIST SPHERE file (reference :28-39)."""
    return read_audio(filename)


def GetFilteredOutputFromArray(array, FILTERBANK_COEFFICIENTS):
    return filters.erb_filterbank(array, FILTERBANK_COEFFICIENTS)


def GetFilteredOutputFromFile(filename, FILTERBANK_COEFFICIENTS):
    framerate, wavArray = GetArrayFromWAV(filename)
    return GetFilteredOutputFromArray(wavArray, FILTERBANK_COEFFICIENTS), framerate


def saveGFBMatrix(filename, matrix):
    numpy.save(filename, matrix)   # (in place: see iopipe.save_npy_atomic for why the batch writers do not rename)


def loadGFBMatrix(filename):
    return numpy.load(filename + '.npy')


def GammatoneFiltering(wavFile, FILTERBANK_COEFFICIENTS, n=None, done=None):
    gfbFilename = os.path.splitext(wavFile)[0] + '.GFB'
    print("Filtering:\t{}".format(wavFile))
    outputMatrix, _ = GetFilteredOutputFromFile(wavFile, FILTERBANK_COEFFICIENTS)
    print("Saving:\t\t{}.npy".format(gfbFilename))
    saveGFBMatrix(gfbFilename, outputMatrix)
    if n is not None:
        print("\t\t{:<50} done ! {}/{} Files.".format(wavFile, done, n))


def filterbank_from_config(cfg=None):
    cfg = cfg or F2Config()
    cf = filters.centre_freqs(cfg.framerate, cfg.nchannels, cfg.low_freq)
    return cf, filters.make_erb_filters(cfg.framerate, cf)


def FilterAllOrganisedFiles(batch_files=16, skip_existing=False, metrics=None):
    """`prepare filter`: every resources/f2cnn/*/*.WAV -> .GFB.npy (reference :93-128). Returns the JobReport."""
    TotalTime = time.time()
    report = JobReport("prepare filter", skip_existing, metrics)
    wavFiles = sorted(glob.glob(os.path.join("resources", "f2cnn", "*", "*.WAV")))
    if not wavFiles:
        print("NO WAV FILES FOUND, PLEASE ORGANIZE FILES")
        exit(-1)
    print("\n###############################\nApplying FilterBank to files in '{}'.".format(
        os.path.split(wavFiles[0])[0]))
    print(len(wavFiles), "files found")
    _, coefs = filterbank_from_config()
    mine = report.pending(shard_for_rank(wavFiles), lambda name: [os.path.splitext(name)[0] + '.GFB.npy'])
    rates = {}

    def load(name):
        print("Filtering:\t{}".format(name))
        rates[name], samples = GetArrayFromWAV(name)
        return samples

    # batches alternate between two device contexts (streams): the device-to-host copy of one batch runs beside the
    # host-to-device copy and the kernel of the next, while writer threads save the batch before
    contexts = _lib.pipeline_contexts(2)
    turn = [0]

    def compute(loaded):
        loaded = [(n, w) for n, w in loaded if not isinstance(w, Unreadable)]
        if not loaded:
            return []
        ctx = contexts[turn[0] % len(contexts)]
        turn[0] += 1
        names = [n for n, _ in loaded]
        finish = filters.erb_filterbank_batch([w for _, w in loaded], coefs, ctx=ctx, alloc=host_pool.empty, wait=False)
        return lambda: zip(names, finish())

    def save(name, m):
        gfb = os.path.splitext(name)[0] + '.GFB'
        print("Saving:\t\t{}.npy".format(gfb))
        saveGFBMatrix(gfb, m)
        print("\t\t{:<50} done ! {}/{} Files.".format(name, report.add(m.shape[1], rates[name]), len(mine)))

    run_batches(mine, report.guard(load), compute, save, batch=batch_files)
    print("Filtered and Saved all files.")
    print('                Total time:', time.time() - TotalTime)
    print('')
    report.finish()
    return report
