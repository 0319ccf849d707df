"""Drop-in for the reference's ``scripts/processing/EnvelopeExtraction.py`` (hot path only).

``ExtractEnvelopeFromMatrix`` (reference :51-67) is HIP kernel K2 reached through
``f2_envelope_batch``; the file drivers keep the reference's names, file layout and messages
(``ExtractEnvelope`` :70-83, ``SaveEnvelope`` :86-99, ``ExtractAndSaveEnvelope`` :101-117,
``ExtractAllEnvelopes`` :125-153) but process files in batches on the GPU instead of a process pool.
"""
import glob
import os
import time

import numpy

from ... import _lib
from ...runtime import shard_for_rank

FFT_PRECISION = _lib.FFT_F64 if os.environ.get("F2CNN_FFT", "f32").lower() in ("f64", "double") else _lib.FFT_F32
METHOD = 1   # suffix of the .ENV<METHOD>.npy files (reference :91)


def ExtractEnvelopeFromMatrix(matrix, LPF=False, CUTOFF=100, ctx=None, precision=None):
    """|analytic signal| of every row (zero-padded FFT Hilbert transform), then an optional first-order
    Butterworth low-pass at ``CUTOFF`` Hz. (C,N) float64 in, (C,N) float64 out; the input is not modified."""
    ctx = ctx or _lib.default_context()
    matrix = numpy.ascontiguousarray(matrix, dtype=numpy.float64)
    if matrix.ndim != 2:
        raise ValueError("matrix must be two dimensional (channels x samples)")
    Cn, N = matrix.shape
    out = numpy.empty_like(matrix)
    offsets = numpy.array([0, N], dtype=numpy.int64)
    ctx.envelope_batch(matrix, offsets, 1, Cn, bool(LPF), CUTOFF if LPF else 0.0,
                       FFT_PRECISION if precision is None else precision, out, _lib.MEM_HOST)
    return out


def ExtractEnvelopesFromMatrices(matrices, LPF=False, CUTOFF=100, ctx=None, precision=None):
    """Batched form: a list of (C,N_b) matrices with the same C -> list of envelopes, one launch per FFT size."""
    ctx = ctx or _lib.default_context()
    if not len(matrices):
        return []
    Cn = matrices[0].shape[0]
    offsets = numpy.zeros(len(matrices) + 1, dtype=numpy.int64)
    offsets[1:] = numpy.cumsum([m.shape[1] for m in matrices])
    flat = numpy.concatenate([numpy.ascontiguousarray(m, dtype=numpy.float64).ravel() for m in matrices])
    out = numpy.empty_like(flat)
    ctx.envelope_batch(flat, offsets, len(matrices), Cn, bool(LPF), CUTOFF if LPF else 0.0,
                       FFT_PRECISION if precision is None else precision, out, _lib.MEM_HOST)
    return [out[Cn * offsets[b]:Cn * offsets[b + 1]].reshape(Cn, -1) for b in range(len(matrices))]


def ExtractEnvelope(gfbFileName, LPF=False, CUTOFF=100):
    """Envelope matrix of one ``.GFB.npy`` file (reference :70-83)."""
    print("File:\t{}".format(gfbFileName))
    return ExtractEnvelopeFromMatrix(numpy.load(gfbFileName), LPF, CUTOFF)


def envelope_filename(gfbFileName):
    """<base>.GFB.npy -> <base>.ENV1 (numpy.save appends .npy), reference :94."""
    return os.path.splitext(os.path.splitext(gfbFileName)[0])[0] + ".ENV" + str(METHOD)


def SaveEnvelope(matrix, gfbFileName, nbf=None, done=None):
    numpy.save(envelope_filename(gfbFileName), matrix)
    if nbf is not None:
        print("\t{:<50} done ! {}/{} Files.".format(envelope_filename(gfbFileName), done, nbf))


def ExtractAndSaveEnvelope(gfbFileName, nbf=None, LPF=False, CUTOFF=100):
    SaveEnvelope(ExtractEnvelope(gfbFileName, LPF, CUTOFF), gfbFileName, nbf, 1)


def ExtractAllEnvelopes(LPF=False, CUTOFF=100, batch_files=16):
    """`prepare envelope`: every resources/f2cnn/*/*.GFB.npy -> .ENV1.npy (reference :125-153)."""
    TotalTime = time.time()
    gfbFiles = sorted(glob.glob(os.path.join("resources", "f2cnn", "*", "*.GFB.npy")))
    if not gfbFiles:   # the reference indexes [0] first and dies with IndexError; same message, checked first
        print("ERROR: NO .GFB.npy FILES FOUND, PLEASE GENERATE FILTERED OUTPUTS")
        exit(-1)
    print("\n###############################\nExtracting Envelopes from files in '{}'.".format(
        os.path.split(gfbFiles[0])[0]))
    if LPF:
        print("Using Low Pass Filtering with a cutoff at {}Hz".format(CUTOFF))
    else:
        print("Not using Low Pass Filtering")
    print(len(gfbFiles), ".GFB.npy files found")
    mine = shard_for_rank(gfbFiles)
    done = 0
    for s in range(0, len(mine), batch_files):
        names = mine[s:s + batch_files]
        mats = []
        for name in names:
            print("File:\t{}".format(name))
            mats.append(numpy.load(name))
        # one launch per distinct channel count (normally one)
        by_c = {}
        for i, m in enumerate(mats):
            by_c.setdefault(m.shape[0], []).append(i)
        envs = [None] * len(mats)
        for idx in by_c.values():
            for i, e in zip(idx, ExtractEnvelopesFromMatrices([mats[i] for i in idx], LPF, CUTOFF)):
                envs[i] = e
        for name, e in zip(names, envs):
            done += 1
            SaveEnvelope(e, name, len(mine), done)
    print("Extracted Envelopes from all files.")
    print('              Total time:', time.time() - TotalTime)
    print('')
