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
from ...config import F2Config
from ...iopipe import JobReport, Unreadable, host_pool, npy_layout, read_npy_into, run_batches
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


def ExtractEnvelopesFromMatrices(matrices, LPF=False, CUTOFF=100, ctx=None, precision=None, alloc=numpy.empty):
    """Batched form: a list of (C,N_b) matrices with the same C -> list of envelopes, one launch per FFT size.
    `alloc(count, dtype=...)` provides the output buffer."""
    ctx = ctx or _lib.default_context()
    if not len(matrices):
        return []
    Cn = matrices[0].shape[0]
    offsets = numpy.zeros(len(matrices) + 1, dtype=numpy.int64)
    offsets[1:] = numpy.cumsum([m.shape[1] for m in matrices])
    # one host buffer: the matrices are packed into it, the library copies it to the device, transforms in place
    # there and copies the envelopes back over it
    out = alloc(Cn * int(offsets[-1]), dtype=numpy.float64)
    for b, m in enumerate(matrices):
        out[Cn * offsets[b]:Cn * offsets[b + 1]].reshape(Cn, -1)[...] = m
    ctx.envelope_batch(out, offsets, len(matrices), Cn, bool(LPF), CUTOFF if LPF else 0.0,
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


def ExtractAllEnvelopes(LPF=False, CUTOFF=100, batch_files=16, skip_existing=False, metrics=None):
    """`prepare envelope`: every resources/f2cnn/*/*.GFB.npy -> .ENV1.npy (reference :125-153). Returns the JobReport."""
    TotalTime = time.time()
    report = JobReport("prepare envelope", skip_existing, metrics)
    framerate = F2Config().framerate
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
    mine = report.pending(shard_for_rank(gfbFiles), lambda name: [envelope_filename(name) + '.npy'],
                          params={"lpf": bool(LPF), "cutoff": CUTOFF if LPF else None, "fft": int(FFT_PRECISION)},
                          stamp_dir=os.path.join('resources', 'f2cnn'))

    contexts = _lib.pipeline_contexts(2)     # alternate streams: one batch's copies beside the next one's kernels
    turn = [0]
    precision = FFT_PRECISION

    def plan(chunk):
        """One pooled page-locked buffer per batch; every file is read straight into its (C, N_b) block of it."""
        try:
            layouts = [npy_layout(name) for name in chunk]
        except OSError:                          # an unreadable file: the general path reports it per file
            return None, [None] * len(chunk)
        plain = all(l is not None and len(l[0]) == 2 and l[0][0] == layouts[0][0][0] for l in layouts)
        if not plain:                            # Fortran order, another dtype, mixed channel counts: general path
            return None, [None] * len(chunk)
        Cn = layouts[0][0][0]
        offsets = numpy.zeros(len(chunk) + 1, dtype=numpy.int64)
        offsets[1:] = numpy.cumsum([l[0][1] for l in layouts])
        buf = host_pool.empty(Cn * int(offsets[-1]))
        slots = [(buf[Cn * offsets[b]:Cn * offsets[b + 1]], layouts[b][1]) for b in range(len(chunk))]
        return {"buf": buf, "offsets": offsets, "C": Cn}, slots

    def load(name, slot):
        print("File:\t{}".format(name))
        if slot is None:
            return numpy.load(name)
        return read_npy_into(name, slot[1], slot[0])

    def compute(loaded, state):
        if any(isinstance(m, Unreadable) for _, m in loaded):
            # (a planned batch with an unreadable file: the good blocks are copied out and go the general way)
            loaded = [(n, numpy.array(m)) for n, m in loaded if not isinstance(m, Unreadable)]
            loaded = [(n, m.reshape(state["C"], -1) if state is not None else m) for n, m in loaded]
            state = None
            if not loaded:
                return []
        names = [n for n, _ in loaded]
        if state is None:
            # one launch per distinct channel count (normally one)
            by_c = {}
            for i, (_, m) in enumerate(loaded):
                by_c.setdefault(m.shape[0], []).append(i)
            envs = [None] * len(loaded)
            for idx in by_c.values():
                for i, e in zip(idx, ExtractEnvelopesFromMatrices([loaded[i][1] for i in idx], LPF, CUTOFF,
                                                                    alloc=host_pool.empty)):
                    envs[i] = e
            return list(zip(names, envs))
        ctx = contexts[turn[0] % len(contexts)]
        turn[0] += 1
        buf, offsets, Cn = state["buf"], state["offsets"], state["C"]
        # in place on the host too: the envelopes come back over the filterbank rows they were computed from
        ctx.envelope_batch(buf, offsets, len(names), Cn, bool(LPF), CUTOFF if LPF else 0.0, precision, buf,
                           _lib.MEM_HOST_ASYNC)

        def finish():
            ctx.synchronize()
            return [(names[b], buf[Cn * offsets[b]:Cn * offsets[b + 1]].reshape(Cn, -1)) for b in range(len(names))]
        return finish

    def save(name, e):
        numpy.save(envelope_filename(name), e)
        print("\t{:<50} done ! {}/{} Files.".format(envelope_filename(name), report.add(e.shape[1], framerate), len(mine)))

    run_batches(mine, report.guard(load), compute, save, batch=batch_files, plan=plan)
    print("Extracted Envelopes from all files.")
    print('              Total time:', time.time() - TotalTime)
    print('')
    report.finish()
    return report


def FilterAndExtractAll(LPF=False, CUTOFF=100, batch_files=16, keep_gfb=True, skip_existing=False, metrics=None):
    """`prepare filter` followed by `prepare envelope` in ONE pass over the corpus: every
    resources/f2cnn/*/*.WAV -> <base>.GFB.npy (unless keep_gfb is False) and <base>.ENV1.npy, through
    f2_filterbank_envelope_fused. Nothing is read back from disk between the two stages (the reference writes the
    16 MB .GFB.npy per second of audio and loads it again, f2cnn.py:17-18)."""
    from ... import _lib
    from ...gammatone import filters
    from .GammatoneFiltering import GetArrayFromWAV, filterbank_from_config
    TotalTime = time.time()
    report = JobReport("prepare features", skip_existing, metrics)
    wavFiles = sorted(glob.glob(os.path.join("resources", "f2cnn", "*", "*.WAV")))
    if not wavFiles:
        print("NO WAV FILES FOUND, PLEASE ORGANIZE FILES")
        exit(-1)
    print("\n###############################\nFiltering and extracting envelopes of files in '{}'.".format(
        os.path.split(wavFiles[0])[0]))
    print(len(wavFiles), "files found")
    _, coefs = filterbank_from_config()
    coefs = numpy.ascontiguousarray(coefs, dtype=numpy.float64)
    Cn = coefs.shape[0]
    contexts = _lib.pipeline_contexts(2)
    turn = [0]
    def outputs(name):
        base = os.path.splitext(name)[0]
        return [base + '.ENV' + str(METHOD) + '.npy'] + ([base + '.GFB.npy'] if keep_gfb else [])
    mine = report.pending(shard_for_rank(wavFiles), outputs,
                          params={"lpf": bool(LPF), "cutoff": CUTOFF if LPF else None, "fft": int(FFT_PRECISION)},
                          stamp_dir=os.path.join('resources', 'f2cnn'))
    rates = {}

    def load(name):
        rates[name], samples = GetArrayFromWAV(name)
        return samples

    def compute(loaded):
        loaded = [(n, w) for n, w in loaded if not isinstance(w, Unreadable)]
        if not loaded:
            return []
        ctx = contexts[turn[0] % len(contexts)]
        turn[0] += 1
        args = [filters._wave_args(w) for _, w in loaded]
        dt = args[0][1] if all(a[1] == args[0][1] for a in args) else _lib.WAVE_F64
        dtype = numpy.int16 if dt == _lib.WAVE_I16 else numpy.float64
        offsets = numpy.zeros(len(args) + 1, dtype=numpy.int64)
        offsets[1:] = numpy.cumsum([a[0].shape[0] for a in args])
        flat = numpy.concatenate([a[0].astype(dtype, copy=False) for a in args])
        env = host_pool.empty(Cn * int(offsets[-1]))
        gfb = host_pool.empty(Cn * int(offsets[-1])) if keep_gfb else None
        ctx.filterbank_envelope_fused(flat, dt, offsets, coefs, len(args), Cn, bool(LPF), CUTOFF if LPF else 0.0,
                                      FFT_PRECISION, env, gfb, _lib.MEM_HOST_ASYNC)

        def finish(keep=(flat,)):
            ctx.synchronize()
            for b, (name, _) in enumerate(loaded):
                sl = slice(Cn * offsets[b], Cn * offsets[b + 1])
                yield name, (env[sl].reshape(Cn, -1), gfb[sl].reshape(Cn, -1) if keep_gfb else None)
        return finish

    def save(name, res):
        env, gfb = res
        base = os.path.splitext(name)[0]
        if gfb is not None:
            numpy.save(base + '.GFB', gfb)
        numpy.save(base + '.ENV' + str(METHOD), env)
        print("\t{:<50} done ! {}/{} Files.".format(base + '.ENV' + str(METHOD), report.add(env.shape[1], rates[name]),
                                                     len(mine)))

    run_batches(mine, report.guard(load), compute, save, batch=batch_files)
    print("Filtered and extracted envelopes of all files.")
    print('              Total time:', time.time() - TotalTime)
    print('')
    report.finish()
    return report
