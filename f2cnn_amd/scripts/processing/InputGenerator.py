"""Drop-in for the reference's ``scripts/processing/InputGenerator.py``: labelled timepoints ->
``input_data*.npy`` ((Ntp, 11, C) float32), gathered by HIP kernel K3 instead of Python list comprehensions."""
import csv
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy

from ... import _lib
from ...config import F2Config
from ...runtime import rank_world


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
    is gathered by K3 straight into its place, while reader threads fetch the next envelope files. With several ranks
    (F2CNN_RANK / F2CNN_WORLD or torchrun's variables) rank r gathers files r::world into one shared pre-sized .npy
    (the only cross-rank step of the path, done through the file system, no collective)."""
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
    shape = (int(first_row[-1]), cfg.dots_per_input, cfg.nchannels)
    target = inputFile or os.path.join('trainingData',
                                       'input_data_LPF{}.npy'.format(CUTOFF) if LPF else 'input_data_NOLPF.npy')
    os.makedirs(os.path.dirname(target) or '.', exist_ok=True)
    rank, world = rank_world()
    print("Output shape:", shape)
    if world == 1:
        windows = numpy.empty(shape, dtype=numpy.float32)
    else:
        # several ranks (one per GPU): every row's place is known up front, so rank 0 creates the pre-sized .npy,
        # every rank gathers its files r::world straight into it, and rank 0 finishes once all ranks have reported
        windows = _shared_output(target, shape, rank, world)

    def fetch(name):
        return numpy.load(os.path.join('resources', 'f2cnn', name))
    mine = list(range(len(order)))[rank::world]
    with ThreadPoolExecutor(4) as readers:
        ahead = [readers.submit(fetch, order[k]) for k in mine[:3]]
        for pos, k in enumerate(mine):
            envelopes = ahead.pop(0).result()
            if pos + 3 < len(mine):
                ahead.append(readers.submit(fetch, order[mine[pos + 3]]))
            windows[first_row[k]:first_row[k + 1]] = gather_windows(envelopes, per_file[order[k]], cfg.radius, cfg.step)
            print("\t\t{:<50} done !  {}/{} Files".format(os.path.join('resources', 'f2cnn', order[k]), k + 1, len(order)))
    print('Generated Input Matrix of shape {}.'.format(shape))
    print("Saving as {}...".format(target))
    backup = os.path.join('trainingData', 'last_input_data.npy')      # the reference's second copy (:90)
    if world == 1:
        numpy.save(target, windows)
        numpy.save(backup, windows)
    else:
        windows.flush()
        del windows
        _finish_shared_output(target, backup, rank, world)
    print('                Total time:', time.time() - started)
    print('')


def _run_token():
    """What tells this launch's marker files from those a dead earlier launch left behind. $F2CNN_RUN_ID when the launcher
    exports one (ranks started by hand, in separate shells, must: they share nothing else). Otherwise torchrun's
    $TORCHELASTIC_RUN_ID (the literal 'none' unless --rdzv-id is given) or $MASTER_PORT - the same on every launch - TOGETHER
    WITH THE PARENT'S PID: the ranks of one single-node launch are children of one launcher process, and a relaunch after a
    crash has another (round-3 advisor finding: a rank that met the dead launch's `ready` marker mapped the old file, which
    rank 0 then replaced)."""
    v = os.environ.get("F2CNN_RUN_ID")
    if v:
        return "".join(ch if ch.isalnum() else "_" for ch in v)
    base = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
    return "".join(ch if ch.isalnum() else "_" for ch in base) + "_p{}".format(os.getppid())


def _marker(target, what, rank):
    return "{}.{}.{}.rank{}".format(target, what, _run_token(), rank)


def _shared_output(target, shape, rank, world, timeout=600.0):
    """The pre-sized output .npy all ranks write into (numpy.lib.format.open_memmap). Rank 0 creates it - under a
    temporary name, renamed into place once its header is written - and announces it with a marker that carries this
    launch's token and the shape; the others wait for exactly that marker (the ranks of a file command share nothing
    but the file system). Every rank removes only its OWN markers of an earlier launch with the same token."""
    for what in ("ready", "done"):
        if os.path.exists(_marker(target, what, rank)):
            os.remove(_marker(target, what, rank))
    if rank == 0:
        tmp = "{}.tmp.{}".format(target, os.getpid())
        out = numpy.lib.format.open_memmap(tmp, mode='w+', dtype=numpy.float32, shape=shape)
        del out
        os.replace(tmp, target)
        out = numpy.lib.format.open_memmap(target, mode='r+')
        with open(_marker(target, "ready", 0), "w") as f:
            f.write(repr(tuple(shape)))
        return out
    deadline = time.time() + timeout
    started = time.time()
    while True:
        mk = _marker(target, "ready", 0)
        try:
            # a marker of this token that predates this rank by more than a day is a leftover, not an announcement
            if os.path.getmtime(mk) >= started - 86400 and open(mk).read() == repr(tuple(shape)):
                break
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError("rank 0 did not create {} (token {})".format(target, _run_token()))
        time.sleep(0.05)
    out = numpy.lib.format.open_memmap(target, mode='r+')
    if out.shape != shape or out.dtype != numpy.float32:
        raise ValueError("{} has shape {} / dtype {}, expected {} float32".format(target, out.shape, out.dtype, shape))
    return out


def _finish_shared_output(target, backup, rank, world, timeout=3600.0):
    open(_marker(target, "done", rank), "w").close()
    if rank != 0:
        return
    deadline = time.time() + timeout
    while not all(os.path.exists(_marker(target, "done", r)) for r in range(world)):
        if time.time() > deadline:
            raise TimeoutError("not every rank finished its part of {}".format(target))
        time.sleep(0.05)
    import shutil
    if os.path.abspath(backup) != os.path.abspath(target):
        shutil.copyfile(target, backup)
    for r in range(world):
        os.remove(_marker(target, "done", r))
    os.remove(_marker(target, "ready", 0))
