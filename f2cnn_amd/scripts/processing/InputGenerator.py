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
    exports one (ranks started by hand, in separate shells or on several nodes, should: they share nothing else).
    Otherwise torchrun's $TORCHELASTIC_RUN_ID (the literal 'none' unless --rdzv-id is given) or $MASTER_PORT - the same on
    every node of a launch, but also on every RE-launch. The parent's PID is added only where it provably is common to all
    ranks: a single-node launch ($LOCAL_WORLD_SIZE == $WORLD_SIZE) whose ranks are direct children of one launcher
    (torchrun / `bench.py --gpus N` style). Ranks on several nodes, or behind per-rank wrapper shells, have different
    parents (round-4 advisor finding: they then waited 600 s for a marker that could not come) and fall back to the
    launcher id alone; what protects THEM from a dead launch's leftovers is the marker's content (`_shared_output`)."""
    v = os.environ.get("F2CNN_RUN_ID")
    if v:
        return "".join(ch if ch.isalnum() else "_" for ch in v)
    base = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
    token = "".join(ch if ch.isalnum() else "_" for ch in base)
    lws, ws = os.environ.get("LOCAL_WORLD_SIZE"), os.environ.get("WORLD_SIZE")
    if lws is not None and lws == ws and os.environ.get("F2CNN_RANKS_SHARE_PARENT", "1") != "0":
        token += "_p{}".format(os.getppid())
    return token


def _marker(target, what, rank):
    return "{}.{}.{}.rank{}".format(target, what, _run_token(), rank)


def _foreign_marker(target, what, rank, newer_than):
    """A fresh marker of the same kind and rank under ANOTHER token: the ranks of this launch disagree about the token."""
    import glob
    mine = _marker(target, what, rank)
    for path in glob.glob("{}.{}.*.rank{}".format(glob.escape(target), what, rank)):
        try:
            if path != mine and os.path.getmtime(path) >= newer_than:
                return path
        except OSError:
            pass
    return None


STALE_WINDOW = 120.0     # seconds a rank may start after rank 0 announced the file (launch skew + clock skew across nodes)
FOREIGN_WINDOW = 5.0     # a marker under ANOTHER token this close to (or after) a rank's own start belongs to its own launch
_SEEN_NONCE = {}         # target -> nonce of the `ready` marker this process mapped the file under


def _shared_output(target, shape, rank, world, timeout=600.0):
    """The pre-sized output .npy all ranks write into (numpy.lib.format.open_memmap). Rank 0 creates it - under a
    temporary name, renamed into place once its header is written - and announces it with a marker that carries this
    launch's token and, as its content, the shape, a nonce, rank 0's start time and the identity (inode, size) of the file it
    created; the others wait for exactly that marker (the ranks of a file command share nothing but the file system),
    accept it only if it is not older than their own start minus STALE_WINDOW, and after mapping the file check that it
    still IS the file the marker describes (a leftover marker of a dead launch with the same token describes a file that
    rank 0 has since replaced). Every rank removes only its OWN markers of an earlier launch with the same token."""
    import json
    started = time.time()
    for what in ("ready", "done"):
        if os.path.exists(_marker(target, what, rank)):
            os.remove(_marker(target, what, rank))
    if rank == 0:
        tmp = "{}.tmp.{}".format(target, os.getpid())
        out = numpy.lib.format.open_memmap(tmp, mode='w+', dtype=numpy.float32, shape=shape)
        del out
        os.replace(tmp, target)
        out = numpy.lib.format.open_memmap(target, mode='r+')
        st = os.stat(target)
        mk = _marker(target, "ready", 0)
        with open(mk + ".tmp", "w") as f:
            json.dump({"shape": list(shape), "nonce": "{}-{}".format(os.getpid(), time.time_ns()), "started": started,
                       "inode": st.st_ino, "size": st.st_size}, f)
        os.replace(mk + ".tmp", mk)
        return out
    deadline = started + timeout
    mk = _marker(target, "ready", 0)
    while True:
        info = None
        try:
            if os.path.getmtime(mk) >= started - STALE_WINDOW:
                with open(mk) as f:
                    info = json.load(f)
        except (OSError, ValueError):
            info = None
        if info is not None and tuple(info.get("shape", ())) == tuple(shape):
            try:
                out = numpy.lib.format.open_memmap(target, mode='r+')
                st = os.stat(target)
            except (OSError, ValueError):
                out = None
            if out is not None and st.st_ino == info.get("inode") and st.st_size == info.get("size"):
                _SEEN_NONCE[os.path.abspath(target)] = info.get("nonce")
                break                       # the file on disk is the one this marker announced
            del out                         # rank 0 is replacing it right now (or the marker is a leftover): look again
        other = _foreign_marker(target, "ready", 0, started - FOREIGN_WINDOW)
        if other:
            raise RuntimeError("{} was announced as {} while rank {} expects token {!r}: the ranks of this launch do not "
                               "share a launch token - export F2CNN_RUN_ID=<same value> on every rank (or another launch "
                               "is writing the same file / died seconds ago: remove its marker)"
                               .format(target, os.path.basename(other), rank, _run_token()))
        if time.time() > deadline:
            raise TimeoutError("rank 0 did not create {} (token {!r}; export F2CNN_RUN_ID=<same value> on every rank if the "
                               "ranks were not started by one launcher)".format(target, _run_token()))
        time.sleep(0.05)
    if out.shape != shape or out.dtype != numpy.float32:
        raise ValueError("{} has shape {} / dtype {}, expected {} float32".format(target, out.shape, out.dtype, shape))
    return out


def _finish_shared_output(target, backup, rank, world, timeout=3600.0):
    if rank != 0:
        # the file this rank wrote into must still be the one rank 0 announced: a leftover `ready` marker of a dead launch
        # with the same token can be met before rank 0 of THIS launch has replaced it - then the rows written are gone
        import json
        try:
            with open(_marker(target, "ready", 0)) as f:
                nonce = json.load(f).get("nonce")
        except (OSError, ValueError):
            nonce = None
        if nonce != _SEEN_NONCE.get(os.path.abspath(target)):
            raise RuntimeError("rank 0 replaced {} after rank {} had mapped it (a marker left by an earlier launch with the "
                               "same token {!r}): this rank's rows are lost - rerun with F2CNN_RUN_ID=<fresh value> on "
                               "every rank".format(target, rank, _run_token()))
    open(_marker(target, "done", rank), "w").close()
    if rank != 0:
        return
    began = time.time()
    deadline = began + timeout
    while not all(os.path.exists(_marker(target, "done", r)) for r in range(world)):
        for r in range(1, world):
            other = None if os.path.exists(_marker(target, "done", r)) else _foreign_marker(target, "done", r, began - FOREIGN_WINDOW)
            if other:
                raise RuntimeError("rank {} reported {} but rank 0 expects token {!r}: export F2CNN_RUN_ID=<same value> on "
                                   "every rank".format(r, os.path.basename(other), _run_token()))
        if time.time() > deadline:
            raise TimeoutError("not every rank finished its part of {} (token {!r})".format(target, _run_token()))
        time.sleep(0.05)
    import shutil
    if os.path.abspath(backup) != os.path.abspath(target):
        shutil.copyfile(target, backup)
    for r in range(world):
        os.remove(_marker(target, "done", r))
    os.remove(_marker(target, "ready", 0))
