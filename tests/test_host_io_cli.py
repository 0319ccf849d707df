"""CPU-only tests of the host side: audio readers, config, CSV parsing, model container, CLI parsing, sharding."""
import os

import numpy as np
import pytest

from f2cnn_amd import cli, config, model, runtime, wavio
from f2cnn_amd.scripts.processing import EnvelopeExtraction, InputGenerator


def test_sphere_and_riff_roundtrip(tmp_path):
    from scipy.io import wavfile
    x = (np.random.default_rng(0).standard_normal(1234) * 3000).astype(np.int16)
    sph, riff = tmp_path / "a.WAV", tmp_path / "b.WAV"
    wavio.write_sphere(sph, 16000, x)
    wavfile.write(riff, 16000, x)
    for p in (sph, riff):
        rate, y = wavio.read_audio(p)
        assert rate == 16000 and y.dtype == np.int16
        np.testing.assert_array_equal(x, y)
    # big-endian SPHERE body
    raw = open(sph, "rb").read()
    be = raw[:1024].replace(b"sample_byte_format -s2 01", b"sample_byte_format -s2 10") + x.astype(">i2").tobytes()
    p = tmp_path / "c.WAV"
    p.write_bytes(be)
    np.testing.assert_array_equal(wavio.read_audio(p)[1], x)
    (tmp_path / "d.WAV").write_bytes(b"JUNKJUNKJUNKJUNKJUNK")
    with pytest.raises(ValueError):
        wavio.read_audio(tmp_path / "d.WAV")


def nist_header(fields, header_size=1024):
    """A NIST SPHERE header assembled from the format's grammar (not through wavio.write_sphere): the magic line
    'NIST_1A', the header size right-justified in 7 characters, then `name -type value` lines (type -i integer,
    -r real, -sN string of N characters), 'end_head', padded with blanks to the declared size."""
    lines = [b"NIST_1A", b"%7d" % header_size]
    for name, value in fields:
        if isinstance(value, int):
            lines.append(b"%s -i %d" % (name.encode(), value))
        elif isinstance(value, float):
            lines.append(b"%s -r %f" % (name.encode(), value))
        else:
            lines.append(b"%s -s%d %s" % (name.encode(), len(value), value.encode()))
    head = b"\n".join(lines + [b"end_head"]) + b"\n"
    assert len(head) <= header_size
    return head.ljust(header_size, b" ")


TIMIT_FIELDS = [("database_id", "TIMIT"), ("database_version", "1.0"), ("utterance_id", "cjf0_sa1"),
                ("channel_count", 1), ("sample_count", 0), ("sample_rate", 16000), ("sample_min", -2191),
                ("sample_max", 2790), ("sample_n_bytes", 2), ("sample_byte_format", "01"), ("sample_sig_bits", 16)]


def fields_with(**over):
    out = []
    for k, v in TIMIT_FIELDS:
        v = over.pop(k, v)
        if v is not None:
            out.append((k, v))
    return out + list(over.items())


def test_sphere_files_built_from_the_header_grammar(tmp_path):
    """What the reference reads through sphfile.SPHFile (GammatoneFiltering.py:33-38): TIMIT-style little-endian
    headers, big-endian bodies, larger headers, a missing sample_count, two channels; compressed bodies are refused."""
    x = (np.random.default_rng(1).standard_normal(3001) * 5000).astype(np.int16)
    p = tmp_path / "x.WAV"
    # 1. a TIMIT header as the corpus ships it (string fields with their lengths, min/max fields the reader ignores)
    p.write_bytes(nist_header(fields_with(sample_count=len(x))) + x.astype("<i2").tobytes())
    rate, y = wavio.read_audio(p)
    assert rate == 16000 and y.dtype == np.int16 and y.flags["C_CONTIGUOUS"]
    np.testing.assert_array_equal(y, x)
    # 2. big-endian samples (sample_byte_format 10), 8 kHz, 2048-byte header, a string value with blanks
    p.write_bytes(nist_header(fields_with(sample_count=len(x), sample_byte_format="10", sample_rate=8000,
                                          speaking_mode="read speech"), 2048) + x.astype(">i2").tobytes())
    rate, y = wavio.read_audio(p)
    assert rate == 8000
    np.testing.assert_array_equal(y, x)
    # 3. no sample_count: everything after the header
    p.write_bytes(nist_header(fields_with(sample_count=None)) + x.astype("<i2").tobytes())
    np.testing.assert_array_equal(wavio.read_audio(p)[1], x)
    # 4. sample_count smaller than the body: trailing bytes are not samples
    p.write_bytes(nist_header(fields_with(sample_count=1000)) + x.astype("<i2").tobytes())
    np.testing.assert_array_equal(wavio.read_audio(p)[1], x[:1000])
    # 5. two interleaved channels: the first one (TIMIT is mono; sample_count counts frames)
    two = np.stack([x, -x], axis=1)
    p.write_bytes(nist_header(fields_with(sample_count=len(x), channel_count=2, sample_byte_format="10")) +
                  two.astype(">i2").tobytes())
    np.testing.assert_array_equal(wavio.read_audio(p)[1], x)
    # 6. shorten-compressed and mu-law bodies are refused loudly, as is a header that is not SPHERE at all
    for coding, nbytes in (("pcm,embedded-shorten-v2.00", 2), ("ulaw", 1), ("pcm", 1), ("pcm", 4)):
        p.write_bytes(nist_header(fields_with(sample_count=len(x), sample_coding=coding, sample_n_bytes=nbytes)) +
                      x.astype("<i2").tobytes())
        with pytest.raises(ValueError):
            wavio.read_audio(p)
    p.write_bytes(b"NIST_1B\n   1024\n".ljust(1024) + x.tobytes())
    with pytest.raises(ValueError):
        wavio.read_audio(p)
    # 7. the writer used for synthetic corpora emits a header this grammar accepts, byte for byte
    wavio.write_sphere(p, 16000, x)
    raw = p.read_bytes()
    want = nist_header([("channel_count", 1), ("sample_count", len(x)), ("sample_rate", 16000), ("sample_n_bytes", 2),
                        ("sample_byte_format", "01"), ("sample_sig_bits", 16), ("sample_coding", "pcm")])
    assert raw[:1024] == want and raw[1024:] == x.astype("<i2").tobytes()


def test_config_defaults_and_file(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    c = config.F2Config()
    assert not c.found and (c.framerate, c.nchannels, c.low_freq, c.radius) == (16000, 128, 100, 5)
    assert c.dots_per_input == 11 and c.step == 160
    config.write_default(nchannels=64, radius=3)
    c = config.F2Config()
    assert c.found and c.nchannels == 64 and c.dots_per_input == 7
    # the file is readable the way the reference reads it (ConfigParser lower-cases keys)
    from configparser import ConfigParser
    cp = ConfigParser()
    cp.read("configF2CNN.conf")
    assert cp.getint("FILTERBANK", "NCHANNELS") == 64 and cp.getint("CNN", "SAMPLING_PERIOD") == 10000


def test_label_csv_parsing(tmp_path):
    p = tmp_path / "l.csv"
    p.write_text("TRAIN,DR2,MABC0,SX10,aa,8000,0.5,0.01,1\nTEST,DR1,FXYZ0,SA1,iy,800,0.5,0.01,0\n"
                 "TRAIN,DR2,MABC0,SX10,eh,960,0.5,0.01,1\n")
    d = InputGenerator.GetListOfEnvelopeFilesAndTimepoints(str(p))
    assert d == {os.path.join("TRAIN", "DR2.MABC0.SX10.ENV1.npy"): [8000, 960],
                 os.path.join("TEST", "DR1.FXYZ0.SA1.ENV1.npy"): [800]}


def test_envelope_filename():
    assert EnvelopeExtraction.envelope_filename("resources/f2cnn/TEST/DR1.X.SA1.GFB.npy") == "resources/f2cnn/TEST/DR1.X.SA1.ENV1"


def test_model_container(tmp_path):
    m = model.F2CNNModel.glorot(7)
    assert sum(t.size for t in m.tensors.values()) == 1057262
    assert [t.shape for t in m.ordered()][:4] == [(3, 3, 1, 32), (32,), (3, 3, 32, 32), (32,)]
    m.save(tmp_path / "w.npz")
    m2 = model.load_model(tmp_path / "w.npz")
    for k in m.tensors:
        np.testing.assert_array_equal(m.tensors[k], m2.tensors[k])
    assert model.flatten_size(11, 128) == (1, 30, 1920) and model.flatten_size(11, 64)[2] == 896
    bad = dict(m.tensors)
    bad["conv2_w"] = bad["conv2_w"][..., :16]
    with pytest.raises(ValueError):
        model.F2CNNModel(bad)


def test_cli_parser():
    p = cli.build_parser()
    a = p.parse_args(["prepare", "envelope", "--cutoff", "50"])
    assert a.prepare_command == "envelope" and a.CUTOFF == 50
    a = p.parse_args(["prepare", "input", "-l", "x.csv", "-i", "y.npy"])
    assert a.labelFile == "x.csv" and a.inputFile == "y.npy" and a.CUTOFF is None
    a = p.parse_args(["cnn", "eval", "-f", "a.WAV", "--lpf", "50", "-m", "w.npz"])
    assert a.cnn_command == "eval" and a.file == "a.WAV" and a.CUTOFF == 50 and a.model == "w.npz"
    a = p.parse_args(["cnn", "evalnoise", "-f", "a.WAV", "-n", "-3"])
    assert a.cnn_command == "evalnoise" and a.SNRdB == -3.0
    a = p.parse_args(["cnn", "evalrand", "-c", "5"])
    assert a.cnn_command == "evalrand" and a.count == 5 and a.file is None
    with pytest.raises(SystemExit):
        p.parse_args(["prepare", "organize"])
    assert cli.main([]) == 1
    assert cli.main(["cnn", "eval"]) == 1


def test_noise_helpers():
    from f2cnn_amd.scripts.CNN import Evaluating
    assert Evaluating.SNRdbToSNRlinear(10) == 10 and abs(Evaluating.SNRdbToSNRlinear(-3) - 0.5011872) < 1e-6
    x = np.full(1000, 30000, np.int16)          # squares overflow int16/int32 arithmetic if not widened
    assert abs(Evaluating.RMS(x) - 30000.0) < 1e-9


def test_empty_corpus_exits_like_the_reference(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    from f2cnn_amd.scripts.processing import GammatoneFiltering
    with pytest.raises(SystemExit) as e:
        GammatoneFiltering.FilterAllOrganisedFiles()
    assert e.value.code == -1 and "NO WAV FILES FOUND" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        EnvelopeExtraction.ExtractAllEnvelopes()
    with pytest.raises(SystemExit):
        InputGenerator.GenerateInputData()


def test_shard_partition():
    files = [f"f{i:03d}" for i in range(23)]
    for world in (1, 2, 3, 8):
        parts = [runtime.shard(files, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == files
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_iopipe_order_and_completeness():
    import threading
    import time
    from f2cnn_amd.iopipe import run_batches
    seen, saved, lock = [], {}, threading.Lock()

    def load(i):
        time.sleep(0.001 * (i % 3))
        return i * 10

    def compute(loaded):
        seen.append([i for i, _ in loaded])
        return [(i, v + 1) for i, v in loaded]

    def save(i, v):
        with lock:
            saved[i] = v

    assert run_batches(range(23), load, compute, save, batch=5, readers=3, writers=2, max_pending_writes=4) == 23
    assert seen == [list(range(s, min(s + 5, 23))) for s in range(0, 23, 5)]
    assert saved == {i: i * 10 + 1 for i in range(23)}
    assert run_batches([], load, compute, save) == 0


def test_run_batches_with_plans_and_deferred_computes(tmp_path):
    """plan -> per-item slots, computes that finish later (the asynchronous device batches), ordered collection."""
    from f2cnn_amd.iopipe import npy_layout, read_npy_into, run_batches
    mats = [np.random.default_rng(i).random((3, 40 + i)) for i in range(11)]
    for i, m in enumerate(mats):
        np.save(tmp_path / f"m{i}.npy", m)
    np.save(tmp_path / "fortran.npy", np.asfortranarray(mats[0]))
    np.save(tmp_path / "f4.npy", mats[0].astype(np.float32))
    assert npy_layout(tmp_path / "fortran.npy") is None and npy_layout(tmp_path / "f4.npy") is None
    shape, off = npy_layout(tmp_path / "m3.npy")
    assert shape == (3, 43) and off % 64 == 0
    events, saved = [], {}

    def plan(chunk):
        lay = [npy_layout(tmp_path / f"m{i}.npy") for i in chunk]
        buf = np.empty(sum(l[0][0] * l[0][1] for l in lay))
        cuts = np.cumsum([0] + [l[0][0] * l[0][1] for l in lay])
        return {"buf": buf, "cuts": cuts}, [(buf[a:b], l[1]) for a, b, l in zip(cuts[:-1], cuts[1:], lay)]

    def load(i, slot):
        return read_npy_into(tmp_path / f"m{i}.npy", slot[1], slot[0])

    def compute(loaded, state):
        ids = [i for i, _ in loaded]
        events.append(("submit", ids[0]))
        for (i, got), a, b in zip(loaded, state["cuts"][:-1], state["cuts"][1:]):
            assert got.base is state["buf"] or got is state["buf"]            # read straight into the batch buffer
            np.testing.assert_array_equal(state["buf"][a:b], mats[i].reshape(-1))

        def finish():
            events.append(("finish", ids[0]))
            return [(i, state["buf"][a:b].copy() * 2) for i, a, b in zip(ids, state["cuts"][:-1], state["cuts"][1:])]
        return finish

    def save(i, v):
        saved[i] = v

    assert run_batches(range(11), load, compute, save, batch=4, readers=3, writers=2, plan=plan, depth=2) == 11
    # two batches are queued before the first is collected; collection keeps submission order
    assert events == [("submit", 0), ("submit", 4), ("finish", 0), ("submit", 8), ("finish", 4), ("finish", 8)]
    for i, m in enumerate(mats):
        np.testing.assert_array_equal(saved[i], m.reshape(-1) * 2)
    with pytest.raises(ValueError):
        read_npy_into(tmp_path / "m0.npy", off, np.empty(10 ** 4))            # more than the file holds


def test_array_pool_recycles_only_unreferenced_owners():
    import gc
    from f2cnn_amd.iopipe import ArrayPool
    pool = ArrayPool(keep=3)
    a = pool.empty(1000)
    addr_a = a.ctypes.data
    a[:] = 7.0                                              # writable
    b = pool.empty(500)
    assert b.ctypes.data != addr_a                          # a is alive: a second buffer
    view = a[10:20].reshape(2, 5)                           # a view of a view keeps the lease
    assert view.base is not None
    del a
    gc.collect()
    c = pool.empty(800)
    assert c.ctypes.data != addr_a
    del view
    gc.collect()
    d = pool.empty(900, dtype=np.float32)
    assert d.ctypes.data == addr_a and d.dtype == np.float32 and d.shape == (900,)
    for _ in range(10):                                     # the pool does not grow without bound
        pool.empty(100)
    gc.collect()
    assert len(pool._buffers) <= 4
    # leases released from other threads (the .npy writers) are seen by the allocating thread
    import threading
    e = pool.empty(2000)
    addr_e = e.ctypes.data
    t = threading.Thread(target=lambda v: None, args=(e[:5],))
    del e
    t.start()
    t.join()
    del t
    gc.collect()
    assert pool.empty(2000).ctypes.data == addr_e


def test_job_report_resume_failures_and_metrics(tmp_path, monkeypatch):
    """The bookkeeping of the file commands (SURVEY section 5 hooks): skip what is up to date, survive unreadable
    inputs, write the totals as JSON, exit status 2 when something failed."""
    import json
    import time
    from f2cnn_amd.iopipe import JobReport, Unreadable
    monkeypatch.chdir(tmp_path)
    srcs = []
    for i in range(4):
        p = tmp_path / f"in{i}.WAV"
        p.write_bytes(b"x")
        srcs.append(str(p))
    import numpy as np
    np.save(tmp_path / "in0.GFB.npy", np.zeros((2, 3)))               # fresh, complete output: skipped
    np.save(tmp_path / "in1.GFB.npy", np.zeros((2, 3)))
    os.utime(tmp_path / "in1.GFB.npy", (time.time() - 100, time.time() - 100))   # older than its input: redone
    outs = lambda name: [os.path.splitext(name)[0] + ".GFB.npy"]
    rep = JobReport("prepare filter", skip_existing=True, metrics=str(tmp_path / "m.json"))
    assert rep.pending(srcs, outs) == srcs[1:] and rep.skipped == 1
    assert JobReport("x").pending(srcs, outs) == srcs                   # without the flag everything runs

    def load(name):
        if name.endswith("in2.WAV"):
            raise ValueError("SPHERE sample_coding 'shorten' is not supported")
        return np.zeros(16000, np.int16)
    got = [rep.guard(load)(s) for s in srcs[1:]]
    assert isinstance(got[1], Unreadable) and "shorten" in got[1].reason and not isinstance(got[0], Unreadable)
    assert rep.add(16000, 16000) == 1 and rep.add(8000, 16000) == 2
    out = rep.finish()
    assert rep.exit_status == 2 and out["files"] == 2 and out["files_skipped"] == 1 and out["files_failed"] == 1
    assert out["audio_seconds"] == 1.5 and out["audio_s_per_s"] > 0
    assert json.load(open(tmp_path / "m.json"))["failed"][0][0].endswith("in2.WAV")
    args = cli.build_parser().parse_args(["prepare", "features", "--skip-existing", "--metrics", "m.json", "-c", "50"])
    assert args.skip_existing and args.metrics == "m.json" and args.CUTOFF == 50


def test_atomic_npy_save_and_resume_check(tmp_path):
    """The file drivers' writers never leave a partial file under the final name, and the resume logic recognises one
    (a killed `numpy.save` used to be skipped for good by --skip-existing)."""
    import numpy as np
    from f2cnn_amd import iopipe
    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    out = iopipe.save_npy_atomic(str(tmp_path / "x.GFB"), a)
    assert out.endswith("x.GFB.npy") and np.array_equal(np.load(out), a)
    assert [f.name for f in tmp_path.iterdir()] == ["x.GFB.npy"]
    assert iopipe.npy_complete(out)
    raw = open(out, "rb").read()
    open(out, "wb").write(raw[:-8])
    assert not iopipe.npy_complete(out)
    open(out, "wb").write(b"not an npy file")
    assert not iopipe.npy_complete(out)

    class Unsavable:            # numpy.save raises half way: nothing appears under the final name, no temp file stays
        def __array__(self, *a, **k):
            raise RuntimeError("boom")
    import pytest
    with pytest.raises(Exception):
        iopipe.save_npy_atomic(str(tmp_path / "y"), Unsavable())
    assert sorted(f.name for f in tmp_path.iterdir()) == ["x.GFB.npy"]
    # JobReport.pending: stale settings disable the skipping
    rep = iopipe.JobReport("prepare envelope", skip_existing=True)
    src = tmp_path / "in.GFB.npy"
    np.save(src, a)
    iopipe.save_npy_atomic(str(tmp_path / "in.ENV1"), a)
    outs = lambda name: [str(tmp_path / "in.ENV1.npy")]
    assert rep.pending([str(src)], outs, params={"cutoff": 50}, stamp_dir=str(tmp_path)) == []
    rep2 = iopipe.JobReport("prepare envelope", skip_existing=True)
    assert rep2.pending([str(src)], outs, params={"cutoff": 100}, stamp_dir=str(tmp_path)) == [str(src)]
    # that run dies before it rewrites the output: a resume with the SAME new settings must not trust the old output
    # (round-3 advisor finding: the stamp used to be overwritten at the start, so the resume skipped everything)
    rep3 = iopipe.JobReport("prepare envelope", skip_existing=True)
    assert rep3.pending([str(src)], outs, params={"cutoff": 100}, stamp_dir=str(tmp_path)) == [str(src)]
    import time
    time.sleep(0.02)
    iopipe.save_npy_atomic(str(tmp_path / "in.ENV1"), a)              # ... now it is recomputed
    rep4 = iopipe.JobReport("prepare envelope", skip_existing=True)
    assert rep4.pending([str(src)], outs, params={"cutoff": 100}, stamp_dir=str(tmp_path)) == [] and rep4.skipped == 1
    assert sorted(f.name for f in tmp_path.iterdir() if f.name.startswith(".f2cnn")) == [".f2cnn_prepare_envelope.json"]


def test_keras_hdf5_model_is_read_through_h5py_when_present(tmp_path, monkeypatch):
    """`last_trained_model` as the reference writes it (keras model.save, Training.py:139) is an HDF5 file. The loader goes
    through h5py; the image has none, so a stand-in with h5py's File / Group / Dataset / attrs interface holds a model the way
    Keras 2.2 lays it out (model_weights/<layer>/<layer>/kernel:0, layer_names / weight_names attributes as bytes, layers
    without weights in between). Without h5py the loader says how to convert instead of failing obscurely."""
    import sys
    import types
    import numpy as np
    import pytest
    from f2cnn_amd import model as mdl
    ref = mdl.F2CNNModel.glorot(11, zero_bias=False)
    path = tmp_path / "last_trained_model"
    path.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)

    class Node(dict):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.attrs = {}

    def keras_layout():
        root, mw = Node(), Node()
        root["model_weights"] = mw
        root["optimizer_weights"] = Node()
        layers = [("conv2d_1", "conv1"), ("conv2d_2", "conv2"), ("max_pooling2d_1", None), ("dropout_1", None),
                  ("conv2d_3", "conv3"), ("conv2d_4", "conv4"), ("max_pooling2d_2", None), ("dropout_2", None),
                  ("flatten_1", None), ("dense_1", "dense1"), ("dropout_3", None), ("dense_2", "dense2")]
        mw.attrs["layer_names"] = np.array([n.encode() for n, _ in layers])
        for lname, ours in layers:
            g = Node()
            mw[lname] = g
            if ours is None:
                g.attrs["weight_names"] = np.array([], dtype="S1")
                continue
            g.attrs["weight_names"] = np.array([f"{lname}/kernel:0".encode(), f"{lname}/bias:0".encode()])
            inner = Node({"kernel:0": ref.tensors[ours + "_w"], "bias:0": ref.tensors[ours + "_b"]})
            g[lname] = inner

        def getitem(node, key):      # h5py resolves 'a/b' paths
            for part in key.split("/"):
                node = dict.__getitem__(node, part)
            return node
        Node.__getitem__ = getitem
        return root

    class File:
        def __init__(self, p, mode="r"):
            assert str(p) == str(path) and mode == "r"
            self.root = keras_layout()

        def __enter__(self):
            return self.root

        def __exit__(self, *exc):
            return False
    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    got = mdl.load_model(str(path))
    for k, v in ref.tensors.items():
        np.testing.assert_array_equal(got.tensors[k], v)
    monkeypatch.setitem(sys.modules, "h5py", None)           # import h5py -> ImportError
    with pytest.raises(ImportError, match="f2cnn_amd.model"):
        mdl.load_model(str(path))
