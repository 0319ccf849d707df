"""Weight container for the F2CNN network (architecture: reference scripts/CNN/Training.py:93-114).

The reference stores a Keras HDF5 model (``last_trained_model``, Training.py:139). Here weights travel as an ``.npz``
holding the 12 tensors in their Keras layouts:

    conv{1..4}_w (3,3,Cin,Cout)  conv{1..4}_b (Cout,)  dense1_w (F,516) dense1_b  dense2_w (516,2) dense2_b

A genuine Keras ``last_trained_model`` (HDF5) is read through h5py where that package exists (``F2CNNModel.load``
recognises the HDF5 signature; ``python -m f2cnn_amd.model last_trained_model out.npz`` converts once, on the machine
that trained the model). h5py is not part of this image, so the reader is exercised against a stand-in that mimics
h5py's File / Group / Dataset interface (tests/test_host_io_cli.py), not against a file Keras wrote.

A PyTorch ``state_dict`` (Conv2d OIHW / Linear (out,in), NCHW flatten) is converted by
``F2CNNModel.from_torch_state_dict``. The forward pass itself is HIP kernel K4 (``f2_cnn_forward``).
"""
import os

import numpy as np

from . import _lib

NAMES = ("conv1", "conv2", "conv3", "conv4", "dense1", "dense2")
CONV_CH = ((1, 32), (32, 32), (32, 64), (64, 64))


def flatten_size(rows, channels):
    hp2 = ((rows - 2) // 2 - 2) // 2
    wp2 = ((channels - 2) // 2 - 2) // 2
    return hp2, wp2, max(hp2, 0) * max(wp2, 0) * 64


def tensor_shapes(rows=11, channels=128):
    flat = flatten_size(rows, channels)[2]
    shapes = {}
    for name, (ci, co) in zip(NAMES[:4], CONV_CH):
        shapes[name + "_w"], shapes[name + "_b"] = (3, 3, ci, co), (co,)
    shapes["dense1_w"], shapes["dense1_b"] = (flat, 516), (516,)
    shapes["dense2_w"], shapes["dense2_b"] = (516, 2), (2,)
    return shapes


class F2CNNModel:
    def __init__(self, tensors, rows=11, channels=128):
        self.rows, self.channels = int(rows), int(channels)
        shapes = tensor_shapes(rows, channels)
        self.tensors = {}
        for k, shp in shapes.items():
            if k not in tensors:
                raise KeyError(f"missing weight tensor {k}")
            a = np.ascontiguousarray(tensors[k], dtype=np.float32)
            if a.shape != shp:
                raise ValueError(f"{k}: expected shape {shp}, got {a.shape}")
            self.tensors[k] = a
        self._handles = {}

    # ---- construction ----
    @classmethod
    def glorot(cls, seed=7, rows=11, channels=128, zero_bias=True):
        """Random-init weights (Keras default glorot_uniform), for tests and synthetic benchmarks."""
        rng = np.random.default_rng(seed)
        t = {}
        for k, shp in tensor_shapes(rows, channels).items():
            if k.endswith("_w"):
                fan_in = int(np.prod(shp[:-1]))
                fan_out = shp[-1] * (9 if len(shp) == 4 else 1)
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                t[k] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
            else:
                t[k] = np.zeros(shp, np.float32) if zero_bias else rng.uniform(-0.05, 0.05, size=shp).astype(np.float32)
        return cls(t, rows, channels)

    @classmethod
    def from_keras_hdf5(cls, path, rows=11, channels=128):
        """Weights of a model saved by the reference (keras model.save / save_weights, Training.py:139): HDF5 group
        `model_weights` (or the file root for save_weights), attribute `layer_names`, per layer `weight_names`
        ('<layer>/kernel:0', '<layer>/bias:0'). The six layers that carry weights - Conv2D x 4, Dense x 2, in model order -
        are taken in order whatever their names; Keras stores exactly the layouts this container uses."""
        try:
            import h5py
        except ImportError as exc:
            raise ImportError(f"{path} is a Keras HDF5 model; reading it needs h5py, which is not installed here. Convert it "
                              "once where it is: python -m f2cnn_amd.model {0} {0}.npz".format(path)) from exc

        def text(v):
            return v.decode() if isinstance(v, bytes) else str(v)
        with h5py.File(path, "r") as f:
            g = f["model_weights"] if "model_weights" in f else f
            pairs = []
            for lname in [text(v) for v in g.attrs["layer_names"]]:
                names = [text(v) for v in g[lname].attrs.get("weight_names", [])]
                if not names:
                    continue                                   # pooling / dropout / flatten layers
                kern = [n for n in names if n.split("/")[-1].startswith("kernel")]
                bias = [n for n in names if n.split("/")[-1].startswith("bias")]
                if len(kern) != 1 or len(bias) != 1:
                    raise ValueError(f"layer {lname}: expected one kernel and one bias, found {names}")
                pairs.append((np.asarray(g[lname][kern[0]]), np.asarray(g[lname][bias[0]])))
        if len(pairs) != len(NAMES):
            raise ValueError(f"{path}: {len(pairs)} layers with weights, the F2CNN network has {len(NAMES)}")
        t = {}
        for name, (k, b) in zip(NAMES, pairs):
            t[name + "_w"], t[name + "_b"] = k, b
        return cls(t, rows, channels)

    @classmethod
    def load(cls, path, rows=11, channels=128):
        if not os.path.exists(path) and os.path.exists(str(path) + ".npz"):
            path = str(path) + ".npz"
        with open(path, "rb") as fp:
            magic = fp.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            return cls.from_keras_hdf5(path, rows, channels)
        z = np.load(path)
        rows = int(z["rows"]) if "rows" in z else 11
        channels = int(z["channels"]) if "channels" in z else 128
        return cls({k: z[k] for k in tensor_shapes(rows, channels)}, rows, channels)

    def save(self, path):
        """Writes exactly `path` (no .npz appended), so 'last_trained_model' keeps the reference's file name."""
        with open(path, "wb") as fp:
            np.savez(fp, rows=self.rows, channels=self.channels, **self.tensors)

    @classmethod
    def from_torch_state_dict(cls, sd, rows=11, channels=128):
        """Keys conv{1..4}.weight/bias (OIHW), dense{1,2}.weight/bias ((out,in), input flattened C,H,W)."""
        t = {}
        for name in NAMES[:4]:
            t[name + "_w"] = np.transpose(np.asarray(sd[name + ".weight"].detach().cpu()), (2, 3, 1, 0))
            t[name + "_b"] = np.asarray(sd[name + ".bias"].detach().cpu())
        hp2, wp2, flat = flatten_size(rows, channels)
        w1 = np.asarray(sd["dense1.weight"].detach().cpu()).reshape(516, 64, hp2, wp2)   # (out, C, H, W)
        t["dense1_w"] = np.transpose(w1, (2, 3, 1, 0)).reshape(flat, 516)                 # (H, W, C) -> out
        t["dense1_b"] = np.asarray(sd["dense1.bias"].detach().cpu())
        t["dense2_w"] = np.asarray(sd["dense2.weight"].detach().cpu()).T
        t["dense2_b"] = np.asarray(sd["dense2.bias"].detach().cpu())
        return cls(t, rows, channels)

    def ordered(self):
        return [self.tensors[n + s] for n in NAMES for s in ("_w", "_b")]

    # ---- device side ----
    def handle(self, ctx=None):
        ctx = ctx or _lib.default_context()
        h = self._handles.get(id(ctx))
        if h is None:
            h = self._handles[id(ctx)] = (ctx, ctx.cnn_create(self.ordered(), self.rows, self.channels))
        return h[1]

    def predict(self, x, ctx=None):
        """Softmax scores (n,2) float32 for x (n, rows, channels[,1]); like keras model.predict."""
        ctx = ctx or _lib.default_context()
        x = np.asarray(x)
        if x.ndim == 4 and x.shape[-1] == 1:
            x = x[..., 0]
        if x.ndim != 3 or x.shape[1:] != (self.rows, self.channels):
            raise ValueError(f"expected input of shape (n,{self.rows},{self.channels}[,1]), got {x.shape}")
        x = np.ascontiguousarray(x, dtype=np.float32)    # Keras casts its float64 input to float32
        scores = np.empty((x.shape[0], 2), np.float32)
        ctx.cnn_forward(self.handle(ctx), x, x.shape[0], scores, None, _lib.MEM_HOST)
        return scores

    def predict_labels(self, x, ctx=None):
        ctx = ctx or _lib.default_context()
        x = np.asarray(x)
        if x.ndim == 4:
            x = x[..., 0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        scores = np.empty((x.shape[0], 2), np.float32)
        labels = np.empty(x.shape[0], np.uint8)
        ctx.cnn_forward(self.handle(ctx), x, x.shape[0], scores, labels, _lib.MEM_HOST)
        return scores, labels


def load_model(path):
    """Counterpart of keras.models.load_model: the .npz container, or a Keras HDF5 file where h5py is installed."""
    return F2CNNModel.load(path)


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 3:
        raise SystemExit("usage: python -m f2cnn_amd.model <keras last_trained_model (HDF5)> <out.npz>")
    m = F2CNNModel.from_keras_hdf5(sys.argv[1])
    m.save(sys.argv[2])
    print("wrote", sys.argv[2], {k: v.shape for k, v in m.tensors.items()})
