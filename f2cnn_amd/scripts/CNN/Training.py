"""Counterpart of the reference's ``scripts/CNN/Training.py``.

``normalizeInput`` (:13-28) is on the hot path and runs through HIP kernel K3. ``SeparateTestTrain`` (:31-44) and
``TrainAndPlotLoss`` (:47-156) are SURVEY section 8f row n4: the same network (:93-114; weights in
``f2cnn_amd.model``), RMSprop(lr 1e-4, decay 1e-6), categorical cross-entropy, early stopping on the validation
accuracy (min_delta 0.01, patience 5), trained with PyTorch-ROCm autograd as the scope table asks. Training only
produces weights; every forward pass used for evaluation is HIP kernel K4 (the trained weights are written in the
``.npz`` container K4 loads, under the reference's file name ``last_trained_model``).
"""
import csv
import json
import os

import numpy

from ... import _lib


def normalizeInput(matrix, ctx=None):
    """Per-window log min-max normalisation: (ln x - ln min)/(ln max - ln min); a constant window gives
    zeros; any value <= 0 raises ValueError("values must all be positive"). Same dtype and shape out.
    Runs on the GPU through the window kernel (one window, step 1)."""
    ctx = ctx or _lib.default_context()
    m = numpy.asarray(matrix)
    shape, dtype = m.shape, m.dtype
    if m.ndim < 2:
        raise ValueError("matrix must be at least two dimensional")
    rows = shape[0]
    flat = numpy.ascontiguousarray(m.reshape(rows, -1), dtype=numpy.float64)
    if rows % 2 == 0:   # the kernel takes 2*radius+1 rows; pad an even window with a copy of its last row
        flat = numpy.vstack([flat, flat[-1:]])
    R, Cn = flat.shape
    env = numpy.ascontiguousarray(flat.T)           # (C, N=R): env[c, k]
    out = numpy.empty((1, R, Cn), numpy.float32)
    try:
        ctx.gather_windows(env, Cn, R, numpy.array([R // 2], numpy.int64), 1, R // 2, 1, True, out, _lib.MEM_HOST)
    except _lib.F2Error as e:
        if e.code == _lib.F2_ERR_NONPOSITIVE:
            print(shape)
            raise ValueError("values must all be positive")
        raise
    res = out[0, :rows].reshape(shape)
    return res.astype(dtype) if numpy.issubdtype(dtype, numpy.floating) else res.astype(numpy.float64)


def normalizeInputBatch(windows, ctx=None):
    """normalizeInput applied to every (rows, C) window of an (n, rows, C) array in one K3 launch; float32 out.
    (The reference loops over the windows, Training.py:72-75. It casts to float32 *before* taking logarithms there;
    K3 takes them in float64 and rounds once, the convention of the evaluation path, Evaluating.py:66-68.)"""
    ctx = ctx or _lib.default_context()
    w = numpy.asarray(windows)
    if w.ndim == 4 and w.shape[-1] == 1:
        w = w[..., 0]
    if w.ndim != 3 or w.shape[1] % 2 == 0:
        raise ValueError("expected (n, 2*radius+1, channels) windows")
    n, R, Cn = w.shape
    out = numpy.empty((n, R, Cn), numpy.float32)
    if n == 0:
        return out
    env = numpy.ascontiguousarray(w.reshape(n * R, Cn).T, dtype=numpy.float64)     # (C, n*R): the windows end to end
    centers = numpy.arange(n, dtype=numpy.int64) * R + R // 2
    try:
        ctx.gather_windows(env, Cn, n * R, centers, n, R // 2, 1, True, out, _lib.MEM_HOST)
    except _lib.F2Error as e:
        if e.code == _lib.F2_ERR_NONPOSITIVE:
            raise ValueError("values must all be positive")
        raise
    return out


def SeparateTestTrain(pathToInput, pathToLabel):
    """(x_test, y_test, x_train, y_train): row i of the label CSV goes with input_data[i]; first column 'TEST'
    selects the test set, the last column is the 0/1 sign (Training.py:31-44)."""
    x = [[], []]
    y = [[], []]
    input_data = numpy.load(pathToInput)
    with open(pathToLabel, 'r') as labels:
        for i, row in enumerate(csv.reader(labels)):
            test, sign = row[0], row[8]
            k = 0 if test == 'TEST' else 1
            x[k].append(input_data[i])
            y[k].append(int(sign))
    return numpy.array(x[0]), numpy.array(y[0]), numpy.array(x[1]), numpy.array(y[1])


def build_network(rows=11, channels=128):
    """The Sequential model of Training.py:93-114 as a torch module (NCHW); layer names match
    F2CNNModel.from_torch_state_dict. Keras defaults: glorot_uniform kernels, zero biases."""
    import torch
    from torch import nn
    from ...model import flatten_size

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(1, 32, 3, padding=1)      # padding='same'
            self.conv2 = nn.Conv2d(32, 32, 3)
            self.conv3 = nn.Conv2d(32, 64, 3, padding=1)
            self.conv4 = nn.Conv2d(64, 64, 3)
            self.dense1 = nn.Linear(flatten_size(rows, channels)[2], 516)
            self.dense2 = nn.Linear(516, 2)
            self.drop1, self.drop2, self.drop3 = nn.Dropout(0.25), nn.Dropout(0.25), nn.Dropout(0.5)
            for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.dense1, self.dense2):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

        def forward(self, x):                                  # logits; softmax lives in the loss / K4
            F = torch.nn.functional
            x = F.relu(self.conv1(x))
            x = self.drop1(F.max_pool2d(F.relu(self.conv2(x)), 2))
            x = F.relu(self.conv3(x))
            x = self.drop2(F.max_pool2d(F.relu(self.conv4(x)), 2))
            x = self.drop3(F.relu(self.dense1(torch.flatten(x, 1))))
            return self.dense2(x)

    return Net()


def train_network(x_train, y_train, x_test, y_test, batch_size=32, epochs=20, device=None, seed=None, verbose=1,
                  lr=1e-4, decay=1e-6, min_delta=0.01, patience=5):
    """model.compile + model.fit of Training.py:117-135 on already normalised float32 windows (n, rows, C).
    Returns (F2CNNModel, history) with history = {'loss','acc','val_loss','val_acc'} per epoch.
    RMSprop as Keras 2.2 runs it: rho 0.9, epsilon 1e-7, lr_t = lr / (1 + decay * iterations)."""
    import torch
    from ...model import F2CNNModel
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("cnn train needs a GPU (PyTorch-ROCm device); pass device='cpu' explicitly to override")
        device = "cuda"
    if seed is not None:
        torch.manual_seed(seed)
    x_train = numpy.asarray(x_train, numpy.float32)
    x_test = numpy.asarray(x_test, numpy.float32)
    rows, channels = x_train.shape[1], x_train.shape[2]
    net = build_network(rows, channels).to(device)
    xt = torch.from_numpy(x_train).unsqueeze(1).to(device)
    yt = torch.from_numpy(numpy.asarray(y_train, numpy.int64)).to(device)
    xv = torch.from_numpy(x_test).unsqueeze(1).to(device)
    yv = torch.from_numpy(numpy.asarray(y_test, numpy.int64)).to(device)
    opt = torch.optim.RMSprop(net.parameters(), lr=lr, alpha=0.9, eps=1e-7)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0 / (1.0 + decay * it))
    loss_fn = torch.nn.CrossEntropyLoss(reduction="sum")

    def evaluate(x, y):
        net.eval()
        tot, hit = 0.0, 0
        with torch.no_grad():
            for i in range(0, len(x), 1024):
                lg = net(x[i:i + 1024])
                tot += float(loss_fn(lg, y[i:i + 1024]))
                hit += int((lg.argmax(1) == y[i:i + 1024]).sum())
        n = max(len(x), 1)
        return tot / n, hit / n

    history = {"loss": [], "acc": [], "val_loss": [], "val_acc": []}
    best, wait = -numpy.inf, 0
    for epoch in range(epochs):
        net.train()
        perm = torch.randperm(len(xt), device=device)          # fit(shuffle=True)
        tot, hit = 0.0, 0
        for i in range(0, len(xt), batch_size):
            idx = perm[i:i + batch_size]
            lg = net(xt[idx])
            loss = loss_fn(lg, yt[idx])
            opt.zero_grad(set_to_none=True)
            (loss / len(idx)).backward()
            opt.step()
            sched.step()
            tot += float(loss.detach())
            hit += int((lg.argmax(1) == yt[idx]).sum())
        vl, va = evaluate(xv, yv) if len(xv) else (float("nan"), float("nan"))
        for k, val in zip(("loss", "acc", "val_loss", "val_acc"), (tot / len(xt), hit / len(xt), vl, va)):
            history[k].append(val)
        if verbose:
            print("Epoch {}/{} - loss: {:.4f} - acc: {:.4f} - val_loss: {:.4f} - val_acc: {:.4f}".format(
                epoch + 1, epochs, history["loss"][-1], history["acc"][-1], vl, va))
        # keras.callbacks.EarlyStopping(monitor='val_acc', min_delta=0.01, patience=5, mode='auto')
        if va - min_delta > best:
            best, wait = va, 0
        else:
            wait += 1
            if wait >= patience:
                if verbose:
                    print("Epoch {:05d}: early stopping".format(epoch + 1))
                break
    model = F2CNNModel.from_torch_state_dict(net.state_dict(), rows, channels)
    return model, history


def TrainAndPlotLoss(labelFile=None, inputFile=None, device=None, seed=None):
    """Trains the CNN on an input tensor (N x 11 x 128, ``prepare input``) and its label CSV (``prepare label``);
    BATCH_SIZE and EPOCHS come from configF2CNN.conf. Saves the weights as 'last_trained_model' (the .npz container
    ``cnn eval`` loads) and the per-epoch history as 'last_trained_model_results.json' (the reference plots it)."""
    from configparser import ConfigParser
    config = ConfigParser()
    config.read('configF2CNN.conf')
    batch_size = config.getint('CNN', 'BATCH_SIZE', fallback=32)
    epochs = config.getint('CNN', 'EPOCHS', fallback=20)
    inputPath = inputFile or os.path.join('trainingData', 'last_input_data.npy')
    labelPath = labelFile or os.path.join('trainingData', 'label_data.csv')
    x_test, y_test, x_train, y_train = SeparateTestTrain(inputPath, labelPath)
    x_train = normalizeInputBatch(x_train)
    x_test = normalizeInputBatch(x_test) if len(x_test) else numpy.empty((0,) + x_train.shape[1:], numpy.float32)
    print('Rising test:', int((y_test == 1).sum()))
    print('Falling test:', int((y_test == 0).sum()))
    print('Rising train:', int((y_train == 1).sum()))
    print('Falling train:', int((y_train == 0).sum()))
    print(x_train.shape, 'train samples')
    print(x_test.shape, 'test samples')
    print("Categories: [falling, rising]")
    model, history = train_network(x_train, y_train, x_test, y_test, batch_size, epochs, device=device, seed=seed)
    model.save('last_trained_model')
    print("Model saved as 'last_trained_model'.")
    # model.evaluate on the test set, through the HIP forward pass that `cnn eval` uses
    if len(x_test):
        scores, labels = model.predict_labels(x_test)
        p = numpy.clip(scores[numpy.arange(len(y_test)), y_test].astype(numpy.float64), 1e-7, 1.0)
        print('Test loss:', float(-numpy.log(p).mean()))
        print('Test accuracy:', float((labels == y_test).mean()))
    with open('last_trained_model_results.json', 'w') as fp:
        json.dump(history, fp)
    print("History saved as 'last_trained_model_results.json'")
    return model, history
