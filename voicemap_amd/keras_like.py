"""The small slice of the Keras 2.2.2 API that the reference's experiment scripts touch, re-implemented around the HIP
engine: layer descriptors (``Dense``), ``Adam``, ``Sequence``, ``to_categorical`` and the four callbacks
(``CSVLogger``, ``ModelCheckpoint``, ``ReduceLROnPlateau`` + the ``Callback`` base) with the semantics the scripts rely
on (experiments/train_siamese.py:56-93, experiments/train_classifier.py:110-151).  Nothing here does model arithmetic;
it is bookkeeping around ``HipEncoderEngine``.
"""
from __future__ import annotations

import csv
import os
import queue
import threading
from collections import OrderedDict
from typing import Iterable, Optional

import numpy as np


# ---------------------------------------------------------------------------------------------------------
# layers / optimizer descriptors
# ---------------------------------------------------------------------------------------------------------
class Layer:
    def __init__(self, name: str, **config):
        self.name = name
        self.config = config

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % kv for kv in self.config.items()))


class InputLayer(Layer):
    pass


class Conv1D(Layer):
    pass


class BatchNormalization(Layer):
    pass


class SpatialDropout1D(Layer):
    pass


class MaxPool1D(Layer):
    pass


class GlobalMaxPool1D(Layer):
    pass


class Subtract(Layer):
    pass


class Lambda(Layer):
    pass


class Dense(Layer):
    """keras.layers.Dense(units, activation=None).  Only what the scripts add: a linear embedding layer inside the
    encoder, the 1-unit sigmoid siamese head and ``Dense(num_classes, activation='softmax')``
    (experiments/train_classifier.py:112)."""

    def __init__(self, units, activation=None, name=None):
        super().__init__(name or "dense", units=int(units), activation=activation)
        self.units = int(units)
        self.activation = activation


class Adam:
    """keras.optimizers.Adam: lr 1e-3, beta_1 0.9, beta_2 0.999, epsilon None -> K.epsilon() = 1e-7, decay 0,
    clipnorm as a global norm over all gradients (standalone Keras 2.2.2)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, amsgrad=False, clipnorm=None,
                 clipvalue=None):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference and not implemented")
        if clipvalue is not None:
            raise NotImplementedError("clipvalue is not used by the reference and not implemented")
        self.lr, self.beta_1, self.beta_2 = float(lr), float(beta_1), float(beta_2)
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)
        self.decay = float(decay)
        self.clipnorm = None if clipnorm is None else float(clipnorm)

    def apply_to(self, engine):
        engine.lr, engine.beta_1, engine.beta_2 = self.lr, self.beta_1, self.beta_2
        engine.adam_eps, engine.decay = self.epsilon, self.decay
        engine.clipnorm = self.clipnorm or 0.0


def to_categorical(y, num_classes=None):
    """keras.utils.to_categorical: integer class vector -> one-hot float32 matrix."""
    y = np.asarray(y, dtype="int64").ravel()
    if num_classes is None:
        num_classes = int(y.max()) + 1
    out = np.zeros((y.shape[0], num_classes), dtype=np.float32)
    out[np.arange(y.shape[0]), y] = 1.0
    return out


class Sequence:
    """keras.utils.Sequence: ``__getitem__``/``__len__`` (+ optional ``on_epoch_end``)."""

    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def on_epoch_end(self):
        pass


# ---------------------------------------------------------------------------------------------------------
# batch producers: generator or Sequence -> iterator with optional background prefetch
# ---------------------------------------------------------------------------------------------------------
class BatchFeeder:
    """What ``fit_generator(workers=, use_multiprocessing=, max_queue_size=)`` provides: batches produced ahead of the
    training loop.  Sequences are indexed in order (shuffle off, like the scripts' defaults for a custom Sequence);
    plain generators are advanced under a lock.  Threads are used (the producers are numpy/IO code that releases the GIL);
    ``workers=0`` produces in the caller's thread."""

    def __init__(self, source, workers: int = 1, max_queue_size: int = 10):
        self.source = source
        self.is_sequence = hasattr(source, "__getitem__") and hasattr(source, "__len__")
        self.workers = max(0, int(workers))
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, max_queue_size))
        self.lock = threading.Lock()
        self.stop = threading.Event()
        self.cursor = 0
        self.threads = []
        self.it = None if self.is_sequence else iter(source)
        for _ in range(min(self.workers, 8)):
            t = threading.Thread(target=self._work, daemon=True)
            t.start()
            self.threads.append(t)

    def _produce(self):
        if self.is_sequence:
            with self.lock:
                i = self.cursor
                self.cursor += 1
                n = len(self.source)
                if n > 0 and self.cursor % n == 0:
                    self.source.on_epoch_end() if hasattr(self.source, "on_epoch_end") else None
            return self.source[i % max(len(self.source), 1)]
        with self.lock:
            return next(self.it)

    def _work(self):
        while not self.stop.is_set():
            try:
                item = self._produce()
            except StopIteration:
                self.q.put(StopIteration)
                return
            except Exception as e:  # surface producer errors in the consumer
                self.q.put(e)
                return
            while not self.stop.is_set():
                try:
                    self.q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue

    def get(self):
        if not self.threads:
            return self._produce()
        item = self.q.get()
        if item is StopIteration:
            raise StopIteration
        if isinstance(item, Exception):
            raise item
        return item

    def close(self):
        """Stop the producer threads and wait for them: a producer that is still sampling would otherwise keep consuming
        the global numpy RNG after fit_generator has returned."""
        self.stop.set()
        for t in self.threads:
            while t.is_alive():
                try:  # unblock a producer waiting on a full queue
                    self.q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
        self.threads = []


# ---------------------------------------------------------------------------------------------------------
# callbacks
# ---------------------------------------------------------------------------------------------------------
class Callback:
    rank0_only = False  # data-parallel fit_generator: callbacks that write files run on rank 0 only

    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class CSVLogger(Callback):
    """keras.callbacks.CSVLogger: one row per epoch, columns = 'epoch' + sorted log keys
    (epoch, acc, loss, lr, val_1-shot_acc, val_acc, val_loss for the siamese script)."""
    rank0_only = True

    def __init__(self, filename, separator=",", append=False):
        super().__init__()
        self.filename, self.sep, self.append = filename, separator, append
        self.keys = None
        self.file = None
        self.writer = None

    def on_train_begin(self, logs=None):
        d = os.path.dirname(self.filename)
        if d:
            os.makedirs(d, exist_ok=True)
        self.file = open(self.filename, "a" if self.append else "w", newline="")

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
            self.writer = csv.DictWriter(self.file, fieldnames=["epoch"] + self.keys, delimiter=self.sep)
            if not self.append or self.file.tell() == 0:
                self.writer.writeheader()
        row = OrderedDict(epoch=epoch)
        row.update((k, logs.get(k, "NA")) for k in self.keys)
        self.writer.writerow(row)
        self.file.flush()

    def on_train_end(self, logs=None):
        if self.file:
            self.file.close()
            self.file = None


class ModelCheckpoint(Callback):
    """keras.callbacks.ModelCheckpoint(filepath, monitor, mode, save_best_only, verbose).  Saves the full model
    (weights + Adam slots + BN moving statistics) with ``model.save``: a Keras-2.2.2 HDF5 file for ``*.hdf5`` / ``*.h5`` names
    (voicemap_amd/keras_hdf5.py), this package's ``.npz`` container otherwise."""
    rank0_only = True

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False, mode="auto",
                 period=1):
        super().__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.period = save_best_only, period
        if mode == "auto":
            mode = "max" if ("acc" in monitor or monitor.startswith("fmeasure")) else "min"
        self.better = (lambda a, b: a > b) if mode == "max" else (lambda a, b: a < b)
        self.best = -np.inf if mode == "max" else np.inf
        self.since = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.since += 1
        if self.since < self.period:
            return
        self.since = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                print("Can save best model only with %s available, skipping." % self.monitor)
                return
            if not self.better(cur, self.best):
                return
            if self.verbose:
                print("Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s" % (epoch + 1, self.monitor, self.best,
                                                                                         cur, path))
            self.best = cur
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        self.model.save(path)


class ReduceLROnPlateau(Callback):
    """keras.callbacks.ReduceLROnPlateau defaults: factor 0.1, patience 10, min_delta 1e-4, cooldown 0, min_lr 0."""

    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4, cooldown=0, min_lr=0):
        super().__init__()
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        if mode == "auto":
            mode = "max" if "acc" in monitor else "min"
        if mode == "max":
            self.better, self.best = (lambda a, b: a > b + self.min_delta), -np.inf
        else:
            self.better, self.best = (lambda a, b: a < b - self.min_delta), np.inf
        self.wait = 0
        self.cooldown_counter = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs["lr"] = self.model.get_lr()
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.better(cur, self.best):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = self.model.get_lr()
                if old > self.min_lr:
                    new = max(old * self.factor, self.min_lr)
                    self.model.set_lr(new)
                    if self.verbose:
                        print("\nEpoch %05d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, new))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


def run_callbacks(callbacks: Iterable[Callback], method: str, *args):
    for cb in callbacks:
        getattr(cb, method)(*args)
