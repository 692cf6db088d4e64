"""Drop-in mirror of ``voicemap/models.py`` of the reference: the same two build functions with the same signatures
and error behaviour, returning Keras-like objects (``compile / fit_generator / predict / layers / summary / save /
get_weights / set_weights / add / pop``) whose arithmetic runs on the HIP path (``HipEncoderEngine``).

    get_baseline_convolutional_encoder(filters, embedding_dimension, input_shape=None, dropout=0.05)   models.py:6
    build_siamese_net(encoder, input_shape, distance_metric='uniform_euclidean')                        models.py:44

Extra keyword ``dtype`` selects storage and GEMM arithmetic: 'f16' (default: IEEE half tensors, f16 MFMAs, loss-scaled gradients --
embeddings within 1e-3 of the fp32 arithmetic), 'bf16' (the same kernels with bf16 tensors: 2 % faster, 6e-3),
'f32' (fp32 tensors, fp32 MFMAs: the exact-parity mode) or 'f32s' (fp32 tensors, split-bf16 products in the k=3 conv GEMMs: ~1e-5 of
'f32' at ~0.55x its step time; section 4.6a).
"""
from __future__ import annotations

import json
import time
from collections import OrderedDict
from typing import List, Optional, Sequence

import numpy as np

from . import keras_like as K
from .keras_like import Adam, Dense


def _is_lazy(x):
    from .utils import LazyWindows
    return isinstance(x, LazyWindows)


# =========================================================================================================
class _TrainableModel:
    """Shared compile / fit / save plumbing of the encoder-classifier and the siamese model."""

    def __init__(self):
        self.engine = None
        self.optimizer: Optional[Adam] = None
        self.loss = None
        self.metrics: List[str] = []
        self.stop_training = False
        self._pending_weights = None
        self._pending_adam = None
        self.history = None
        self.defer_batch_logs = True   # fit_generator: read the per-batch loss / acc in blocks when no callback has an on_batch_end

    # ---- engine lifecycle -------------------------------------------------------------------------------
    def _engine_args(self):
        raise NotImplementedError

    def _make_engine(self):
        from .engine import HipEncoderEngine
        return HipEncoderEngine(**self._engine_args())

    def _ensure_engine(self):
        if self.engine is None:
            self.engine = self._make_engine()
            if self._pending_weights is not None:
                self.engine.set_params(self._pending_weights)
                self._pending_weights = None
            if self._pending_adam is not None:
                self._set_adam_state(self._pending_adam)
                self._pending_adam = None
            if self.optimizer is not None:
                self.optimizer.apply_to(self.engine)
        return self.engine

    # ---- Keras surface ---------------------------------------------------------------------------------
    def compile(self, loss=None, optimizer=None, metrics=None, **_):
        self.loss = loss
        self.metrics = list(metrics or [])
        if isinstance(optimizer, str):
            if optimizer.lower() != "adam":
                raise NotImplementedError("only Adam is implemented (the reference only uses Adam)")
            optimizer = Adam()
        self.optimizer = optimizer or Adam()
        if self.engine is not None:
            self.optimizer.apply_to(self.engine)

    def get_lr(self) -> float:
        return self.engine.lr if self.engine is not None else self.optimizer.lr

    def set_lr(self, lr: float):
        self.optimizer.lr = float(lr)
        if self.engine is not None:
            self.engine.lr = float(lr)

    def weight_names(self) -> List[str]:
        """Keras ``model.weights`` order per layer: conv kernel, bias; BN gamma, beta, moving_mean, moving_variance."""
        eng = self._ensure_engine()
        names = []
        for i in range(eng.nb):
            names += [f"conv{i+1}.kernel", f"conv{i+1}.bias", f"bn{i+1}.gamma", f"bn{i+1}.beta", f"bn{i+1}.moving_mean",
                      f"bn{i+1}.moving_variance"]
        names += ["dense.kernel", "dense.bias"]
        if eng.head is not None:
            names += ["head.kernel", "head.bias"]
        return names

    def get_weights(self):
        p = self._ensure_engine().get_params()
        return [p[k] for k in self.weight_names()]

    def set_weights(self, weights: Sequence[np.ndarray]):
        names = self.weight_names()
        if len(weights) != len(names):
            raise ValueError("expected %d weight arrays, got %d" % (len(names), len(weights)))
        self._ensure_engine().set_params(dict(zip(names, weights)))

    def count_params(self) -> int:
        return self._ensure_engine().n_params

    # ---- persistence ------------------------------------------------------------------------------------
    def _set_adam_state(self, st: dict):
        import torch
        eng = self.engine
        eng.iterations = int(st.get("iterations", 0))
        for slot, buf in (("m", eng.M), ("v", eng.V)):
            for name, val in (st.get(slot) or {}).items():
                v = eng.view(name, buf)
                v.copy_(torch.as_tensor(np.asarray(val, dtype=np.float32)).to(eng.device).reshape(v.shape))

    def _keras_state(self):
        """(kind, geometry, params, optimizer, training) in keras_hdf5's vocabulary."""
        from . import keras_hdf5 as KH
        import torch
        eng = self._ensure_engine()
        torch.cuda.synchronize()
        cfg = self.get_config()
        enc = cfg["encoder"] if cfg["class_name"] == "SiameseNet" else cfg
        kind = "siamese" if cfg["class_name"] == "SiameseNet" else ("classifier" if enc["classifier_units"] else "encoder")
        shape = cfg.get("input_shape") if kind == "siamese" else enc.get("input_shape")
        geo = {"filters": enc["filters"], "embedding_dimension": enc["embedding_dimension"], "dropout": enc["dropout"],
               "first_pool": enc["first_pool"], "input_shape": tuple(shape) if shape else (None, 1),  # Conv1D takes any length
               "dtype": enc["dtype"],
               "classifier_units": enc["classifier_units"], "distance_metric": cfg.get("distance_metric")}
        names = KH.trainable_names(kind != "encoder")
        opt = {"config": {"lr": eng.lr, "beta_1": eng.beta_1, "beta_2": eng.beta_2, "epsilon": eng.adam_eps, "decay": eng.decay,
                          "amsgrad": False}, "iterations": int(eng.iterations),
               "m": {n: eng.view(n, eng.M).detach().cpu().numpy() for n in names},
               "v": {n: eng.view(n, eng.V).detach().cpu().numpy() for n in names}}
        if eng.clipnorm:
            opt["config"]["clipnorm"] = float(eng.clipnorm)
        loss = self.loss if isinstance(self.loss, str) or self.loss is None else getattr(self.loss, "__name__", str(self.loss))
        return kind, geo, eng.get_params(), opt, {"loss": loss, "metrics": list(self.metrics)}

    def save(self, filepath: str):
        """``model.save``: a Keras-2.2.2 HDF5 file for ``*.hdf5`` / ``*.h5`` names (what the reference's ModelCheckpoint
        writes, keras_hdf5.py), otherwise one .npz with weights, Adam slots, moving statistics and the model config."""
        import torch
        if str(filepath).lower().endswith((".hdf5", ".h5")):
            enc = getattr(self, "encoder", self)
            if type(enc).__name__ == "SpectrogramEncoder":
                raise NotImplementedError("Keras HDF5 checkpoints describe the reference's 1-D encoder; save the spectrogram variant as .npz")
            from . import keras_hdf5 as KH
            kind, geo, params, opt, training = self._keras_state()
            KH.write_checkpoint(filepath, kind, geo, params, opt, training)
            return
        eng = self._ensure_engine()
        torch.cuda.synchronize()
        cfg = dict(self.get_config())
        loss = self.loss if isinstance(self.loss, str) or self.loss is None else getattr(self.loss, "__name__", str(self.loss))
        cfg["training"] = {"loss": loss, "metrics": list(self.metrics),
                           "optimizer": {"lr": eng.lr, "beta_1": eng.beta_1, "beta_2": eng.beta_2, "epsilon": eng.adam_eps,
                                         "decay": eng.decay, "clipnorm": float(eng.clipnorm) if eng.clipnorm else None}}
        blob = {"P": eng.P.cpu().numpy(), "M": eng.M.cpu().numpy(), "V": eng.V.cpu().numpy(), "NT": eng.NT.cpu().numpy(),
                "ZD": eng.ZD.cpu().numpy(), "bn_steps": np.int64(eng.bn_steps),   # zero-debias accumulators of the moving statistics
                "iterations": np.int64(eng.iterations),
                "config": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
        with open(filepath, "wb") as f:
            np.savez(f, **blob)

    def _load_state(self, blob):
        import torch
        eng = self._ensure_engine()
        for name in ("P", "M", "V", "NT"):
            getattr(eng, name).copy_(torch.from_numpy(blob[name]).to(eng.device))
        if "ZD" in blob:
            eng.ZD.copy_(torch.from_numpy(blob["ZD"]).to(eng.device))
            eng.bn_steps = int(blob["bn_steps"])
        eng.iterations = int(blob["iterations"])
        eng.refresh_weights()

    def save_weights(self, filepath: str):
        """``model.save_weights``: HDF5 in Keras' layout (weights only; the model_weights group of ``save``)."""
        from . import keras_hdf5 as KH
        kind, geo, params, _, _ = self._keras_state()
        KH.write_checkpoint(filepath, kind, geo, params, None, None)

    def load_weights(self, filepath: str, by_name: bool = False):
        """``model.load_weights`` from a Keras HDF5 file (full model or weights only)."""
        from . import keras_hdf5 as KH
        w = KH.read_weights(filepath)
        want = set(self.weight_names()) if self.engine is not None else None
        if want is not None:
            w = OrderedDict((k, v) for k, v in w.items() if k in want)
        if self.engine is not None:
            self.engine.set_params(w)
        else:
            self._pending_weights = w

    def get_config(self) -> dict:
        raise NotImplementedError

    # ---- training loop (Keras fit_generator semantics used by the scripts) ------------------------------
    def train_on_batch(self, x, y):
        raise NotImplementedError

    def test_on_batch(self, x, y):
        raise NotImplementedError

    def _batch_size(self, x) -> int:
        x0 = x[0] if isinstance(x, (list, tuple)) else x
        return int(x0.shape[0])

    def _evaluate_sums(self, feeder, steps):
        """(samples, sum of loss x batch size, sum of acc x batch size) over ``steps`` batches of a BatchFeeder."""
        tot, wl, wa = 0, 0.0, 0.0
        for _ in range(steps):
            x, y = feeder.get()[:2]
            l, a = self.test_on_batch(x, y)
            n = self._batch_size(x)
            tot += n
            wl += l * n
            wa += a * n
        return tot, wl, wa

    def evaluate_generator(self, generator, steps, workers=1, max_queue_size=10, use_multiprocessing=False):
        """Under torchrun every rank evaluates ceil(steps / world) batches of ITS generator and the sample-weighted means are
        taken over all ranks (one small all-reduce)."""
        from . import parallel
        _, world = parallel.rank_world()
        feeder = generator if isinstance(generator, K.BatchFeeder) else K.BatchFeeder(generator, workers, max_queue_size)
        tot, wl, wa = self._evaluate_sums(feeder, -(-steps // world))
        if not isinstance(generator, K.BatchFeeder):
            feeder.close()
        m = parallel.weighted_mean_logs({"loss": wl, "acc": wa}, tot)
        return [m["loss"], m["acc"]]

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, class_weight=None, max_queue_size=10, workers=1, use_multiprocessing=False,
                      shuffle=True, initial_epoch=0):
        """keras fit_generator as the scripts call it (experiments/train_siamese.py:65-94): `steps_per_epoch` training
        batches, then `validation_steps` validation batches with inference-mode BatchNorm, then the callbacks IN LIST
        ORDER (the n-shot callback must write logs['val_1-shot_acc'] before CSVLogger / ModelCheckpoint /
        ReduceLROnPlateau read it).  Running means of loss / acc are weighted by batch size like Keras.

        Data parallel (new; the reference is single-device): launched under torchrun (one process per GPU, after
        ``parallel.init_distributed()``) every rank runs this same loop on its OWN generator -- global batch = world x
        batchsize -- with the gradient all-reduce hooked into the engine (voicemap_amd/parallel.py); replicas start from
        rank 0's weights.  Epoch metrics are reduced over ranks, so every rank's callbacks see identical logs: callbacks that
        write files (``rank0_only``: CSVLogger, ModelCheckpoint) run on rank 0 only, the others (ReduceLROnPlateau, the n-shot
        evaluation, which shards its tasks over ranks) on every rank."""
        from . import parallel
        eng = self._ensure_engine()
        rank, world = parallel.attach_if_distributed(eng)
        if rank != 0:
            verbose = 0
        if steps_per_epoch is None:
            steps_per_epoch = len(generator)
        history = K.History()
        cbs = list(callbacks or []) + [history]  # Keras appends History last: it sees what the other callbacks logged
        cbs = [cb for cb in cbs if rank == 0 or not getattr(cb, "rank0_only", False)]
        for cb in cbs:
            cb.set_model(self)
            cb.set_params({"epochs": epochs, "steps": steps_per_epoch, "verbose": verbose})
        train = K.BatchFeeder(generator, workers, max_queue_size)
        valid = None
        if validation_data is not None and not isinstance(validation_data, tuple):
            valid = K.BatchFeeder(validation_data, workers, max_queue_size)
        self.stop_training = False
        import torch
        # Deferred batch logs (``defer_batch_logs``, default on): when NO callback looks at a batch -- neither a class-level override of
        # on_batch_begin / on_batch_end nor a hook assigned on the instance (LambdaCallback style) -- the loop does not read every step's
        # (loss, acc) back; it collects up to 256 steps' device values and reads them in one go.  The host then runs up to 256 steps ahead
        # of the GPU, so a ``stop_training`` set from another hook is honoured at the next epoch boundary, as in Keras.
        def _plain(cb, hook):
            h = getattr(cb, hook, None)
            return h is None or getattr(h, "__func__", None) is getattr(K.Callback, hook)
        deferred = self.defer_batch_logs and hasattr(self, "_train_step") and \
            all(_plain(cb, "on_batch_end") and _plain(cb, "on_batch_begin") for cb in cbs)
        K.run_callbacks(cbs, "on_train_begin")
        try:
            for epoch in range(initial_epoch, epochs):
                K.run_callbacks(cbs, "on_epoch_begin", epoch)
                t0 = time.time()
                tot, wl, wa = 0, 0.0, 0.0
                pending = []   # (loss_acc on the device, batch size) of steps whose numbers nobody has asked for yet
                for step in range(steps_per_epoch):
                    x, y = train.get()[:2]
                    n = self._batch_size(x)
                    tot += n
                    if deferred:
                        # no callback looks at a batch's loss: the host keeps enqueueing (it needs ~1.2 ms per step) instead of waiting
                        # for every step's two numbers -- at the scripts' 64 pairs that wait is a quarter of the epoch
                        pending.append((self._train_step(x, y)["loss_acc"].clone(), n))
                        if len(pending) == 256 or step + 1 == steps_per_epoch:
                            vals = torch.stack([t for t, _ in pending]).cpu().numpy()
                            for (l_, a_), (_, n_) in zip(vals[:, :2], pending):
                                wl += float(l_) * n_
                                wa += float(a_) * n_
                            pending = []
                        continue
                    loss, acc = self.train_on_batch(x, y)
                    wl += loss * n
                    wa += acc * n
                    K.run_callbacks(cbs, "on_batch_end", step, {"loss": loss, "acc": acc, "size": n})
                logs = parallel.weighted_mean_logs({"loss": wl, "acc": wa}, tot)
                logs = {"loss": logs["loss"], "acc": logs["acc"]}
                if validation_data is not None:
                    if isinstance(validation_data, tuple):
                        vl, va = self.test_on_batch(validation_data[0], validation_data[1])
                        # data parallel: every rank holds its own tuple -- the sample-weighted mean over ranks, so that callbacks
                        # acting on val_loss / val_acc (ReduceLROnPlateau, early stopping) decide identically everywhere
                        nv = self._batch_size(validation_data[0])
                        r_ = parallel.weighted_mean_logs({"val_loss": vl * nv, "val_acc": va * nv}, nv)
                        vl, va = r_["val_loss"], r_["val_acc"]
                    else:
                        vl, va = self.evaluate_generator(valid, validation_steps)
                    logs["val_loss"], logs["val_acc"] = vl, va
                if verbose:
                    print("Epoch %d/%d - %.0fs - %s" % (epoch + 1, epochs, time.time() - t0,
                                                       " - ".join("%s: %.4f" % kv for kv in logs.items())))
                K.run_callbacks(cbs, "on_epoch_end", epoch, logs)
                # (f16 storage: the dynamic loss scale is driven by the engine's optimizer_step itself, every 16 steps)
                if self.stop_training:
                    break
        finally:
            train.close()
            if valid is not None:
                valid.close()
        K.run_callbacks(cbs, "on_train_end")
        self.history = history
        return history


# =========================================================================================================
class ConvolutionalEncoder(_TrainableModel):
    """The Sequential returned by ``get_baseline_convolutional_encoder`` (voicemap/models.py:6-41)."""

    def __init__(self, filters, embedding_dimension, input_shape=None, dropout=0.05, dtype="f16", first_pool=4):
        super().__init__()
        self.filters, self.embedding_dimension = int(filters), int(embedding_dimension)
        self.input_shape = tuple(input_shape) if input_shape is not None else None
        self.dropout = float(dropout)
        self.dtype = dtype
        f = self.filters
        self.blocks = [(32, f, int(first_pool)), (3, 2 * f, 2), (3, 3 * f, 2), (3, 4 * f, 2)]
        self.name = "sequential_1"
        self.layers: List[K.Layer] = []
        for i, (k, c, p) in enumerate(self.blocks):
            self.layers += [K.Conv1D(f"conv1d_{i+1}", filters=c, kernel_size=k, padding="same", activation="relu"),
                            K.BatchNormalization(f"batch_normalization_{i+1}", epsilon=1e-3, momentum=0.99),
                            K.SpatialDropout1D(f"spatial_dropout1d_{i+1}", rate=self.dropout),
                            K.MaxPool1D(f"max_pooling1d_{i+1}", pool_size=p, strides=p)]
        self.layers += [K.GlobalMaxPool1D("global_max_pooling1d_1"), Dense(self.embedding_dimension, name="dense_1")]
        self.classifier_units = 0  # > 0 after .add(Dense(num_classes, activation='softmax'))

    def weight_names(self) -> List[str]:
        """The encoder's own weights: once it shares a siamese engine (SiameseNet._ensure_engine) that engine also holds the
        siamese Dense(1) head, which is not a layer of this Sequential."""
        names = super().weight_names()
        if not self.classifier_units:
            names = [n for n in names if not n.startswith("head.")]
        return names

    # ---- Sequential surface ------------------------------------------------------------------------------
    def add(self, layer):
        """Only what the reference does with it: append ``Dense(num_classes, activation='softmax')``
        (experiments/train_classifier.py:112)."""
        if not isinstance(layer, Dense) or layer.activation != "softmax":
            raise NotImplementedError("only Dense(num_classes, activation='softmax') can be added to the encoder")
        if self.classifier_units:
            raise NotImplementedError("a classification layer is already present")
        if self.engine is not None:
            raise RuntimeError("add the classification layer before the model is used")
        self.classifier_units = layer.units
        layer.name = "dense_2"
        self.layers.append(layer)

    def pop(self):
        """Remove the last layer (voicemap/utils.py:145 pops the softmax layer to get the bottleneck encoder)."""
        if not self.classifier_units:
            raise NotImplementedError("only the added classification layer can be popped")
        weights = None
        if self.engine is not None:
            weights = {k: v for k, v in self.engine.get_params().items() if not k.startswith("head.")}
        self.layers.pop()
        self.classifier_units = 0
        self.engine = None
        self._pending_weights = weights

    def clone(self):
        c = ConvolutionalEncoder(self.filters, self.embedding_dimension, self.input_shape, self.dropout, self.dtype,
                                 self.blocks[0][2])
        if self.classifier_units:
            c.add(Dense(self.classifier_units, activation="softmax"))
        return c

    def _engine_args(self):
        return dict(blocks=self.blocks, embedding_dimension=self.embedding_dimension, dropout=self.dropout,
                    head="classifier" if self.classifier_units else None, num_classes=self.classifier_units, dtype=self.dtype)

    def _make_engine(self, head="__own__"):
        """The engine of this encoder on its own (``head='__own__'``: classifier head if one was added) or as the shared encoder of
        a siamese model (``head`` = its distance metric)."""
        from .engine import HipEncoderEngine
        args = self._engine_args()
        if head != "__own__":
            args.update(head=head, num_classes=0)
        return HipEncoderEngine(**args)

    def get_config(self):
        return {"class_name": "ConvolutionalEncoder", "filters": self.filters, "embedding_dimension": self.embedding_dimension,
                "input_shape": self.input_shape, "dropout": self.dropout, "dtype": self.dtype, "first_pool": self.blocks[0][2],
                "classifier_units": self.classifier_units}

    def output_shape_for(self, length: int):
        return (None, self.classifier_units or self.embedding_dimension)

    def summary(self, print_fn=print):
        length = self.input_shape[0] if self.input_shape else None
        rows, cin, total = [], 1, 0
        for i, (k, c, p) in enumerate(self.blocks):
            n_conv, n_bn = k * cin * c + c, 4 * c
            rows.append((f"conv1d_{i+1} (Conv1D)", (None, length, c), n_conv))
            rows.append((f"batch_normalization_{i+1} (BatchNormalization)", (None, length, c), n_bn))
            rows.append((f"spatial_dropout1d_{i+1} (SpatialDropout1D)", (None, length, c), 0))
            length = None if length is None else length // p
            rows.append((f"max_pooling1d_{i+1} (MaxPooling1D)", (None, length, c), 0))
            total += n_conv + n_bn
            cin = c
        rows.append(("global_max_pooling1d_1 (GlobalMaxPooling1D)", (None, cin), 0))
        n = cin * self.embedding_dimension + self.embedding_dimension
        rows.append(("dense_1 (Dense)", (None, self.embedding_dimension), n))
        total += n
        if self.classifier_units:
            n = self.embedding_dimension * self.classifier_units + self.classifier_units
            rows.append(("dense_2 (Dense)", (None, self.classifier_units), n))
            total += n
        non_trainable = sum(2 * c for (_, c, _) in self.blocks)
        print_fn("_" * 80)
        print_fn("%-48s%-22s%s" % ("Layer (type)", "Output Shape", "Param #"))
        print_fn("=" * 80)
        for name, shape, n in rows:
            print_fn("%-48s%-22s%d" % (name, str(shape), n))
        print_fn("=" * 80)
        print_fn("Total params: {:,}".format(total))
        print_fn("Trainable params: {:,}".format(total - non_trainable))
        print_fn("Non-trainable params: {:,}".format(non_trainable))
        print_fn("_" * 80)

    # ---- inference / training ----------------------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0):
        """(n, L, 1) windows -> (n, E) embeddings, or (n, num_classes) probabilities for the classifier."""
        eng = self._ensure_engine()
        if _is_lazy(x):
            emb = eng.embed(x.raw, preprocessed=False, downsampling=x.downsampling, whitening=x.whitening)
        else:
            emb = eng.embed(np.asarray(x, dtype=np.float32))
        if self.classifier_units:
            pl = eng.plan(emb.shape[0], eng_last_l0(eng), False)
            return eng.classifier_head(pl, None).cpu().numpy().copy()
        return emb.cpu().numpy().copy()

    def _labels(self, y):
        y = np.asarray(y)
        if y.ndim == 2 and y.shape[1] > 1:  # one-hot (label_preprocessor -> to_categorical, train_classifier.py:93-98)
            return y.argmax(axis=1).astype(np.int32)
        return y.reshape(-1).astype(np.int32)

    def _train_step(self, x, y):
        """One optimizer step enqueued; returns the engine's plan (``loss_acc`` on the device: reading it is the only host sync)."""
        if not self.classifier_units:
            raise RuntimeError("the bare encoder has no loss; add Dense(num_classes, activation='softmax') or wrap it in "
                               "build_siamese_net")
        if self.loss not in ("categorical_crossentropy", None):
            raise NotImplementedError("classifier loss %r" % (self.loss,))
        eng = self._ensure_engine()
        if _is_lazy(x):
            return eng.classifier_train_step(x.raw, self._labels(y), preprocessed=False, downsampling=x.downsampling,
                                             whitening=x.whitening)
        return eng.classifier_train_step(np.asarray(x, dtype=np.float32), self._labels(y))

    def train_on_batch(self, x, y):
        la = self._train_step(x, y)["loss_acc"].cpu().numpy()
        return float(la[0]), float(la[1])

    def test_on_batch(self, x, y):
        eng = self._ensure_engine()
        import torch
        if _is_lazy(x):
            emb = eng.embed(x.raw, preprocessed=False, downsampling=x.downsampling, whitening=x.whitening)
        else:
            emb = eng.embed(np.asarray(x, dtype=np.float32))
        pl = eng.plan(emb.shape[0], eng_last_l0(eng), False)
        lab = torch.as_tensor(self._labels(y)).to(eng.device, torch.int32)
        eng.classifier_head_eval(pl, lab)
        la = pl["loss_acc"].cpu().numpy()
        return float(la[0]), float(la[1])


class SpectrogramEncoder(ConvolutionalEncoder):
    """The log-mel + 2-D CNN variant of the encoder (BASELINE.json config 4; not in the reference -- DESIGN.md section 9): the same
    Sequential surface, 4 x [Conv2D 3x3 -> BatchNorm -> SpatialDropout2D -> MaxPool2D] -> GlobalMaxPool2D -> Dense over the log-mel
    image that ``vm_stft_logmel`` computes from the RAW 16 kHz window (so the batch pre-processor must not decimate or whiten:
    ``preprocess_instances(1, whitening=False)``)."""

    def __init__(self, filters, embedding_dimension, input_shape=None, dropout=0.05, dtype="f16", n_mels=64):
        super().__init__(filters, embedding_dimension, input_shape, dropout, dtype)
        from . import spectro
        self.n_mels = int(n_mels)
        f = self.filters
        self.blocks = [(3, f, 2), (3, 2 * f, 2), (3, 3 * f, 2), (3, 4 * f, 2)]
        self.name = "sequential_1"
        self.layers = [K.Lambda("log_mel_spectrogram", function="log(mel(|stft|^2) + %g): win %d hop %d n_fft %d mels %d"
                                % (spectro.LOG_FLOOR, spectro.WIN_LENGTH, spectro.HOP, spectro.N_FFT, self.n_mels))]
        for i, (k, c, p) in enumerate(self.blocks):
            self.layers += [K.Layer(f"conv2d_{i+1}", filters=c, kernel_size=(3, 3), padding="same", activation="relu"),
                            K.BatchNormalization(f"batch_normalization_{i+1}", epsilon=1e-3, momentum=0.99),
                            K.Layer(f"spatial_dropout2d_{i+1}", rate=self.dropout),
                            K.Layer(f"max_pooling2d_{i+1}", pool_size=(2, 2), strides=(2, 2))]
        self.layers += [K.Layer("global_max_pooling2d_1"), Dense(self.embedding_dimension, name="dense_1")]

    def add(self, layer):
        raise NotImplementedError("the spectrogram variant is an embedding encoder only (no classification layer)")

    def clone(self):
        return SpectrogramEncoder(self.filters, self.embedding_dimension, self.input_shape, self.dropout, self.dtype, self.n_mels)

    def _make_engine(self, head="__own__"):
        from .spectro_engine import HipSpectrogramEncoderEngine
        return HipSpectrogramEncoderEngine(self.filters, self.embedding_dimension, dropout=self.dropout,
                                           head=None if head == "__own__" else head, dtype=self.dtype, n_mels=self.n_mels)

    def get_config(self):
        return {"class_name": "SpectrogramEncoder", "filters": self.filters, "embedding_dimension": self.embedding_dimension,
                "input_shape": self.input_shape, "dropout": self.dropout, "dtype": self.dtype, "n_mels": self.n_mels,
                "first_pool": 2, "classifier_units": 0}

    def summary(self, print_fn=print):
        print_fn("log-mel front-end (25 ms / 10 ms frames, %d mel bands) + 2-D CNN encoder: Conv2D 3x3 channels %s, embedding %d, %d parameters"
                 % (self.n_mels, [b[1] for b in self.blocks], self.embedding_dimension, self.count_params()))

    def _keras_state(self):
        raise NotImplementedError("Keras HDF5 checkpoints describe the reference's 1-D encoder; save the spectrogram variant as .npz")


def get_spectrogram_convolutional_encoder(filters, embedding_dimension, input_shape=None, dropout=0.05, dtype="f16", n_mels=64):
    """The build function of the log-mel / 2-D CNN variant, same signature as ``get_baseline_convolutional_encoder``
    (voicemap/models.py:6); ``input_shape`` = (samples, 1) of the RAW window."""
    return SpectrogramEncoder(filters, embedding_dimension, input_shape, dropout, dtype=dtype, n_mels=n_mels)


def eng_last_l0(eng) -> int:
    """length of the most recent inference plan (set by HipEncoderEngine.embed)."""
    return eng.last_infer_l0


# =========================================================================================================
class SiameseNet(_TrainableModel):
    """The Model returned by ``build_siamese_net`` (voicemap/models.py:44-81): two inputs through ONE shared encoder,
    a distance layer and Dense(1, sigmoid).  ``layers[2]`` is the encoder (voicemap/utils.py:141 relies on it)."""

    def __init__(self, encoder: ConvolutionalEncoder, input_shape, distance_metric: str):
        super().__init__()
        self.encoder = encoder
        self.input_shape = tuple(input_shape)
        self.distance_metric = distance_metric
        self.name = "model_1"
        if distance_metric == "weighted_l1":
            mid = [K.Subtract("subtract_1"), K.Lambda("lambda_1", function="abs")]
        else:
            mid = [K.Subtract("subtract_embeddings"), K.Lambda("euclidean_distance", function="sqrt(sum(square(x)))")]
        self.layers = [K.InputLayer("input_1", shape=self.input_shape), K.InputLayer("input_2", shape=self.input_shape), encoder,
                       mid[0], mid[1], Dense(1, activation="sigmoid", name="dense_2")]

    def _engine_args(self):
        e = self.encoder
        return dict(blocks=e.blocks, embedding_dimension=e.embedding_dimension, dropout=e.dropout, head=self.distance_metric,
                    dtype=e.dtype)

    def _make_engine(self):
        return self.encoder._make_engine(head=self.distance_metric)

    def _ensure_engine(self):
        if self.engine is None:
            # Keras shares the encoder's variables with the siamese model (voicemap/models.py:52-53): weights the encoder
            # already holds -- set_weights / load_weights before wrapping, a trained classifier after .pop()
            # (voicemap/utils.py:143-145) -- are what the siamese model starts from; weights loaded into the siamese model
            # itself (load_model) take precedence
            enc = self.encoder
            inherited = None
            if enc.engine is not None:
                inherited = enc.engine.get_params()
            elif enc._pending_weights is not None:
                inherited = enc._pending_weights
                enc._pending_weights = None
            if inherited is not None:
                merged = OrderedDict((k, v) for k, v in inherited.items() if not k.startswith("head."))
                merged.update(self._pending_weights or {})
                self._pending_weights = merged
        eng = super()._ensure_engine()
        # the encoder object shares the siamese engine: encoder.predict() embeds with the trained weights
        self.encoder.engine = eng
        return eng

    def get_config(self):
        return {"class_name": "SiameseNet", "encoder": self.encoder.get_config(), "input_shape": self.input_shape,
                "distance_metric": self.distance_metric}

    def summary(self, print_fn=print):
        self.encoder.summary(print_fn)
        e = self.encoder.embedding_dimension
        print_fn("siamese head: %s -> Dense(1, sigmoid), %d parameters" % (self.distance_metric,
                                                                          (1 if self.distance_metric == "uniform_euclidean" else e) + 1))

    _LOSSES = {"binary_crossentropy": "bce", "contrastive_loss": "contrastive"}

    def _loss_name(self):
        l = self.loss
        if callable(l):
            l = getattr(l, "__name__", None)
        if l not in self._LOSSES:
            raise NotImplementedError("siamese loss %r (the reference uses 'binary_crossentropy' and contrastive_loss)" % (self.loss,))
        return self._LOSSES[l]

    @staticmethod
    def _pair(x):
        if not isinstance(x, (list, tuple)) or len(x) != 2:
            raise ValueError("the siamese model takes [input_1, input_2]")
        return x[0], x[1]

    def _run(self, fn, x1, x2, **kw):
        if _is_lazy(x1) != _is_lazy(x2):
            x1 = np.asarray(x1)
            x2 = np.asarray(x2)
        if _is_lazy(x1):
            assert (x1.downsampling, x1.whitening) == (x2.downsampling, x2.whitening)
            r1, r2 = x1.raw, x2.raw
            if type(r1).__name__ == "DeviceWindows":  # windows that only exist as offsets into a device buffer (shards.py)
                r1, r2 = r1.gather(), r2.gather()
            return fn(r1, r2, preprocessed=False, downsampling=x1.downsampling, whitening=x1.whitening, **kw)
        return fn(np.asarray(x1, dtype=np.float32), np.asarray(x2, dtype=np.float32), **kw)

    def _train_step(self, x, y):
        """One optimizer step enqueued; returns the engine's plan (``loss_acc`` on the device: reading it is the only host sync)."""
        eng = self._ensure_engine()
        x1, x2 = self._pair(x)
        loss = self._loss_name()
        if _is_lazy(x1) and _is_lazy(x2) and type(x1.raw).__name__ == type(x2.raw).__name__ == "DeviceWindows" \
                and x1.raw.audio is x2.raw.audio:
            # device data path: the crop happens inside the preprocessing kernel (vm_crop_decimate_whiten)
            assert (x1.downsampling, x1.whitening) == (x2.downsampling, x2.whitening)
            return eng.siamese_train_step_from_offsets(x1.raw.audio, x1.raw.offsets_host, x2.raw.offsets_host,
                                                       np.asarray(y, dtype=np.float32), x1.raw.length, loss=loss,
                                                       downsampling=x1.downsampling, whitening=x1.whitening)
        return self._run(lambda a, b, **kw: eng.siamese_train_step(a, b, np.asarray(y, dtype=np.float32), loss=loss, **kw), x1, x2)

    def train_on_batch(self, x, y):
        la = self._train_step(x, y)["loss_acc"].cpu().numpy()
        return float(la[0]), float(la[1])

    def test_on_batch(self, x, y):
        eng = self._ensure_engine()
        x1, x2 = self._pair(x)
        loss = self._loss_name()
        pl = self._run(lambda a, b, **kw: eng.siamese_eval(a, b, np.asarray(y, dtype=np.float32), loss=loss, **kw), x1, x2)
        la = pl["loss_acc"].cpu().numpy()
        return float(la[0]), float(la[1])

    def predict(self, x, batch_size=None, verbose=0):
        """siamese.predict([input_1, input_2]) -> (pairs, 1) probabilities; lower = more alike."""
        eng = self._ensure_engine()
        x1, x2 = self._pair(x)
        return self._run(lambda a, b, **kw: eng.siamese_predict(a, b, **kw), x1, x2).cpu().numpy().copy()


# =========================================================================================================
# the reference's two build functions
# =========================================================================================================
def get_baseline_convolutional_encoder(filters, embedding_dimension, input_shape=None, dropout=0.05, dtype="f16",
                                       first_pool=4):
    """voicemap/models.py:6-41.  ``input_shape`` only matters for ``summary()``: the siamese wrapper supplies it
    (models.py:10-16).  ``first_pool=2`` reproduces the geometry of the checkpoint the reference ships."""
    return ConvolutionalEncoder(filters, embedding_dimension, input_shape, dropout, dtype=dtype, first_pool=first_pool)


SIAMESE_METRICS = ("uniform_euclidean", "weighted_euclidean", "uniform_l1", "weighted_l1", "dot_product", "cosine_distance")


def build_siamese_net(encoder, input_shape, distance_metric="uniform_euclidean"):
    """voicemap/models.py:44-81: AssertionError for names outside the six allowed ones (:45-47), NotImplementedError for
    the four the reference does not implement (:70-77)."""
    assert distance_metric in SIAMESE_METRICS
    if distance_metric not in ("weighted_l1", "uniform_euclidean"):
        raise NotImplementedError
    if not isinstance(encoder, ConvolutionalEncoder) or encoder.classifier_units:
        raise ValueError("encoder must come from get_baseline_convolutional_encoder (without a classification layer)")
    return SiameseNet(encoder, input_shape, distance_metric)


def clone_model(model):
    """keras.models.clone_model for the encoder/classifier (voicemap/utils.py:143): same architecture, fresh weights."""
    return model.clone()


def load_model(filepath: str, custom_objects=None, dtype=None):
    """Load a model written by ``model.save`` (experiments/k_way_accuracy.py:45-46 uses keras.models.load_model): a Keras
    2.2.2 HDF5 checkpoint -- the reference's own files included -- or this package's .npz container."""
    from . import keras_hdf5 as KH
    if KH.is_hdf5(filepath):
        return _load_keras_hdf5(filepath, dtype)
    blob = np.load(filepath, allow_pickle=False)
    cfg = json.loads(bytes(blob["config"]).decode())

    def enc_from(c):
        if c.get("class_name") == "SpectrogramEncoder":
            return SpectrogramEncoder(c["filters"], c["embedding_dimension"], c["input_shape"], c["dropout"], c["dtype"], c["n_mels"])
        e = ConvolutionalEncoder(c["filters"], c["embedding_dimension"], c["input_shape"], c["dropout"], c["dtype"],
                                 c["first_pool"])
        if c["classifier_units"]:
            e.add(Dense(c["classifier_units"], activation="softmax"))
        return e

    if cfg["class_name"] == "SiameseNet":
        m = build_siamese_net(enc_from(cfg["encoder"]), cfg["input_shape"], cfg["distance_metric"])
    else:
        m = enc_from(cfg)
    tr = cfg.get("training") or {}
    m.compile(loss=tr.get("loss"), optimizer=Adam(**tr["optimizer"]) if tr.get("optimizer") else Adam(), metrics=tr.get("metrics"))
    m._load_state(blob)
    return m


def _load_keras_hdf5(filepath: str, dtype=None):
    """``dtype``: activation storage mode of the loaded model; default = what the file records (files written here) or
    "f16" (files written by Keras carry no such thing)."""
    from . import keras_hdf5 as KH
    ck = KH.read_checkpoint(filepath)
    g = ck["config"]
    dtype = dtype or g.get("dtype") or "f16"
    enc = get_baseline_convolutional_encoder(g["filters"], g["embedding_dimension"], input_shape=g["input_shape"],
                                             dropout=g["dropout"], dtype=dtype, first_pool=g["first_pool"])
    if ck["kind"] == "classifier":
        enc.add(Dense(g["classifier_units"], activation="softmax"))
    m = build_siamese_net(enc, g["input_shape"], g["distance_metric"]) if ck["kind"] == "siamese" else enc
    m._pending_weights = ck["params"]
    opt, tr = ck["optimizer"], ck["training"]
    if opt is not None:
        c = opt["config"]
        adam = Adam(lr=c.get("lr", 0.001), beta_1=c.get("beta_1", 0.9), beta_2=c.get("beta_2", 0.999), epsilon=c.get("epsilon"),
                    decay=c.get("decay", 0.0), clipnorm=c.get("clipnorm"))
        loss = tr["loss"] if tr else None
        m.compile(loss=contrastive_loss_by_name(loss), optimizer=adam, metrics=(tr or {}).get("metrics"))
        if opt.get("m") is not None:
            m._pending_adam = {"iterations": opt["iterations"], "m": opt["m"], "v": opt["v"]}
    m._ensure_engine()  # like the .npz path: a loaded model is live (weights, Adam slots and counters are on the device)
    return m


def contrastive_loss_by_name(loss):
    """training_config stores a custom loss by its function name; the reference's only one is utils.contrastive_loss."""
    return "contrastive_loss" if loss == "contrastive_loss" else loss


def load_keras_checkpoint_npz(weights_npz: str, dtype="f16"):
    """Build the siamese model of the reference's shipped Keras checkpoint from its exported arrays
    (tests/golden/ckpt_cfgCK_weights.npz, produced by tests/golden/extract_reference_fixtures.py with h5py): filters and
    embedding size are read from the arrays; first pool 2 and the weighted-L1 head are that checkpoint's geometry
    (tests/golden/ckpt_cfgCK_meta.json)."""
    w = np.load(weights_npz)
    f, e = w["conv1d_1/kernel"].shape[2], w["dense_1/kernel"].shape[1]
    head = "weighted_l1" if w["dense_2/kernel"].shape[0] == e and e > 1 else "uniform_euclidean"
    enc = get_baseline_convolutional_encoder(f, e, dropout=0.05, dtype=dtype, first_pool=2)
    net = build_siamese_net(enc, (12000, 1), head)
    params = OrderedDict()
    for i in range(1, 5):
        params[f"conv{i}.kernel"] = w[f"conv1d_{i}/kernel"]
        params[f"conv{i}.bias"] = w[f"conv1d_{i}/bias"]
        for a, b in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"), ("moving_variance", "moving_variance")):
            params[f"bn{i}.{a}"] = w[f"batch_normalization_{i}/{b}"]
    params["dense.kernel"], params["dense.bias"] = w["dense_1/kernel"], w["dense_1/bias"]
    params["head.kernel"], params["head.bias"] = w["dense_2/kernel"], w["dense_2/bias"]
    net._pending_weights = params
    net.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.0), metrics=["accuracy"])
    return net
