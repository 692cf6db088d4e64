"""CPU oracle for the voicemap siamese hot path -- TEST INFRASTRUCTURE ONLY.

This module is a *restatement* (PyTorch CPU, float64 or float32) of the arithmetic of
oscarknagg/voicemap's 1-D CNN siamese encoder + loss step.  It is the checker that the HIP path is
compared against.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``voicemap_amd/`` does.

Why a restatement and not the reference itself: the reference is Python-2-only code on
Keras 2.2.2 / tensorflow-gpu 1.10.1 (requirements.txt:30,69); neither runs in this toolchain
(SURVEY.md section 8c).  The arithmetic below therefore follows

* the reference's own files (cited per function as ``file:line`` relative to the reference root), and
* the published semantics of the pinned third-party layers it delegates to, marked **[3P]**:
  Keras==2.2.2 (layers, losses, optimizers) and tensorflow==1.10.1 (SAME padding, moments, pooling).

PARITY PIN STATUS: the reference's tests (tests/tests.py) pin only ``whiten`` and the sampling
invariants of the pair/task API; for the encoder / heads / losses / gradients / optimizer the
reference holds NO golden vectors and it cannot be executed here, so no output of the reference on
a given input exists to compare with.  The oracle is pinned instead to the numbers the reference's
own forward and backward pass left in the tree -- the shipped Keras checkpoint
(tests/golden/extract_reference_fixtures.py), tests/test_oracle_reference_pin.py:
  * the 640 BatchNormalization moving means / variances  <- the oracle's training-mode batch statistics
    on the 8 LibriSpeech clips the notebooks embed (per-layer correlation 0.92-0.99, median ratio
    0.92-1.04; pool-4 geometry, ReLU after BN, no whitening, no decimation, no bias, inference-mode BN
    each break it);
  * the 20 Adam second moments: their total 0.9985 = the global-norm clip at 1.0; the oracle's clipped BCE
    gradient matches mean(v) per tensor within 2.3x across 4.4 decades (with SpatialDropout1D masks per
    tower); dense_1/bias and batch_normalization_4/beta are the structural zeros the head implies;
  * dense_1/bias' drift over 11 000 iterations = Adam's epsilon 1e-7 outside the root, lr 1e-3;
plus the known-answer 5-way task of notebooks/Human_Evaluation.ipynb cell 8 (a HUMAN quiz's ground
truth: a plausibility check, not a reference output), and float64 finite differences of every
hand-derived formula.  These are statistical pins (the clips are not the training batches): they fix
the rules and scales, not the last digits -- "pinned to reference-computed statistics", not to
reference outputs.  NOT discriminated by any of it (rests on the source text / TF's documented
rule): the 15/16 vs 16/15 SAME split, the batch-global vs per-sample whitening scale, the
n/(n-(1+eps)) factor of the moving variance, zero_debias of the moving averages.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# architecture description
# ---------------------------------------------------------------------------------------------


@dataclass
class EncoderArch:
    """Geometry of get_baseline_convolutional_encoder (voicemap/models.py:6-41).

    blocks: (kernel_size, out_channels, pool) per conv block.  The current reference source uses
    kernel sizes 32,3,3,3, channels F,2F,3F,4F and pools 4,2,2,2 (models.py:13-35; MaxPool1D()
    defaults to pool 2 [3P]).  The shipped checkpoint (cfg-CK) was produced by an older revision with
    first pool 2 (tests/golden/ckpt_cfgCK_meta.json).
    """

    blocks: List[Tuple[int, int, int]]
    embedding_dimension: int
    dropout: float = 0.05
    bn_eps: float = 1e-3          # Keras BatchNormalization default epsilon [3P]
    bn_momentum: float = 0.99     # Keras BatchNormalization default momentum [3P]

    @staticmethod
    def baseline(filters: int, embedding_dimension: int, dropout: float = 0.05, first_pool: int = 4) -> "EncoderArch":
        f = filters
        return EncoderArch(blocks=[(32, f, first_pool), (3, 2 * f, 2), (3, 3 * f, 2), (3, 4 * f, 2)],
                           embedding_dimension=embedding_dimension, dropout=dropout)

    def in_channels(self, i: int) -> int:
        return 1 if i == 0 else self.blocks[i - 1][1]

    def lengths(self, l0: int) -> List[int]:
        out = [l0]
        for (_, _, p) in self.blocks:
            out.append(out[-1] // p)
        return out


def param_names(arch: EncoderArch, head: Optional[str] = "uniform_euclidean", num_classes: int = 0) -> List[str]:
    """Trainable tensors in Keras ``model.trainable_weights`` order [3P] (conv kernel, conv bias,
    BN gamma, BN beta per block; dense kernel/bias; head kernel/bias)."""
    names = []
    for i in range(len(arch.blocks)):
        names += [f"conv{i+1}.kernel", f"conv{i+1}.bias", f"bn{i+1}.gamma", f"bn{i+1}.beta"]
    names += ["dense.kernel", "dense.bias"]
    if head is not None:
        names += ["head.kernel", "head.bias"]
    return names


def init_params(arch: EncoderArch, head: Optional[str] = "uniform_euclidean", num_classes: int = 0,
                seed: int = 1234, dtype=torch.float64) -> "OrderedDict[str, torch.Tensor]":
    """Keras default initialisers [3P]: glorot_uniform kernels (limit sqrt(6/(fan_in+fan_out)),
    conv fan = K*C), zero biases, BN gamma 1 / beta 0 / moving mean 0 / moving variance 1 -- all
    confirmed by the checkpoint's model_config (tests/golden/ckpt_cfgCK_meta.json)."""
    g = torch.Generator().manual_seed(seed)
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def glorot(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype)

    for i, (k, c, _) in enumerate(arch.blocks):
        cin = arch.in_channels(i)
        p[f"conv{i+1}.kernel"] = glorot((k, cin, c), k * cin, k * c)
        p[f"conv{i+1}.bias"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.gamma"] = torch.ones(c, dtype=dtype)
        p[f"bn{i+1}.beta"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.moving_mean"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.moving_variance"] = torch.ones(c, dtype=dtype)
    cl = arch.blocks[-1][1]
    e = arch.embedding_dimension
    p["dense.kernel"] = glorot((cl, e), cl, e)
    p["dense.bias"] = torch.zeros(e, dtype=dtype)
    if head == "uniform_euclidean":
        p["head.kernel"] = glorot((1, 1), 1, 1)
        p["head.bias"] = torch.zeros(1, dtype=dtype)
    elif head == "weighted_l1":
        p["head.kernel"] = glorot((e, 1), e, 1)
        p["head.bias"] = torch.zeros(1, dtype=dtype)
    elif head == "classifier":
        p["head.kernel"] = glorot((e, num_classes), e, num_classes)
        p["head.bias"] = torch.zeros(num_classes, dtype=dtype)
    elif head is not None:
        raise ValueError(head)
    return p


def params_from_checkpoint(npz: Dict[str, np.ndarray], dtype=torch.float64):
    """Map the cfg-CK checkpoint arrays (tests/golden/ckpt_cfgCK_weights.npz) onto oracle names."""
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i in range(1, 5):
        p[f"conv{i}.kernel"] = torch.tensor(npz[f"conv1d_{i}/kernel"], dtype=dtype)
        p[f"conv{i}.bias"] = torch.tensor(npz[f"conv1d_{i}/bias"], dtype=dtype)
        p[f"bn{i}.gamma"] = torch.tensor(npz[f"batch_normalization_{i}/gamma"], dtype=dtype)
        p[f"bn{i}.beta"] = torch.tensor(npz[f"batch_normalization_{i}/beta"], dtype=dtype)
        p[f"bn{i}.moving_mean"] = torch.tensor(npz[f"batch_normalization_{i}/moving_mean"], dtype=dtype)
        p[f"bn{i}.moving_variance"] = torch.tensor(npz[f"batch_normalization_{i}/moving_variance"], dtype=dtype)
    p["dense.kernel"] = torch.tensor(npz["dense_1/kernel"], dtype=dtype)
    p["dense.bias"] = torch.tensor(npz["dense_1/bias"], dtype=dtype)
    p["head.kernel"] = torch.tensor(npz["dense_2/kernel"], dtype=dtype)
    p["head.bias"] = torch.tensor(npz["dense_2/bias"], dtype=dtype)
    f = p["conv1.kernel"].shape[2]
    arch = EncoderArch.baseline(f, p["dense.kernel"].shape[1], dropout=0.05, first_pool=2)
    return arch, p


# ---------------------------------------------------------------------------------------------
# preprocessing  (voicemap/utils.py:22-34, 88-101)
# ---------------------------------------------------------------------------------------------


def whiten(batch: np.ndarray, rms: float = 0.038021) -> np.ndarray:
    """voicemap/utils.py:88-101, closed form.

    The reference subtracts the per-sample mean over the time axis (utils.py:94-95) and multiplies by
    ONE scalar for the whole batch, ``rms / sqrt(mean(batch**2))`` taken over the *un-centred* batch
    (utils.py:98): ``np.power(batch, 2).mean()`` has no axis argument.  The tile/transpose dance at
    :95 and :99 is a broadcast.  Raises on non-3-D input like :90-91.
    """
    if batch.ndim != 3:
        raise ValueError("Input must be a 3D array of shape (n_segments, n_timesteps, 1).")
    mean = batch.mean(axis=1, keepdims=True)
    scale = rms / np.sqrt(np.power(batch, 2).mean())
    return (batch - mean) * scale


def whiten_reference_literal(batch: np.ndarray, rms: float = 0.038021) -> np.ndarray:
    """The same function written with the reference's literal tile/transpose steps (utils.py:94-99),
    kept only so a test can show the closed form above is exact."""
    sample_wise_mean = batch.mean(axis=1)
    w = batch - np.tile(sample_wise_mean, (1, 1, batch.shape[1])).transpose((1, 2, 0))
    resc = rms / np.sqrt(np.power(batch, 2).mean())
    return w * np.tile(resc, (1, 1, batch.shape[1])).transpose((1, 2, 0))


def preprocess_instances(downsampling: int, whitening: bool = True):
    """voicemap/utils.py:22-34: ``instances[:, ::downsampling, :]`` (no anti-alias filter) then whiten."""
    def fn(instances):
        instances = instances[:, ::downsampling, :]
        if whitening:
            instances = whiten(instances)
        return instances
    return fn


# ---------------------------------------------------------------------------------------------
# layers [3P semantics], channels-last (N, L, C) like Keras
# ---------------------------------------------------------------------------------------------


def same_padding(k: int) -> Tuple[int, int]:
    """TensorFlow SAME padding for stride 1 [3P]: total K-1, left = floor((K-1)/2), right = rest
    (15/16 for K=32, 1/1 for K=3)."""
    tot = k - 1
    left = tot // 2
    return left, tot - left


def conv1d_same_relu(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Keras Conv1D(padding='same', activation='relu') [3P] as used at voicemap/models.py:13-32.
    x (N,L,Cin); kernel (K,Cin,Cout) in Keras layout; cross-correlation."""
    k = kernel.shape[0]
    pl, pr = same_padding(k)
    xt = F.pad(x.transpose(1, 2), (pl, pr))
    w = kernel.permute(2, 1, 0)  # (Cout, Cin, K)
    y = F.conv1d(xt, w, bias)
    return torch.relu(y).transpose(1, 2)


def batchnorm_train(z: torch.Tensor, gamma, beta, eps: float):
    """Keras BatchNormalization(axis=-1) in training mode on a 3-D input [3P]: tf.nn.moments over
    axes (0,1) (biased variance) then tf.nn.batch_normalization: inv = gamma*rsqrt(var+eps);
    y = z*inv + (beta - mean*inv)."""
    mean = z.mean(dim=(0, 1))
    var = z.var(dim=(0, 1), unbiased=False)
    inv = gamma * torch.rsqrt(var + eps)
    return z * inv + (beta - mean * inv), mean, var


def batchnorm_infer(z, gamma, beta, moving_mean, moving_variance, eps: float):
    inv = gamma * torch.rsqrt(moving_variance + eps)
    return z * inv + (beta - moving_mean * inv)


def moving_update(moving, batch_value, momentum: float):
    """K.moving_average_update [3P]: moving -= (moving - value) * (1 - momentum)."""
    return moving - (moving - batch_value) * (1.0 - momentum)


def apply_moving_updates(new_p, collects, n_blocks: int, eps: float, momentum: float, unbiased_moving_variance: bool = True,
                         bn_state="fresh"):
    """The BatchNorm moving-statistic updates of one training step for the encoder calls in ``collects`` (one dict per tower,
    in call order).  Keras 2.2.2: ``K.moving_average_update(x, value, momentum)`` is
    ``tf.train.assign_moving_average(x, value, momentum, zero_debias=True)`` and TF 1.10's ``_zero_debias`` keeps, PER UPDATE OP
    (= per encoder call and statistic), a zero-initialised ``biased`` accumulator and a ``local_step`` [3P]:
        biased -= (biased - value) * (1 - momentum);  local_step += 1;  x = biased / (1 - momentum ** local_step)
    ``bn_state``: dict carried from step to step ({"step": t, ("m" | "v", tower, block): biased}); "fresh" = a new one (what a
    freshly built or freshly LOADED Keras model has: the accumulators are not weights and are not in a checkpoint);
    None = the plain exponential average x -= (x - value)(1 - momentum) (TF's zero_debias=False).  With two towers the reference
    runs the two update ops of a layer in an unspecified order; here they are applied in call order, so the moving statistic
    ends as the LAST tower's de-biased average.  Returns the state."""
    if bn_state == "fresh":
        bn_state = {}
    if bn_state is not None:
        bn_state["step"] = bn_state.get("step", 0) + 1
    for tw, c in enumerate(collects):
        for i in range(n_blocks):
            var = c["bn_var"][i]
            if unbiased_moving_variance:
                var = bn_unbiased_variance(var, c["bn_count"][i], eps)
            for key, name, value in (("m", f"bn{i+1}.moving_mean", c["bn_mean"][i]), ("v", f"bn{i+1}.moving_variance", var)):
                if bn_state is None:
                    new_p[name] = moving_update(new_p[name], value, momentum)
                else:
                    b = bn_state.get((key, tw, i), torch.zeros_like(value))
                    b = b - (b - value) * (1.0 - momentum)
                    bn_state[(key, tw, i)] = b
                    new_p[name] = b / (1.0 - momentum ** bn_state["step"])
    return bn_state


def bn_unbiased_variance(var, n: int, eps: float):
    """Keras 2.2.x multiplies the batch variance by n / (n - (1 + eps)) before the moving update
    (normalization.py, 'sample variance - unbiased estimator of population variance') [3P].  Not
    pinned by any reference fixture; the product exposes it as a switch (default on)."""
    return var * (n / (n - (1.0 + eps)))


def spatial_dropout(y: torch.Tensor, mask: Optional[torch.Tensor], rate: float):
    """Keras SpatialDropout1D [3P]: Bernoulli keep-mask of shape (N,1,C) scaled by 1/(1-rate).
    The TF RNG stream cannot be reproduced, so callers inject the keep-mask (0/1)."""
    if mask is None or rate == 0.0:
        return y
    return y * (mask.to(y.dtype) / (1.0 - rate))


def maxpool1d(y: torch.Tensor, pool: int) -> torch.Tensor:
    """Keras MaxPool1D(pool, pool) VALID [3P]: L_out = floor(L/pool); gradient goes to the first
    maximum of a window (torch's CPU kernel has the same tie rule)."""
    return F.max_pool1d(y.transpose(1, 2), pool, pool).transpose(1, 2)


# ---------------------------------------------------------------------------------------------
# encoder / heads  (voicemap/models.py)
# ---------------------------------------------------------------------------------------------


# Storage emulation: what error does 16-bit storage ALONE introduce?  Used only to calibrate the tolerances of the bf16 / f16
# parity tests.  ``storage`` is None (exact), "bf16" or "f16"; with "f16" the HIP path multiplies the loss gradient by a loss scale
# (engine.loss_scale) so that the stored activation gradients stay inside half's range: _GSCALE is that factor for the backward
# roundings (set by siamese_train_step for the duration of a step).
_STORE_DT = {"bf16": torch.bfloat16, "f16": torch.float16}
_GSCALE = [1.0]


def _rnd(x, storage):
    return x.to(_STORE_DT[storage]).to(x.dtype)


def _rnd_grad(g, storage):
    s = _GSCALE[0] if storage == "f16" else 1.0
    return _rnd(g * s, storage) / s


class _Store16(torch.autograd.Function):
    """Emulates a tensor that the HIP path keeps in a 16-bit type in HBM: the value is rounded on the way forward and the
    gradient that is stored at the same point (du / dp) is rounded on the way back."""

    @staticmethod
    def forward(ctx, x, storage):
        ctx.storage = storage
        return _rnd(x, storage)

    @staticmethod
    def backward(ctx, g):
        return _rnd_grad(g, ctx.storage), None


def _store(x, storage):
    return _Store16.apply(x, storage) if storage in _STORE_DT else x


class _StoreAs16(torch.autograd.Function):
    """Forward: the given (already rounded) value; backward: like _Store16 (the gradient stored there is 16-bit)."""

    @staticmethod
    def forward(ctx, x, value, storage):
        ctx.storage = storage
        return value

    @staticmethod
    def backward(ctx, g):
        return _rnd_grad(g, ctx.storage), None, None


class _Bf16GradOnly(torch.autograd.Function):
    """A tensor that never reaches HBM (forward exact) but whose gradient is consumed as a bf16 matrix operand (block 1's du in
    the fused kernels: bf16 whatever the storage type -- bf16 has fp32's exponent range, no scale needed)."""

    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def encoder_forward(arch: EncoderArch, p, x: torch.Tensor, training: bool,
                    drop_masks: Optional[Sequence[Optional[torch.Tensor]]] = None,
                    collect: Optional[dict] = None, storage: Optional[str] = None) -> torch.Tensor:
    """get_baseline_convolutional_encoder (voicemap/models.py:6-41): 4 x [Conv1D SAME + bias -> ReLU
    -> BatchNorm -> SpatialDropout1D -> MaxPool1D] -> GlobalMaxPool1D (:37) -> Dense(E) linear (:39).
    x is (N, L0, 1).  ``collect`` (optional dict) receives per-block batch statistics and
    activations."""
    h = x
    if storage == "f16":
        # (round 6) f16 storage emulation only: the HIP block-1 convolution takes the whitened waveform rounded to half -- the precision every
        # other layer's input has in this mode -- against filters split hi + lo in halves (conv1_fused.hip, f1_products = 2)
        h = _rnd(h, storage)
    for i, (k, c, pool) in enumerate(arch.blocks):
        kern = p[f"conv{i+1}.kernel"]
        if storage in _STORE_DT and i > 0:  # GEMM operand copies of the k=3 kernels are 16-bit; block 1's filters keep 16+ bits (split)
            kern = kern + (_rnd(kern.detach(), storage) - kern.detach())
        # bf16 emulation of block 1: the fused kernels never store z1 (statistics, arg-max and the backward recompute see
        # the fp32 accumulator; only du1 becomes a bf16 MFMA operand); what IS stored is bf16(pooled extreme of z1), and the
        # BN affine + dropout run over that -- so the pooled activation is rounded twice.
        fused1 = storage in _STORE_DT and i == 0 and pool in (2, 4)
        z = conv1d_same_relu(h, kern, p[f"conv{i+1}.bias"])
        z = _Bf16GradOnly.apply(z) if fused1 else _store(z, storage)
        if training:
            y, mean, var = batchnorm_train(z, p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"], arch.bn_eps)
            if collect is not None:
                collect.setdefault("bn_mean", []).append(mean.detach())
                collect.setdefault("bn_var", []).append(var.detach())
                collect.setdefault("bn_count", []).append(z.shape[0] * z.shape[1])
            m = drop_masks[i] if drop_masks is not None else None
            y = spatial_dropout(y, m, arch.dropout)
        else:
            y = batchnorm_infer(z, p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"], p[f"bn{i+1}.moving_mean"],
                                p[f"bn{i+1}.moving_variance"], arch.bn_eps)
        if fused1:
            with torch.no_grad():
                gam, bet = p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"]
                if training:
                    inv = gam * torch.rsqrt(var + arch.bn_eps)
                    sh = bet - mean * inv
                    mult = 1.0 if (m is None or arch.dropout == 0.0) else m.to(z.dtype) / (1.0 - arch.dropout)
                else:
                    inv = gam * torch.rsqrt(p[f"bn{i+1}.moving_variance"] + arch.bn_eps)
                    sh = bet - p[f"bn{i+1}.moving_mean"] * inv
                    mult = 1.0
                ext = torch.where(inv >= 0, maxpool1d(z, pool), -maxpool1d(-z, pool))
                act = _rnd((_rnd(ext, storage) * inv + sh) * mult, storage)
            h = _StoreAs16.apply(maxpool1d(y, pool), act, storage)
        else:
            h = _store(maxpool1d(y, pool), storage)
        if collect is not None:
            collect.setdefault("z", []).append(z.detach())
            collect.setdefault("pooled", []).append(h.detach())
    g = h.max(dim=1).values  # GlobalMaxPool1D
    if collect is not None:
        collect["global_max"] = g.detach()
    return g @ p["dense.kernel"] + p["dense.bias"]


SIAMESE_METRICS = ("uniform_euclidean", "weighted_euclidean", "uniform_l1", "weighted_l1", "dot_product",
                   "cosine_distance")


def siamese_head(p, e1: torch.Tensor, e2: torch.Tensor, distance_metric: str = "uniform_euclidean") -> torch.Tensor:
    """build_siamese_net head (voicemap/models.py:55-77).  uniform_euclidean: sqrt(sum((e1-e2)^2))
    keepdims -> Dense(1, sigmoid) (:61-69); weighted_l1: |e1-e2| -> Dense(1, sigmoid) (:55-60);
    the other allowed names raise NotImplementedError (:70-77); unknown names fail the assert (:45-47).
    No epsilon under the sqrt: identical embeddings give an infinite derivative like the reference."""
    assert distance_metric in SIAMESE_METRICS
    if distance_metric == "weighted_l1":
        d = (e1 - e2).abs()
    elif distance_metric == "uniform_euclidean":
        d = torch.sqrt(((e1 - e2) ** 2).sum(dim=-1, keepdim=True))
    else:
        raise NotImplementedError
    return torch.sigmoid(d @ p["head.kernel"] + p["head.bias"])


def siamese_forward(arch, p, x1, x2, training: bool, distance_metric="uniform_euclidean",
                    drop_masks1=None, drop_masks2=None, collect1=None, collect2=None, storage=None):
    """One shared encoder called on each input (voicemap/models.py:52-53): in training mode each call
    normalises with its OWN batch statistics."""
    e1 = encoder_forward(arch, p, x1, training, drop_masks1, collect1, storage)
    e2 = encoder_forward(arch, p, x2, training, drop_masks2, collect2, storage)
    return siamese_head(p, e1, e2, distance_metric), e1, e2


def classifier_forward(arch, p, x, training: bool, drop_masks=None, collect=None):
    """experiments/train_classifier.py:110-112: encoder + Dense(num_classes, softmax)."""
    e = encoder_forward(arch, p, x, training, drop_masks, collect)
    return torch.softmax(e @ p["head.kernel"] + p["head.bias"], dim=-1), e


# ---------------------------------------------------------------------------------------------
# losses / metrics
# ---------------------------------------------------------------------------------------------

KERAS_EPSILON = 1e-7  # K.epsilon() [3P]


def contrastive_loss(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """voicemap/utils.py:77-85: mean((1-y)*p^2 + y*max(margin-p,0)^2), margin 1 (:81), mean over all
    elements.  Labels: 0 = same speaker, 1 = different (voicemap/librispeech.py:194)."""
    margin = 1.0
    return ((1 - y_true) * y_pred ** 2 + y_true * torch.clamp(margin - y_pred, min=0) ** 2).mean()


def binary_crossentropy(y_true, y_pred):
    """Keras 'binary_crossentropy' on a sigmoid output with the TF backend [3P]: clip p to
    [eps, 1-eps], logit = log(p/(1-p)), sigmoid_cross_entropy_with_logits, mean
    (experiments/train_siamese.py:57)."""
    pc = torch.clamp(y_pred, KERAS_EPSILON, 1 - KERAS_EPSILON)
    logit = torch.log(pc / (1 - pc))
    return (torch.clamp(logit, min=0) - logit * y_true + torch.log1p(torch.exp(-logit.abs()))).mean()


def categorical_crossentropy(y_true_onehot, y_pred):
    """Keras 'categorical_crossentropy' on a softmax output [3P]: renormalise, clip to [eps, 1-eps],
    -sum(t*log(p)) per sample, mean (experiments/train_classifier.py:115)."""
    q = y_pred / y_pred.sum(dim=-1, keepdim=True)
    q = torch.clamp(q, KERAS_EPSILON, 1 - KERAS_EPSILON)
    return (-(y_true_onehot * torch.log(q)).sum(dim=-1)).mean()


def binary_accuracy(y_true, y_pred):
    """Keras metrics=['accuracy'] with a 1-unit output -> binary_accuracy: mean(round(p) == y) [3P]."""
    return (torch.round(y_pred) == y_true).to(y_pred.dtype).mean()


def categorical_accuracy(y_true_onehot, y_pred):
    return (y_pred.argmax(-1) == y_true_onehot.argmax(-1)).to(y_pred.dtype).mean()


# ---------------------------------------------------------------------------------------------
# optimizer: Keras Adam(clipnorm=1.) [3P]  (experiments/train_siamese.py:56)
# ---------------------------------------------------------------------------------------------


@dataclass
class AdamState:
    lr: float = 1e-3
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = KERAS_EPSILON
    decay: float = 0.0
    clipnorm: Optional[float] = 1.0
    iterations: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def global_norm(grads: Dict[str, torch.Tensor]) -> torch.Tensor:
    return torch.sqrt(sum((g ** 2).sum() for g in grads.values()))


def adam_step(state: AdamState, params, grads: Dict[str, torch.Tensor]):
    """Standalone-Keras 2.2.2 Adam.get_updates + Optimizer.get_gradients [3P]:
    norm = sqrt(sum_g sum(g^2)) over ALL gradients; g <- g*clip/norm if norm >= clip;
    lr <- lr/(1+decay*iter) if decay; t = iter+1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1*m+(1-b1)g; v = b2*v+(1-b2)g^2; p -= lr_t*m/(sqrt(v)+eps)."""
    names = list(grads.keys())
    if state.clipnorm is not None and state.clipnorm > 0:
        n = global_norm(grads)
        if float(n) >= state.clipnorm:
            grads = {k: g * (state.clipnorm / n) for k, g in grads.items()}
    lr = state.lr
    if state.decay > 0:
        lr = lr * (1.0 / (1.0 + state.decay * state.iterations))
    t = state.iterations + 1
    lr_t = lr * math.sqrt(1.0 - state.beta_2 ** t) / (1.0 - state.beta_1 ** t)
    for k in names:
        g = grads[k]
        m = state.m.get(k, torch.zeros_like(g))
        v = state.v.get(k, torch.zeros_like(g))
        m = state.beta_1 * m + (1 - state.beta_1) * g
        v = state.beta_2 * v + (1 - state.beta_2) * g * g
        params[k] = params[k] - lr_t * m / (torch.sqrt(v) + state.epsilon)
        state.m[k], state.v[k] = m, v
    state.iterations = t
    return params


# ---------------------------------------------------------------------------------------------
# one training step of either siamese script
# ---------------------------------------------------------------------------------------------


def siamese_train_step(arch, p, state: Optional[AdamState], x1, x2, y, loss: str = "contrastive",
                       distance_metric: str = "uniform_euclidean", drop_masks1=None, drop_masks2=None,
                       unbiased_moving_variance: bool = True, storage: Optional[str] = None, bn_state="fresh",
                       loss_scale: float = 4096.0):
    """train_on_batch of experiments/siamese_contrastive_loss.py:70 (loss='contrastive') or
    experiments/train_siamese.py:57 (loss='bce'): forward both towers in training mode, loss, grads
    wrt the 20 trainable tensors, global-norm clip + Adam, two sequential BN moving-stat updates
    (tower 1 then tower 2).  Returns a dict with everything a parity test needs."""
    head = "head.kernel" in p
    names = param_names(arch)
    leaf = OrderedDict((k, (v.detach().clone().requires_grad_(k in names))) for k, v in p.items())
    c1, c2 = {}, {}
    pred, e1, e2 = siamese_forward(arch, leaf, x1, x2, True, distance_metric, drop_masks1, drop_masks2, c1, c2, storage)
    if loss == "contrastive":
        l = contrastive_loss(y, pred)
    elif loss in ("bce", "binary_crossentropy"):
        l = binary_crossentropy(y, pred)
    else:
        raise ValueError(loss)
    acc = binary_accuracy(y, pred)
    _GSCALE[0] = float(loss_scale) if storage == "f16" else 1.0   # storage emulation only: the backward roundings see scaled gradients
    try:
        gl = torch.autograd.grad(l, [leaf[k] for k in names])
    finally:
        _GSCALE[0] = 1.0
    grads = OrderedDict((k, g.detach()) for k, g in zip(names, gl))
    new_p = OrderedDict((k, v.detach().clone()) for k, v in p.items())
    bn_state = apply_moving_updates(new_p, (c1, c2), len(arch.blocks), arch.bn_eps, arch.bn_momentum, unbiased_moving_variance, bn_state)
    gnorm = global_norm(grads)
    if state is not None:
        tr = OrderedDict((k, new_p[k]) for k in names)
        tr = adam_step(state, tr, grads)
        for k in names:
            new_p[k] = tr[k]
    return {"loss": l.detach(), "acc": acc.detach(), "pred": pred.detach(), "e1": e1.detach(), "e2": e2.detach(),
            "grads": grads, "grad_norm": gnorm, "params": new_p, "collect1": c1, "collect2": c2, "bn_state": bn_state}


def classifier_train_step(arch, p, state: Optional[AdamState], x, y_onehot, drop_masks=None,
                          unbiased_moving_variance: bool = True, bn_state="fresh"):
    """train_on_batch of experiments/train_classifier.py:110-115 (categorical CE + Adam(clipnorm 1))."""
    names = param_names(arch)
    leaf = OrderedDict((k, (v.detach().clone().requires_grad_(k in names))) for k, v in p.items())
    c = {}
    prob, e = classifier_forward(arch, leaf, x, True, drop_masks, c)
    l = categorical_crossentropy(y_onehot, prob)
    acc = categorical_accuracy(y_onehot, prob)
    gl = torch.autograd.grad(l, [leaf[k] for k in names])
    grads = OrderedDict((k, g.detach()) for k, g in zip(names, gl))
    new_p = OrderedDict((k, v.detach().clone()) for k, v in p.items())
    bn_state = apply_moving_updates(new_p, (c,), len(arch.blocks), arch.bn_eps, arch.bn_momentum, unbiased_moving_variance, bn_state)
    if state is not None:
        tr = adam_step(state, OrderedDict((k, new_p[k]) for k in names), grads)
        for k in names:
            new_p[k] = tr[k]
    return {"loss": l.detach(), "acc": acc.detach(), "prob": prob.detach(), "e": e.detach(), "grads": grads,
            "grad_norm": global_norm(grads), "params": new_p, "collect": c, "bn_state": bn_state}


# ---------------------------------------------------------------------------------------------
# n-shot k-way evaluation distances  (voicemap/utils.py:104-216)
# ---------------------------------------------------------------------------------------------


def n_shot_prediction(query_embedding: np.ndarray, support_set_embeddings: np.ndarray, n: int, k: int,
                      distance: str = "euclidean") -> np.ndarray:
    """The ``pred`` vector of utils.py:159-206 for one task (float64 numpy like the reference):
    euclidean: per-class mean of support embeddings then L2 (:159-170); cosine: per-class mean of unit
    vectors then scipy cdist 'cosine' (:171-184); dot_product: mean magnitude x mean unit vector, then
    negative dot product (:185-206).  Correct iff argmin == 0 (:208-210)."""
    q = np.asarray(query_embedding, dtype=np.float64).reshape(1, -1)
    s = np.asarray(support_set_embeddings, dtype=np.float64)
    if distance == "euclidean":
        means = np.stack([s[i:i + n].mean(axis=0) for i in range(0, n * k, n)])
        return np.sqrt(np.power(np.concatenate([q] * k) - means, 2).sum(axis=1))
    mag = np.linalg.norm(s, axis=1, keepdims=True)
    unit = s / mag
    mean_unit = np.stack([unit[i:i + n].mean(axis=0) for i in range(0, n * k, n)])
    if distance == "cosine":
        qn = q / np.linalg.norm(q)
        mu = mean_unit / np.linalg.norm(mean_unit, axis=1, keepdims=True)
        return (1.0 - qn @ mu.T)[0]
    if distance == "dot_product":
        mean_mag = np.vstack([mag[i:i + n].sum() / n for i in range(0, n * k, n)])
        return (-(q @ (mean_mag * mean_unit).T))[0]
    raise ValueError("Distance must be in (euclidean, cosine, dot_product)")


# ---------------------------------------------------------------------------------------------
# synthetic workload of SURVEY section 8(d) + the timed CPU baseline leg of bench.py
# ---------------------------------------------------------------------------------------------


def synthetic_pairs(batch_pairs: int, seed: int = 1234, samples: int = 48000):
    """raw windows N(0, 0.05^2) + per-window DC offset U(-0.01, 0.01); labels zeros for
    the first half, ones for the rest (voicemap/librispeech.py:194 layout)."""
    rng = np.random.default_rng(seed)
    def one():
        x = rng.normal(0.0, 0.05, size=(batch_pairs, samples, 1))
        return (x + rng.uniform(-0.01, 0.01, size=(batch_pairs, 1, 1))).astype(np.float32)
    x1, x2 = one(), one()
    y = np.concatenate([np.zeros(batch_pairs // 2), np.ones(batch_pairs - batch_pairs // 2)])[:, None].astype(np.float32)
    return x1, x2, y



# ---------------------------------------------------------------------------------------------
# log-mel front-end + 2-D CNN encoder variant (BASELINE.json config 4).  NOT in the reference
# (SURVEY.md D9): the specification is DESIGN.md section 9 / voicemap_amd/spectro.py and this is
# its float64 restatement -- PARITY UNPINNED by construction (there is nothing in the reference
# tree to pin it on); the HIP path (vm_stft_logmel, the band-stacked Conv2D lowering) is checked
# against this code, and this code against numpy.fft on the same frames.
# ---------------------------------------------------------------------------------------------


def logmel_features(raw: np.ndarray, win_length: int = 400, hop: int = 160, n_fft: int = 512, n_mels: int = 64,
                    sample_rate: int = 16000, log_floor: float = 1e-6) -> np.ndarray:
    """raw (B, n) -> (B, T, n_mels) float64: frames of win_length samples every hop samples (no centre padding), periodic
    Hann window, n_fft-point DFT (frame zero-extended at the END), power of bins 0..n_fft/2-1, triangular HTK-mel filters with
    unit peak between 0 Hz and Nyquist, natural log(mel + log_floor).  Written with numpy.fft, independently of the DFT-basis
    GEMM the HIP kernel uses."""
    raw = np.asarray(raw, dtype=np.float64)
    if raw.ndim == 3:
        raw = raw[:, :, 0]
    B, n = raw.shape
    T = 1 + (n - win_length) // hop
    idx = np.arange(T)[:, None] * hop + np.arange(win_length)[None, :]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    frames = raw[:, idx] * win[None, None, :]
    spec = np.fft.rfft(frames, n=n_fft, axis=-1)
    power = (spec.real ** 2 + spec.imag ** 2)[..., :n_fft // 2]
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    edges = 700.0 * (10.0 ** (np.linspace(mel(0.0), mel(sample_rate / 2.0), n_mels + 2) / 2595.0) - 1.0)
    f = (np.arange(n_fft // 2) * sample_rate / n_fft)[:, None]
    lo, mid, hi = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    w = np.maximum(0.0, np.minimum((f - lo) / (mid - lo), (hi - f) / (hi - mid)))
    return np.log(power @ w + log_floor)


@dataclass
class Encoder2dArch:
    """The 2-D variant of get_baseline_convolutional_encoder over a (T, M) log-mel image: 4 x [Conv2D 3x3 SAME + bias -> ReLU
    -> BatchNorm -> SpatialDropout2D -> MaxPool2D(2, 2)] -> GlobalMaxPool2D -> Dense(E); channels F, 2F, 3F, 4F like the 1-D
    encoder (voicemap/models.py:13-35)."""

    filters: int
    embedding_dimension: int
    dropout: float = 0.05
    bn_eps: float = 1e-3
    bn_momentum: float = 0.99

    @property
    def channels(self) -> List[int]:
        return [self.filters * (i + 1) for i in range(4)]


def init_params2d(arch: Encoder2dArch, head: Optional[str] = "uniform_euclidean", seed: int = 1234, dtype=torch.float64):
    """Keras defaults [3P]: glorot_uniform Conv2D kernels (fan = 9 * C), zero biases, gamma 1, beta 0, moving 0 / 1."""
    g = torch.Generator().manual_seed(seed)
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def glorot(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype)

    cin = 1
    for i, c in enumerate(arch.channels):
        p[f"conv{i+1}.kernel"] = glorot((3, 3, cin, c), 9 * cin, 9 * c)
        p[f"conv{i+1}.bias"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.gamma"] = torch.ones(c, dtype=dtype)
        p[f"bn{i+1}.beta"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.moving_mean"] = torch.zeros(c, dtype=dtype)
        p[f"bn{i+1}.moving_variance"] = torch.ones(c, dtype=dtype)
        cin = c
    e = arch.embedding_dimension
    p["dense.kernel"] = glorot((cin, e), cin, e)
    p["dense.bias"] = torch.zeros(e, dtype=dtype)
    if head == "uniform_euclidean":
        p["head.kernel"] = glorot((1, 1), 1, 1)
        p["head.bias"] = torch.zeros(1, dtype=dtype)
    elif head == "weighted_l1":
        p["head.kernel"] = glorot((e, 1), e, 1)
        p["head.bias"] = torch.zeros(1, dtype=dtype)
    elif head is not None:
        raise ValueError(head)
    return p


def encoder2d_forward(arch: Encoder2dArch, p, feats: torch.Tensor, training: bool, drop_masks=None, collect: Optional[dict] = None):
    """feats (B, T, M) log-mel -> (B, E).  Conv2D kernels are Keras-shaped (kT, kM, C_in, C_out) over a channels-last (B, T, M, C)
    image; BatchNorm statistics over (B, T, M) of THIS call (one tower); SpatialDropout2D masks are (B, C) keep masks."""
    h = feats[:, None, :, :]  # (B, C=1, T, M)
    for i, c in enumerate(arch.channels):
        w = p[f"conv{i+1}.kernel"].permute(3, 2, 0, 1)  # (C_out, C_in, kT, kM)
        z = torch.relu(F.conv2d(h, w, p[f"conv{i+1}.bias"], padding=1))
        gam, bet = p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"]
        if training:
            mean = z.mean(dim=(0, 2, 3))
            var = z.var(dim=(0, 2, 3), unbiased=False)
            y = (z - mean[None, :, None, None]) * torch.rsqrt(var + arch.bn_eps)[None, :, None, None] * gam[None, :, None, None] \
                + bet[None, :, None, None]
            if collect is not None:
                collect.setdefault("bn_mean", []).append(mean.detach())
                collect.setdefault("bn_var", []).append(var.detach())
                collect.setdefault("bn_count", []).append(z.shape[0] * z.shape[2] * z.shape[3])
            if drop_masks is not None and drop_masks[i] is not None and arch.dropout > 0.0:
                y = y * (drop_masks[i].to(y.dtype) / (1.0 - arch.dropout))[:, :, None, None]
        else:
            inv = gam * torch.rsqrt(p[f"bn{i+1}.moving_variance"] + arch.bn_eps)
            y = z * inv[None, :, None, None] + (bet - p[f"bn{i+1}.moving_mean"] * inv)[None, :, None, None]
        h = F.max_pool2d(y, 2, 2)
        if collect is not None:
            collect.setdefault("z", []).append(z.detach())
            collect.setdefault("pooled", []).append(h.detach())
    g = h.amax(dim=(2, 3))
    return g @ p["dense.kernel"] + p["dense.bias"]


def siamese2d_train_step(arch: Encoder2dArch, p, state: Optional["AdamState"], f1, f2, y, loss: str = "contrastive",
                         distance_metric: str = "uniform_euclidean", drop_masks1=None, drop_masks2=None,
                         unbiased_moving_variance: bool = True, bn_state="fresh"):
    """siamese_train_step for the 2-D variant on log-mel features f1, f2 (B, T, M): same loss / clip / Adam / moving-statistics
    arithmetic as the 1-D step."""
    names = [k for k in p if "moving" not in k]
    q = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    c1, c2 = {}, {}
    e1 = encoder2d_forward(arch, q, f1, True, drop_masks1, c1)
    e2 = encoder2d_forward(arch, q, f2, True, drop_masks2, c2)
    pred = siamese_head(q, e1, e2, distance_metric)
    l = contrastive_loss(y, pred) if loss in ("contrastive", "contrastive_loss") else binary_crossentropy(y, pred)
    acc = binary_accuracy(y, pred)
    grads = dict(zip(names, torch.autograd.grad(l, [q[k] for k in names])))
    new_p = OrderedDict((k, v.detach().clone()) for k, v in p.items())
    bn_state = apply_moving_updates(new_p, (c1, c2), 4, arch.bn_eps, arch.bn_momentum, unbiased_moving_variance, bn_state)
    if state is not None:
        upd = adam_step(state, {k: new_p[k] for k in names}, grads)
        new_p.update(upd)
    return {"loss": l.detach(), "acc": acc.detach(), "pred": pred.detach(), "e1": e1.detach(), "e2": e2.detach(), "grads": grads,
            "params": new_p, "z1": c1["z"], "pooled1": c1["pooled"], "bn_state": bn_state}


def time_cpu_train_steps(arch, batch_pairs: int, steps: int, loss: str = "contrastive", threads: Optional[int] = None,
                         seed: int = 1234, downsampling: int = 4, budget_s: float = 25.0):
    """bench.py cpu_baseline leg: the oracle's fp32 training step (preprocess + twin forward + loss +
    backward + clip + Adam) on a bounded sample, on the host cores.  With ``threads=None`` a few intra-op
    thread counts are tried for one step each (a small batch does not scale to hundreds of cores; the best
    count is what a user of the CPU path would run) and the best is used for the timed steps.
    Returns (seconds_per_step, threads_used); ``time_cpu_train_steps.last_trials`` holds {threads: seconds of the trial step}."""
    import os
    import time
    time_cpu_train_steps.last_trials = {}
    x1, x2, y = synthetic_pairs(batch_pairs, seed)
    pre = preprocess_instances(downsampling)
    yt = torch.tensor(y)

    def run(n_steps, deadline):
        p = init_params(arch, seed=seed, dtype=torch.float32)
        st = AdamState()
        ts = []
        for i in range(n_steps + 1):
            t0 = time.perf_counter()
            a = torch.tensor(pre(x1.astype(np.float64)).astype(np.float32))
            b = torch.tensor(pre(x2.astype(np.float64)).astype(np.float32))
            out = siamese_train_step(arch, p, st, a, b, yt, loss=loss)
            p = out["params"]
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() > deadline and len(ts) >= 2:
                break
        return float(np.mean(ts[1:])) if len(ts) > 1 else float(ts[0])

    t_start = time.perf_counter()
    if threads is None:
        cores = os.cpu_count() or 1
        # a batch of 8 pairs stops scaling well below 64 threads (256 threads: 50 s per step on the GPU node's host); the counts tried
        # and their step times go into the bench line next to os.cpu_count()
        cands = sorted({max(1, min(cores, c)) for c in (8, 16, 32, 64)})
        best, best_t = None, None
        for c in cands:
            if time.perf_counter() - t_start > budget_s * 0.5 and best is not None:
                break
            torch.set_num_threads(c)
            t = run(1, time.perf_counter() + budget_s * 0.15)
            time_cpu_train_steps.last_trials[c] = t
            if best_t is None or t < best_t:
                best, best_t = c, t
        threads = best
    torch.set_num_threads(threads)
    sec = run(steps, t_start + budget_s)
    return sec, threads


def time_cpu_embed_only(arch, windows: int, reps: int, threads: int, seed: int = 1234, downsampling: int = 4, budget_s: float = 8.0):
    """bench.py cpu_baseline leg: the inference embedding pass (preprocess + encoder forward with the moving statistics,
    voicemap/utils.py:141-156 `encoder.predict`) of ``windows`` 3 s windows, fp32, ``threads`` intra-op threads.  Seconds per pass."""
    import time
    torch.set_num_threads(threads)
    x1, _, _ = synthetic_pairs(windows, seed)
    pre = preprocess_instances(downsampling)
    p = init_params(arch, seed=seed, dtype=torch.float32)
    ts, t_start = [], time.perf_counter()
    with torch.no_grad():
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            a = torch.tensor(pre(x1.astype(np.float64)).astype(np.float32))
            encoder_forward(arch, p, a, False)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and len(ts) >= 2:
                break
    return float(np.mean(ts[1:])) if len(ts) > 1 else float(ts[0])


def time_cpu_classifier_steps(arch, batch: int, num_classes: int, steps: int, threads: int, seed: int = 1234, downsampling: int = 4,
                              budget_s: float = 8.0):
    """bench.py cpu_baseline leg: BASELINE.json config 1 -- one train_on_batch of experiments/train_classifier.py:110-127 (encoder +
    Dense(num_classes, softmax), categorical cross-entropy, Adam(clipnorm 1)) at batch ``batch``, fp32.  Seconds per step."""
    import time
    torch.set_num_threads(threads)
    x, _, _ = synthetic_pairs(batch, seed)
    pre = preprocess_instances(downsampling)
    p = init_params(arch, head="classifier", num_classes=num_classes, seed=seed, dtype=torch.float32)
    st = AdamState()
    labels = np.arange(batch) % num_classes
    y = torch.tensor(np.eye(num_classes, dtype=np.float32)[labels])
    ts, t_start = [], time.perf_counter()
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        a = torch.tensor(pre(x.astype(np.float64)).astype(np.float32))
        out = classifier_train_step(arch, p, st, a, y)
        p = out["params"]
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and len(ts) >= 2:
            break
    return float(np.mean(ts[1:])) if len(ts) > 1 else float(ts[0])

