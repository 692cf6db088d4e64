"""Host-side driver of the HIP hot path: owns the flat parameter / gradient / Adam buffers and the activation
workspaces, and enqueues the libvoicemap_hip.so kernels for one encoder forward, backward and optimizer step.

PyTorch is used for device memory, streams and (in parallel.py) torch.distributed -- plumbing only; every
arithmetic op of the path is a call through the C ABI (include/voicemap_hip.h).  Mirrors, at the level of one
``train_on_batch``, what Keras does for the reference scripts (experiments/train_siamese.py:54-57,65;
experiments/siamese_contrastive_loss.py:67-70; experiments/train_classifier.py:110-115).
"""
from __future__ import annotations

import ctypes
import math
import struct
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import VM_BF16, VM_F16, VM_F32, VM_F32S

# "f32s": fp32 storage, split-bf16 products in the k=3 convolution GEMMs (VM_F32S in include/voicemap_hip.h)
# "f16": IEEE half storage -- the bf16 kernels with 11 instead of 8 significand bits per stored value, loss-scaled gradients
_DT = {"f32": VM_F32, "fp32": VM_F32, "float32": VM_F32, "bf16": VM_BF16, "bfloat16": VM_BF16, "f32s": VM_F32S,
       "f16": VM_F16, "fp16": VM_F16, "float16": VM_F16}
_TORCH_DT = {VM_F32: torch.float32, VM_BF16: torch.bfloat16, VM_F32S: torch.float32, VM_F16: torch.float16}
DEFAULT_F16_LOSS_SCALE = 4096.0
# the drop-in surface's default storage mode: the 16-bit one whose embeddings stay within 1e-3 of the reference arithmetic (DESIGN.md
# 4.6b); "bf16" -- BASELINE.json's word for config 2 -- is the same kernels with 8 significand bits (2 % faster, 6e-3)
DEFAULT_DTYPE = "f16"
HEADS = {"uniform_euclidean": _lib.VM_HEAD_UNIFORM_EUCLIDEAN, "weighted_l1": _lib.VM_HEAD_WEIGHTED_L1}
LOSSES = {"contrastive": _lib.VM_LOSS_CONTRASTIVE, "contrastive_loss": _lib.VM_LOSS_CONTRASTIVE,
          "bce": _lib.VM_LOSS_BCE, "binary_crossentropy": _lib.VM_LOSS_BCE}
CONV1_HALO_L, CONV1_HALO = 15, 31  # TF SAME padding of the k=32 first conv: 15 left / 16 right


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def flat_layout(blocks: Sequence[Tuple[int, int, int]], embedding_dimension: int, head: Optional[str], num_classes: int = 0):
    """Layout of the flat fp32 buffers (parameters / gradients / Adam slots share it): tensors in Keras ``trainable_weights``
    order, each starting on a 64-element boundary, and the non-trainable BatchNorm moving statistics in a second buffer.
    Returns (spec, offsets {name: (offset, size, shape)}, n_flat, nt_off {name: (offset, size)}, n_nt).  Pure host code:
    parallel.py's collectives and the CPU tests use the same layout the engine does."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    for i, (k, c, _) in enumerate(blocks):
        cin = 1 if i == 0 else blocks[i - 1][1]
        spec += [(f"conv{i+1}.kernel", (k, cin, c)), (f"conv{i+1}.bias", (c,)),
                 (f"bn{i+1}.gamma", (c,)), (f"bn{i+1}.beta", (c,))]
    cl = blocks[-1][1]
    E = int(embedding_dimension)
    spec += [("dense.kernel", (cl, E)), ("dense.bias", (E,))]
    if head == "uniform_euclidean":
        spec += [("head.kernel", (1, 1)), ("head.bias", (1,))]
    elif head == "weighted_l1":
        spec += [("head.kernel", (E, 1)), ("head.bias", (1,))]
    elif head == "classifier":
        spec += [("head.kernel", (E, num_classes)), ("head.bias", (num_classes,))]
    elif head is not None:
        raise NotImplementedError(head)
    offsets: Dict[str, Tuple[int, int, Tuple[int, ...]]] = OrderedDict()
    off = 0
    for name, shape in spec:
        n = int(np.prod(shape))
        offsets[name] = (off, n, shape)
        off += _align(n)
    n_flat = off
    nt_off: Dict[str, Tuple[int, int]] = OrderedDict()
    off = 0
    for i, (_, c, _) in enumerate(blocks):
        nt_off[f"bn{i+1}.moving_mean"] = (off, c)
        off += _align(c)
        nt_off[f"bn{i+1}.moving_variance"] = (off, c)
        off += _align(c)
    return spec, offsets, n_flat, nt_off, off


class FlatState:
    """The flat buffers themselves (P parameters, G gradients, M / V Adam slots, NT moving statistics) on ``device``."""

    def __init__(self, blocks, embedding_dimension, head, num_classes, device):
        self.spec, self.offsets, self.n_flat, self.nt_off, n_nt = flat_layout(blocks, embedding_dimension, head, num_classes)
        self.n_params = sum(n for _, n, _ in self.offsets.values())
        self.P = torch.zeros(self.n_flat, dtype=torch.float32, device=device)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.NT = torch.zeros(n_nt, dtype=torch.float32, device=device)
        self.iterations = 0

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        if name in self.nt_off:
            o, n = self.nt_off[name]
            return self.NT[o:o + n]
        o, n, shape = self.offsets[name]
        return (self.P if buf is None else buf)[o:o + n].view(shape)

    def refresh_weights(self):  # the engine re-derives its GEMM-layout weight copies here
        pass


# entry points whose launches bench.py books under another name: (name, index of an extra argument to drop so that the leading
# arguments line up with that entry point's)
# entry points timed under the name (and argument positions) of the plain form: name -> (plain name, argument indices to drop)
_TIMED_AS = {"vm_conv_dgrad_bnred": ("vm_conv_dgrad", ()), "vm_conv_fwd_e": ("vm_conv_fwd", (3,)),
             "vm_conv_fwd_fold": ("vm_conv_fwd", (3, 4, 6, 15)), "vm_conv_wgrad_fold": ("vm_conv_wgrad", (3,)),
             "vm_bn_pool_bwd_apply_pairs": ("vm_bn_pool_bwd_apply", (1,))}


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:   # older torch: the documented way
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class _DynI(int):
    """An integer C-ABI argument (a pointer, normally) that changes from step to step: recorded as a patch slot (see _Program)."""
    def __new__(cls, value, key):
        o = int.__new__(cls, value)
        o.key = key
        return o


class _DynF(float):
    def __new__(cls, value, key):
        o = float.__new__(cls, value)
        o.key = key
        return o


_SHARED_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def _shared_stream(device: torch.device, role: str) -> "torch.cuda.Stream":
    """The process's ONE stream per (device, role) -- every engine's tower / side / misc stream.  The runtime multiplexes streams onto a
    few hardware queues (four by default): with three private streams per engine, the streams of the fourth and sixth engine of a
    process landed on shared queues and their three-stream step ran 7.5 % slower than the first engine's (2.80 against 2.60 ms, every
    launch as fast as before when run serially: tools/probe/placement_spread.py).  Engines of one process enqueue one after the other,
    so sharing the streams costs them nothing."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SHARED_STREAMS.get((idx, role))
    if st is None:
        st = _SHARED_STREAMS[(idx, role)] = torch.cuda.Stream(device=device)
    return st


class _Program:
    """One training step as a flat list of what the host enqueued -- [0, cfunc, args, name] C-ABI calls, [1, event, stream] records,
    [2, stream, event] waits -- plus the (command, argument) slots whose value changes per step (input / label / mask pointers, the
    BatchNorm zero-debias factor, the loss scale, Adam's lr_t).  The reference runs the same train_on_batch 500 times per epoch
    (experiments/train_siamese.py:65-94); at its batch sizes (32 / 64 pairs) the step here is bound by the HOST deriving ~55 argument
    lists and stream hand-overs through Python, not by the GPU -- replaying the recorded list costs a third of it."""

    def __init__(self):
        self.cmds, self.patches, self.events = [], [], {}
        # the same list for the library's own runner (vm_program_run): segments -- int64 word arrays, or host calls between them -- and
        # the (segment, word index, type) of every per-step slot; None: this program replays through the Python loop
        self.native = None


class HipEncoderEngine:
    """The voicemap encoder (voicemap/models.py:6-41) + optional head on one MI355X.

    blocks: [(kernel_size, channels, pool)] -- first block must be (32, F, p) on a 1-channel waveform, the others
    k=3 (the only geometries the reference builds).  head: None | 'uniform_euclidean' | 'weighted_l1' |
    'classifier'.  dtype: storage type of activations / GEMM operands: 'bf16', 'f16' (half, loss-scaled), 'f32' (fp32 MFMAs) or
    'f32s' (fp32 storage, split-bf16 products).
    """

    # defaults for engines that borrow methods without running this __init__ (spectro_engine.py): no recording, no stream stack
    _rec = None
    _stream_stack = ()
    replay = False
    native_replay = True   # a recorded step is replayed by the library's own runner (vm_program_run) instead of a Python loop of ctypes calls
    _fail_at_v = ctypes.c_int64(0)
    _fail_at = ctypes.byref(_fail_at_v)
    fused_tail = False
    _packed_weights = False

    def __init__(self, blocks: Sequence[Tuple[int, int, int]], embedding_dimension: int, dropout: float = 0.05,
                 head: Optional[str] = None, num_classes: int = 0, dtype: str = DEFAULT_DTYPE, device="cuda",
                 bn_eps: float = 1e-3, bn_momentum: float = 0.99, unbiased_moving_variance: bool = True,
                 seed: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipEncoderEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = _lib.lib()
        self.timed = {}
        self._call("vm_check_device")
        self.blocks = [tuple(int(v) for v in b) for b in blocks]
        assert self.blocks[0][0] == 32, "first block must be the k=32 waveform conv"
        assert all(b[0] == 3 for b in self.blocks[1:]), "blocks 2.. must be k=3"
        assert all(b[1] % 8 == 0 for b in self.blocks), "channel counts must be multiples of 8"
        self.E = int(embedding_dimension)
        self.dropout = float(dropout)
        self.head = head
        self.num_classes = int(num_classes)
        self.dtype = _DT[dtype]
        self.tdt = _TORCH_DT[self.dtype]
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._views = {}
        self.bn_eps, self.bn_momentum = float(bn_eps), float(bn_momentum)
        self.unbiased = bool(unbiased_moving_variance)
        self.nb = len(self.blocks)
        # bf16 storage: block 1 runs the fused MFMA kernels (no full-resolution z1 / du1 in HBM); fp32 storage keeps
        # the exact fp32 vector-ALU path
        self.is16 = self.dtype in (VM_BF16, VM_F16)
        self.fuse_block1 = self.is16 and self.blocks[0][2] in (2, 4)
        self._init_loss_scale()

        # ---- flat parameter store, Keras trainable_weights order -------------------------------------
        if head not in (None, "uniform_euclidean", "weighted_l1", "classifier"):
            raise NotImplementedError(head)
        assert head != "classifier" or num_classes > 0
        st = FlatState(self.blocks, self.E, head, self.num_classes, self.device)
        self.spec, self.offsets, self.n_flat, self.n_params = st.spec, st.offsets, st.n_flat, st.n_params
        self.P, self.G, self.M, self.V = st.P, st.G, st.M, st.V
        self.nt_off, self.NT = st.nt_off, st.NT
        dev = self.device
        self._init_zero_debias()
        self.wf: Dict[int, torch.Tensor] = {}
        self.wd: Dict[int, torch.Tensor] = {}
        self.wt: Dict[int, torch.Tensor] = {}   # fp32 kernels in wf's layout: the input of the per-step BatchNorm fold (16-bit storage)
        for i in range(1, self.nb):
            cin, cout = self.blocks[i - 1][1], self.blocks[i][1]
            self.wf[i] = torch.empty(cout * 3 * cin, dtype=self.tdt, device=dev)
            self.wd[i] = torch.empty(cin * 3 * cout, dtype=self.tdt, device=dev)
            if self.is16 and self.nb - 1 <= 8:
                self.wt[i] = torch.empty(cout * 3 * cin, dtype=torch.float32, device=dev)
        # round 4: the same weights in MFMA fragment order (vm_pack_nt_weights) -- the forward / dgrad GEMMs then take them from L2
        # straight into registers (conv_nt3_kernel) instead of staging them in LDS; only where the library serves the shape
        self.packed_weights = self.is16
        self.wfp: Dict[int, torch.Tensor] = {}   # packed wf (inference forward)
        self.wdp: Dict[int, torch.Tensor] = {}   # packed wd (dgrad)
        for i in range(1, self.nb):
            cin, cout = self.blocks[i - 1][1], self.blocks[i][1]
            if self.is16 and self.lib.query("vm_pack_nt_weights_supported", cout, cin, self.dtype):
                self.wfp[i] = torch.empty(cout * 3 * cin, dtype=self.tdt, device=dev)
            if self.is16 and self.lib.query("vm_pack_nt_weights_supported", cin, cout, self.dtype):
                self.wdp[i] = torch.empty(cin * 3 * cout, dtype=self.tdt, device=dev)
        self._sq_ws = torch.empty(self.lib.query("vm_sqnorm_workspace_bytes", self.n_flat) // 8, dtype=torch.float64, device=dev)
        self._sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        # Adam(clipnorm=1.) defaults of the reference scripts
        self.lr, self.beta_1, self.beta_2, self.adam_eps, self.decay, self.clipnorm = 1e-3, 0.9, 0.999, 1e-7, 0.0, 1.0
        self.iterations = 0
        self.last_infer_l0 = 0
        # weight-gradient GEMMs only feed the optimizer: they run on a side stream, concurrently with the dgrad -> BN-backward
        # chain of the earlier blocks (the critical path of backward).  Measured +4.5 % step throughput at cfg-A (4.29 -> 4.11
        # ms); the results are bit-identical either way (tests/test_gpu_e2e.py).  Per-kernel timings of the backward pass are
        # only attributable with it off (bench.py --breakdown / --no-overlap-wgrad).
        self.overlap_wgrad = True
        # BN-backward reduce of blocks 2..n-1 from the pooled forward output instead of z (vm_bn_pool_bwd_reduce_pooled): -0.34 GB
        # of reads per step at cfg-A.  Changes the two sums by the storage rounding of the pooled tensor (1e-3 relative in
        # bf16), so it is on for bf16 storage (the throughput mode) and off for fp32 (the exact-parity mode).
        self.pooled_reduce = self.is16
        # throughput mode: the two BatchNorm-backward sums of block i come out of the epilogue of block i+1's dgrad GEMM
        # (vm_conv_dgrad_bnred) instead of a separate pass over (act, dp); only where that kernel serves the shape
        self.fused_bn_reduce = self.is16
        # ... and those sums go straight into the column reduction (vm_bn_bwd_from_sums_finalize: two small launches per block instead
        # of three, no per-window partial tensors in between); False = the separate vm_bn_bwd_from_sums + vm_bn_bwd_finalize calls
        self.fused_sums_finalize = True
        # side-stream wgrad of block i enqueued after (True) or before (False) that block's dgrad.  Round 2 (BatchNorm passes in the
        # forward, dense z in the backward): after it, beside the memory-bound passes of the block below, was 0.7 % faster.  Round 3
        # (folded forward, pair-form apply pass): before it is 1.5 % faster -- 2.888 / 2.923 ms against 2.854 / 2.863 ms, interleaved
        # on one box -- the weight-gradient GEMM then starts as soon as du exists and the step's tail (block-1 backward, slab sums,
        # optimizer) no longer waits for the last of them
        self.wgrad_after_dgrad = False
        self.defer_wgrad_tail = 0   # experiment (backward()): where the tap sums / per-tower slab folds behind every weight-gradient GEMM go
        self.misc_stream = _shared_stream(self.device, "misc")
        # training option (bf16): vm_conv_fwd_e -- the conv epilogue also writes the pool-window extreme, the pool pass reads that
        # pooled-size tensor (same bits) and the fused BatchNorm-backward sums are taken against the exact extreme.  Off by default:
        # the passes get 0.07 ms shorter and the two epilogues 0.04 ms longer, the step does not move
        self.fused_pool_extreme = False
        # training without the BatchNorm / pool pass between blocks (dropout rate 0, 16-bit storage, one tower per launch): block i+1
        # reads block i's pool extreme e with block i's BatchNorm affine folded into its weights (vm_fold_bn_weights + vm_conv_fwd_fold,
        # weight gradient vm_conv_wgrad_fold) -- the pooled BatchNorm output is never written or read, and it is no longer rounded to
        # the storage type either.  Falls back to the pass wherever a kernel does not serve a shape (_fold_ok)
        self.fold_affine = self.is16
        # ... and the folded convs leave (extreme, other element + position flag) instead of (z, extreme): same bytes as a plain
        # forward epilogue, the BatchNorm-backward apply pass reads the pair form (vm_bn_pool_bwd_apply_pairs)
        self.fold_pairs = True
        # f16 storage, round 5: the folded blocks listed here (0-based) compute and store their tile CENTRED on the per-channel pedestal
        # max(start, 0) their accumulators start from (vm_fold_bn_weights ctr_out -> vm_conv_fwd_fold e_center).  Block 2 (index 1): its
        # input is block 1's output, whose variance can sit under BatchNorm's epsilon, so a channel can be pedestal 10 x spread
        self.center_blocks = (1,)
        import os as _os
        if _os.environ.get("VOICEMAP_CENTER_BLOCKS") is not None:   # experiments (tools/probe/convergence_modes.py): "" = off, "1,2"
            self.center_blocks = tuple(int(v) for v in _os.environ["VOICEMAP_CENTER_BLOCKS"].split(",") if v.strip())
        self._fold = {}
        self.fused_infer_pool = self.is16  # inference: vm_conv_fwd_pool where the kernel serves the shape
        # round 6, measured and left OFF: the last block's forward leaves (e, o) pairs, GlobalMaxPool1D reads e alone (half of z's bytes
        # on the forward -> backward turn-around) and the sparse backward takes the pair form.  Bit-identical (tests/test_gpu_fold.py), and
        # an even trade: cfg-A 128 pairs 2.554 -> 2.555, 2.566 -> 2.571 ms -- what the pass over z saves, the pair arithmetic of the
        # forward epilogue costs (profiles/r06_wgrad_tail.txt)
        self.last_pairs = False
        self.tower_swap = False  # experiment (forward()): the first tower on the tower stream, the second on the current one
        self.tower_stagger = 0  # experiment: tower 2's forward starts after tower 1's block-1 conv (1) / whole block 1 (2)
        # (round 6, last session) the weight-gradient chain rides the TOWER stream: that stream is idle during the backward, the side
        # stream was idle during the forward, and one stream beside the main one beats two at every size -- interleaved on one box
        # (tools/probe/stream_merge_ab.py): cfg-A 128 pairs 2.559 -> 2.549 ms, 64 pairs 1.440 -> 1.427, 32 pairs 0.875 -> 0.860; cfg-B 32
        # pairs 0.482 -> 0.454 (- 5.8 %), 128 pairs 1.023 -> 1.007: fewer cross-queue hand-overs (the same direction as capping the
        # runtime at two hardware queues, GPU_MAX_HW_QUEUES=2: 0.474 -> 0.455; one queue: + 12 %).  Same launches, same bits.
        self.side_stream = _shared_stream(self.device, "tower")
        self._side_priority = 0
        # training forward: the second tower on its own stream (see forward())
        self.split_towers = True
        self.tower_stream = _shared_stream(self.device, "tower")
        # round 6: a training step whose input does not depend on the main stream (input_ready: a resident corpus + staged offsets, synthetic
        # data) runs its preprocessing on the tower stream as soon as the PREVIOUS step's block-1 backward has read x0 -- beside that
        # step's optimizer tail (reduce, norm, Adam, weight copies: six dependent ~5 us launches during which the chip is idle)
        self.pre_overlap = True
        # round 5: GlobalMaxPool1D finish -> Dense(E) -> head -> loss -> their backward as ONE launch per step (vm_tail_fwd_bwd) + one launch of
        # parameter gradients on the side stream (vm_tail_param_grads) instead of six; bit-identical (tests/test_gpu_kernels.py)
        self.fused_tail = True
        self._seg_rows = self.lib.query("vm_bn_part_rows")
        self.defer_head_reduce = True   # backward() enqueues the head's sums on the side stream (engines that borrow siamese_head: no)
        self.grad_sync = None       # callable(flat_grad_tensor) for data parallelism (parallel.py)
        self.grad_prescale = 1.0
        # SyncBN (optional; off by default as in the reference, whose BatchNormalization is per process): under data
        # parallelism the batch statistics and the two BatchNorm-backward means are taken over the GLOBAL batch -- one small all-reduce
        # per BatchNorm and direction (8 + 8 per siamese step) -- which makes N ranks x B pairs the same arithmetic as one device with
        # N x B pairs (tests/test_parallel_cpu.py).  parallel.attach() sets sync_bn_world.
        self.sync_bn = False
        self.sync_bn_world = 1
        self._sync_bufs = {}
        self._plans: Dict[Tuple, dict] = {}
        # recorded training steps (see _Program): on by default; a configuration is recorded on its second sighting and replayed from
        # the third on.  Off: data parallelism (the gradient hook runs torch.distributed calls), SyncBN, per-kernel timing (bench.py)
        self.replay = True
        self._rec: Optional[_Program] = None
        self._programs: Dict[Tuple, object] = {}
        self._stream_stack: List[int] = []
        self._drop_bufs: Dict[int, tuple] = {}
        self.init_params(seed)

    def _init_loss_scale(self):
        """f16 storage: activation gradients of this net are 1e-5 .. 1e-8 per element -- under half's normal range (6e-5) -- so the
        loss gradient is multiplied by loss_scale in the head kernel (every kernel between the head and the optimizer is linear in
        it), the flat gradient buffer G then holds loss_scale x the gradients and the optimizer kernel divides it out again
        (grad_prescale) before the clip; a step whose scaled gradients overflowed (non-finite norm) is skipped on the device and
        counted in ``skipped_steps()``.  1.0 for the other storage types: their arithmetic is untouched."""
        self.loss_scaled = (self.dtype == VM_F16)   # the switch for scaling / skipping -- NOT the scale's value
        self.loss_scale = DEFAULT_F16_LOSS_SCALE if self.loss_scaled else 1.0
        self._skipped = torch.zeros(1, dtype=torch.int32, device=self.device)   # running count, incremented by the optimizer kernel
        self._skip_seen, self._clean_checks = 0, 0
        # dynamic scale driven from optimizer_step (every train_on_batch loop gets it, not only fit_generator): every
        # ``scale_poll_every`` steps the count is copied to pinned memory behind the step; ``scale_poll_lag`` steps later -- a fixed
        # lag, so data-parallel replicas change their scale at the same step -- it is read (the copy is long done: no drain)
        self.scale_poll_every, self.scale_poll_lag = 16, 8
        self.scale_grow_after, self.scale_max, self.scale_min = 2000, 2.0 ** 24, 1.0
        # ... and the scale follows the gradient's SIZE, not only its overflows: the same poll reads the squared norm of the scaled
        # gradient buffer the optimizer kernel leaves behind; under ``scale_norm_low`` (a gradient of norm 1e-6 -- a net whose pairs
        # sit past the contrastive margin -- leaves activation gradients of 1e-10 that 4096 does not lift into half's range: measured
        # 58 % error on the first layer's gradient, 5 % with a scale of 2^20) the scale is multiplied by 8 per poll.  A scaled norm of
        # 2^6 keeps every element 2^10 under half's maximum, so the growth itself cannot overflow the parameter gradients.
        self.scale_norm_low = 64.0
        self.scale_norm_hold_polls, self._norm_growth_hold = 8, 0   # polls without a norm-driven lift after a skipped step
        self._opt_calls, self._clean_steps, self._poll = 0, 0, None
        self._skip_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.loss_scaled else None
        self._norm_host = torch.zeros(1, dtype=torch.float32).pin_memory() if self.loss_scaled else None

    def _init_zero_debias(self):
        """Keras 2.2.2 BatchNormalization updates its moving statistics with TF 1.10's assign_moving_average(zero_debias=True)
        (keras/backend/tensorflow_backend.py moving_average_update [3P]): per encoder call (tower) and statistic a zero-initialised
        "biased" accumulator and a step counter, moving = biased / (1 - momentum^step).  ZD holds the accumulators of every
        BatchNorm layer as (2 towers, 2 statistics, C) blocks; they are NOT Keras weights (a Keras checkpoint does not carry them:
        after load_model the first update overwrites the moving statistics with the de-biased batch statistics, here as there).
        ``bn_zero_debias = False`` gives the plain exponential average."""
        self.bn_zero_debias = True
        self.zd_off = OrderedDict()
        off = 0
        for i, (_, c, _) in enumerate(self.blocks):
            self.zd_off[i] = (off, c)
            off += 4 * _align(c)
        self.ZD = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.bn_steps = 0

    def _zd(self, i: int):
        """(pointer to the block's accumulators, 1 / (1 - momentum^t)) for the training forward in progress, or (None, 0)."""
        if not self.bn_zero_debias:
            return None, 0.0
        o, c = self.zd_off[i]
        return self.ZD[o:o + 4 * c].data_ptr(), 1.0 / (1.0 - self.bn_momentum ** self._bn_t)

    # ------------------------------------------------------------------------------------------------
    def stream(self):
        # the raw HIP stream of torch's current stream (asked ~25 times per step; torch.cuda.current_stream() costs 8 us a call)
        return self._stream_stack[-1] if self._stream_stack else _raw_stream(self._dev_index)

    def _sync_rows(self, key, a_ptr, b_ptr, rows_per_tower, ntw, c, row_stride, cr_ws, st):
        """SyncBN: two partial-sum tensors of ``rows_per_tower`` rows per tower -> one row per tower (vm_colsum), summed over the ranks
        (one all-reduce of 2 x ntw x C floats on the current stream).  Returns the pointers of the two reduced (ntw * row_stride, C)
        tensors: the tower's sums in its first row, zeros in the other row_stride - 1 (the finalize kernels walk row_stride rows per
        window)."""
        import torch.distributed as dist
        buf = self._sync_bufs.get((key, ntw, c, row_stride))
        if buf is None:
            buf = self._sync_bufs[(key, ntw, c, row_stride)] = torch.zeros(2, ntw * row_stride, c, dtype=torch.float32, device=self.device)
        for j, ptr in enumerate((a_ptr, b_ptr)):
            for t in range(ntw):
                self._call("vm_colsum", ptr + t * rows_per_tower * c * 4, rows_per_tower, c, buf[j, t * row_stride].data_ptr(), cr_ws, st)
        if self.sync_bn_world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return buf[0].data_ptr(), buf[1].data_ptr()

    def _bn_bwd_finalize(self, key, pa, pb, n, wpt, c, count, c1, c2, g_gamma, g_beta, cr_ws, st):
        """vm_bn_bwd_finalize; with SyncBN the two means (c1, c2) are taken over the global batch while the gamma / beta gradients stay
        this rank's (they are averaged with every other gradient later)."""
        self._call("vm_bn_bwd_finalize", pa, pb, n, wpt, c, count, c1, c2, g_gamma, g_beta, cr_ws, st)
        if not self.sync_bn:
            return
        prow = self.lib.query("vm_bn_part_rows")
        ntw = n // wpt
        ga, gb = self._sync_rows(key, pa, pb, wpt * prow, ntw, c, prow, cr_ws, st)
        scratch = self._sync_bufs.get(("gscratch", c))
        if scratch is None:
            scratch = self._sync_bufs[("gscratch", c)] = torch.empty(2, c, dtype=torch.float32, device=self.device)
        self._call("vm_bn_bwd_finalize", ga, gb, ntw, 1, c, count * self.sync_bn_world, c1, c2, scratch[0].data_ptr(), scratch[1].data_ptr(),
                   cr_ws, st)

    def _call(self, name, *args):
        """Enqueue one C-ABI entry point; entry points listed in ``self.timed`` are bracketed by HIP events on the
        launch stream (bench.py uses this for the per-kernel roofline figure)."""
        as_name, drop = _TIMED_AS.get(name, (name, ()))
        rec = self.timed.get(as_name) if self.timed else None
        if rec is None:
            self.lib.call(name, *args)
            prog = self._rec
            if prog is not None:
                a = list(args)
                for j, v in enumerate(a):
                    if isinstance(v, (_DynI, _DynF)):
                        prog.patches.append((len(prog.cmds), j, v.key))
                        a[j] = int(v) if isinstance(v, _DynI) else float(v)
                prog.cmds.append([0, getattr(self.lib.cdll, name), a, name])
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.lib.call(name, *args)
        e1.record()
        rec.append((e0, e1, tuple(a for k, a in enumerate(args) if k not in drop) + ((name,) if as_name != name else ())))

    # ---- stream hand-overs: through these, so that a recorded step (see _Program) carries them ----------------------------------
    def _dyn(self, key, value):
        """Mark a C-ABI argument whose value changes from step to step (None stays None: its presence is part of the configuration)."""
        if self._rec is None or value is None:
            return value
        return _DynF(value, key) if isinstance(value, float) else _DynI(value, key)

    class _On:
        def __init__(self, eng, stream):
            self.eng, self.stream, self.ctx = eng, stream, torch.cuda.stream(stream)

        def __enter__(self):
            self.ctx.__enter__()
            self.eng._stream_stack.append(self.stream.cuda_stream)

        def __exit__(self, *exc):
            self.eng._stream_stack.pop()
            return self.ctx.__exit__(*exc)

    def _on(self, stream):
        """``with torch.cuda.stream(stream)`` + the raw handle for stream() (no torch lookups inside)."""
        return HipEncoderEngine._On(self, stream)

    def _record(self, ev):
        """ev.record() on the current stream."""
        ev.record()
        if self._rec is not None:
            self._rec.cmds.append([1, id(ev), self.stream()])

    def _wait(self, stream, ev):
        """stream.wait_event(ev)"""
        stream.wait_event(ev)
        if self._rec is not None:
            self._rec.cmds.append([2, stream.cuda_stream, id(ev)])

    def _join(self, waiter, waited):
        """waiter.wait_stream(waited): everything enqueued on ``waited`` so far precedes what ``waiter`` gets from here on."""
        waiter.wait_stream(waited)
        if self._rec is not None:
            key = ("join", len(self._rec.cmds))
            self._rec.cmds.append([1, key, waited.cuda_stream])
            self._rec.cmds.append([2, waiter.cuda_stream, key])

    def _host_call(self, fn):
        """A step's host-side call that is not a C-ABI entry point (torch.distributed collectives): run now and, in a step being
        recorded, kept in the program at this position -- a replay calls it again between the same launches."""
        fn()
        if self._rec is not None:
            self._rec.cmds.append([3, fn, None, "host call"])

    def _begin_grad_tail(self, pl: dict):
        """Data parallelism: every gradient of G[conv2.kernel:] that the main stream writes is enqueued -- hand that range to the
        gradient hook's early collective (parallel.GradAllReduce.begin_tail), ordered behind the main stream's position and the side
        stream's weight-gradient GEMMs.  Hooks with ``ordered_begin`` get the ordering from the engine (so a recorded step carries
        it) and only enqueue the collective."""
        if "sync_ev" not in pl:
            pl["sync_ev"] = torch.cuda.Event()
        gs = self.grad_sync
        if not hasattr(gs, "ordered_begin"):
            pl["sync_ev"].record()  # main stream: every gradient of G[conv2.kernel:] that is not on the side stream
            gs.begin_tail(self, pl["sync_ev"])
            return
        self._record(pl["sync_ev"])
        self._wait(self.side_stream, pl["sync_ev"])
        self._host_call(lambda: gs.ordered_begin(self))

    def _finish_program(self, prog: _Program) -> _Program:
        """Event keys -> events of the program's own (created once; a replay never touches torch's events)."""
        for c in prog.cmds:
            if c[0] == 0 or c[0] == 3:
                continue
            slot = 1 if c[0] == 1 else 2
            h = prog.events.get(c[slot])
            if h is None:
                out = ctypes.c_void_p()
                self.lib.call("vm_event_create", ctypes.byref(out))
                h = prog.events[c[slot]] = out.value
            c[slot] = h
        self._ev_record, self._ev_wait = self.lib.cdll.vm_event_record, self.lib.cdll.vm_stream_wait_event
        prog.native = self._native_program(prog)   # (used while engine.native_replay is on)
        return prog

    @staticmethod
    def _word(v, t):
        """One argument as the int64 word vm_program_run expects: pointers / integers as they are, floats / doubles as their bits."""
        if t == "F":
            return struct.unpack("<I", struct.pack("<f", float(v)))[0]
        if t == "D":
            return struct.unpack("<q", struct.pack("<d", float(v)))[0]
        if v is None:
            return 0
        if isinstance(v, (ctypes.Array, ctypes.Structure)):   # a host-side argument block (pointer / size tables of the batched entry
            v = ctypes.addressof(v)                           # points): its address -- the object stays alive in the program's list
        elif isinstance(v, ctypes._SimpleCData):
            v = v.value or 0
        v = int(v)
        return v - (1 << 64) if v >= (1 << 63) else v

    def _native_program(self, prog: _Program):
        """The recorded step as word arrays for vm_program_run (include/voicemap_hip.h): one array per run of C-ABI calls and event
        records / waits, host calls (the gradient collectives of data parallelism) between them.  None where the library's table is
        not the binding's or an argument is not a pointer / number."""
        tab = _lib.program_table()
        if tab is None:
            return None
        ids, sigs = tab
        segs, cur, where = [], [], {}
        try:
            for ci, c in enumerate(prog.cmds):
                if c[0] == 3:
                    segs.append(cur)
                    segs.append(c[1])
                    cur = []
                    continue
                if c[0] == 0:
                    name, args = c[3], c[2]
                elif c[0] == 1:
                    name, args = "vm_event_record", (c[1], c[2])
                else:
                    name, args = "vm_stream_wait_event", (c[1], c[2])
                sig = sigs[name]
                if len(sig) != len(args):
                    return None
                where[ci] = (len(segs), len(cur) + 2)
                cur += [ids[name], len(args)] + [self._word(v, t) for v, t in zip(args, sig)]
            segs.append(cur)
            patches = []
            for ci, ai, key in prog.patches:
                si, w0 = where[ci]
                patches.append((si, w0 + ai, sigs[prog.cmds[ci][3]][ai], key))
        except (KeyError, TypeError, ValueError, struct.error):
            return None
        segs = [np.array(sg, dtype=np.int64) if isinstance(sg, list) else sg for sg in segs]
        return segs, patches

    def _destroy_program(self, prog):
        if isinstance(prog, _Program):
            for h in prog.events.values():
                self.lib.cdll.vm_event_destroy(h)
            prog.events = {}

    def _drop_programs(self):
        """Forget every recorded step.  Programs hold raw device pointers: whoever frees or reallocates a buffer a program may
        reference (the fold buffers, a plan's lazily sized buffers, a stream) calls this."""
        progs = getattr(self, "_programs", None)
        if progs:
            if torch.cuda.is_available():
                torch.cuda.synchronize(self.device)   # a replayed step may still be in flight on the events about to go
            for prog in progs.values():
                self._destroy_program(prog)
            progs.clear()

    def __del__(self):
        # the recorded programs' events are the only library-side objects an engine owns
        try:
            for prog in getattr(self, "_programs", {}).values():
                self._destroy_program(prog)
        except Exception:   # interpreter shutdown: the library may be gone already
            pass

    def _run_program(self, prog: _Program, dyn: dict):
        if prog.native is not None and self.native_replay:
            segs, patches = prog.native
            word = self._word
            for si, wi, t, key in patches:
                segs[si][wi] = word(dyn[key], t)
            run, fail = self.lib.cdll.vm_program_run, self._fail_at
            for sg in segs:
                if isinstance(sg, np.ndarray):
                    if sg.size:
                        rc = run(sg.ctypes.data, sg.size, fail)
                        if rc != 0:
                            msg = self.lib.cdll.vm_last_error()
                            raise _lib.VoicemapHipError("a replayed step failed (%d) at word %d of its program: %s"
                                                        % (rc, self._fail_at_v.value, msg.decode() if msg else ""))
                else:
                    sg()          # a host call of the step (the gradient collectives of data parallelism): _host_call
            return
        cmds = prog.cmds
        for ci, ai, key in prog.patches:
            cmds[ci][2][ai] = dyn[key]
        rec, wait = self._ev_record, self._ev_wait
        for c in cmds:
            k = c[0]
            if k == 0:
                rc = c[1](*c[2])
            elif k == 1:
                rc = rec(c[1], c[2])
            elif k == 2:
                rc = wait(c[1], c[2])
            else:
                c[1]()          # a host call of the step (the gradient collectives of data parallelism): _host_call
                rc = 0
            if rc != 0:
                msg = self.lib.cdll.vm_last_error()
                raise _lib.VoicemapHipError("%s failed (%d) in a replayed step: %s" % (c[3] if k == 0 else "stream ordering", rc,
                                                                                       msg.decode() if msg else ""))

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        # (memoised per (name, buffer): a training step asks ~70 times, and at small batches the step is bound by the host)
        nt = name in self.nt_off
        base = self.NT if nt else (self.P if buf is None else buf)
        key = (name, None if (nt or buf is None) else id(buf))
        v = self._views.get(key)
        if v is not None and v[0] is base:
            return v[1]
        if nt:
            o, n = self.nt_off[name]
            out = self.NT[o:o + n]
        else:
            o, n, shape = self.offsets[name]
            out = base[o:o + n].view(shape)
        self._views[key] = (base, out)
        return out

    def init_params(self, seed: Optional[int] = None):
        """Keras default initialisers (glorot_uniform kernels, zero biases, gamma 1, beta 0, moving 0 / 1)."""
        if seed is None:
            # like Keras' unseeded initialisers: a fresh draw per model, reproducible through torch.manual_seed
            # (experiments/_common.seed_everything sets it)
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.seed = int(seed)
        g = torch.Generator().manual_seed(self.seed)
        # SpatialDropout1D masks come from an engine-owned generator: reproducible under the same seed, decorrelated across
        # data-parallel ranks (every rank holds the same weights but must drop different channels)
        import os as _os
        self._drop_gen = torch.Generator(device=self.device)
        self._drop_gen.manual_seed(self.seed * 1000003 + 7919 * (int(_os.environ.get("RANK", "0")) + 1))
        self.P.zero_()
        for name, (o, n, shape) in self.offsets.items():
            if name.endswith(".kernel"):
                if len(shape) == 3:
                    fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
                else:
                    fan_in, fan_out = shape
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                w = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim
                self.P[o:o + n] = w.flatten().to(torch.float32).to(self.device)
            elif name.endswith(".gamma"):
                self.P[o:o + n] = 1.0
        for name, (o, n) in self.nt_off.items():
            self.NT[o:o + n] = 1.0 if name.endswith("moving_variance") else 0.0
        self.M.zero_()
        self.V.zero_()
        self.iterations = 0
        if hasattr(self, "ZD"):
            self.ZD.zero_()
            self.bn_steps = 0
        self.refresh_weights()

    def set_params(self, params: Dict[str, "np.ndarray"]):
        for name, val in params.items():
            t = torch.as_tensor(np.asarray(val, dtype=np.float32)).to(self.device)
            v = self.view(name)
            v.copy_(t.reshape(v.shape))
        self.refresh_weights()

    def get_params(self) -> "OrderedDict[str, np.ndarray]":
        out = OrderedDict()
        for name in list(self.offsets) + list(self.nt_off):
            out[name] = self.view(name).detach().cpu().numpy().copy()
        return out

    def get_grads(self) -> "OrderedDict[str, np.ndarray]":
        """The gradients of the last backward pass (G holds loss_scale x them with f16 storage: divided out here)."""
        inv = 1.0 / float(self.loss_scale)
        return OrderedDict((name, self.view(name, self.G).detach().cpu().numpy().copy() * np.float32(inv)) for name in self.offsets)

    def skipped_steps(self) -> int:
        """Optimizer steps skipped because the loss-scaled gradients were not finite (f16 storage only; synchronises)."""
        return int(self._skipped.item())

    def _account_skips(self, total: int) -> int:
        """``total`` skipped steps so far: the new ones are taken off ``iterations`` again (a skipped step is not an Adam step: its
        bias correction and lr decay must not advance) and reported once."""
        # two readers feed this (the lagged pinned copy of _poll_loss_scale and the live counter of adjust_loss_scale /
        # skipped_steps): a stale total must neither move the watermark backwards nor count a step twice
        new = max(0, total - self._skip_seen)
        self._skip_seen = max(self._skip_seen, total)
        if new > 0:
            self.iterations = max(0, self.iterations - new)
            import warnings
            warnings.warn("f16 storage: %d optimizer step(s) skipped (non-finite loss-scaled gradients, scale %g); %d in total"
                          % (new, self.loss_scale, total), RuntimeWarning, stacklevel=3)
        return new

    def adjust_loss_scale(self, grow_after: int = 4, max_scale: float = 65536.0) -> float:
        """Dynamic loss scaling at the caller's cadence (one device read per call; ``optimizer_step`` does the same on its own
        without a read on the critical path, see ``_poll_loss_scale``): steps were skipped since the last look -> halve the scale
        once per skipped step (at most 2^-4); none for ``grow_after`` calls in a row -> double it.  No-op for the storage types
        that do not scale.  The scale may reach ``scale_min`` (1.0) and grows back from there."""
        if not self.loss_scaled:
            return 1.0
        new = self._account_skips(self.skipped_steps())
        self._poll = None   # a copy in flight predates this read: consumed here, not a second time 8 steps later
        if new > 0:
            self.loss_scale = max(self.loss_scale / float(2 ** min(new, 4)), self.scale_min)
            self._clean_checks = self._clean_steps = 0
            self._norm_growth_hold = self.scale_norm_hold_polls
        else:
            self._clean_checks += 1
            if self._clean_checks >= grow_after:
                self.loss_scale = min(self.loss_scale * 2.0, max_scale)
                self._clean_checks = 0
        return self.loss_scale

    def calibrate_loss_scale(self, run_backward, max_rounds: int = 8) -> float:
        """The norm-driven part of the loss-scale search, synchronously: ``run_backward()`` runs one forward + backward (no update)
        and leaves the scaled gradient in G; while its norm (as Adam will see it after grad_prescale) is under ``scale_norm_low`` the
        scale is multiplied by 8 -- what ``_poll_loss_scale`` does every ``scale_poll_every`` steps of a training loop -- and halved
        on an overflow.  For callers that take ONE step from a cold start (tests, a first evaluation of gradients)."""
        if not self.loss_scaled:
            return 1.0
        for _ in range(max_rounds):
            run_backward()
            g = float(self.G.double().norm().item()) * float(self.grad_prescale)
            if not math.isfinite(g):
                self.loss_scale = max(self.loss_scale / 2.0, self.scale_min)
            elif 0.0 < g < self.scale_norm_low and self.loss_scale < self.scale_max:
                self.loss_scale = min(self.loss_scale * 8.0, self.scale_max)
            else:
                break
        return self.loss_scale

    def _poll_loss_scale(self):
        """Called once per optimizer step (f16 storage).  Step k with k % scale_poll_every == 0 enqueues a copy of the device's skip
        count to pinned memory; step k + scale_poll_lag consumes it."""
        k = self._opt_calls
        self._opt_calls += 1
        if self._poll is not None and k >= self._poll[0] + self.scale_poll_lag:
            k0, ev = self._poll
            self._poll = None
            ev.synchronize()   # enqueued scale_poll_lag steps ago
            new = self._account_skips(int(self._skip_host[0]))
            sq = float(self._norm_host[0])   # sum of squares of the scaled gradient buffer at the polled step (inf / nan: a skipped one)
            # G holds the SUM over data-parallel ranks; grad_prescale (1 / world) is applied in the clip: compare what Adam sees
            sq *= float(getattr(self, "grad_prescale", 1.0)) ** 2
            if new > 0:
                self.loss_scale = max(self.loss_scale / float(2 ** min(new, 4)), self.scale_min)
                self._clean_steps = 0
                self._norm_growth_hold = self.scale_norm_hold_polls   # no x8 lifts right after an overflow (limit cycle)
            elif (math.isfinite(sq) and 0.0 < sq < self.scale_norm_low ** 2 and self.loss_scale < self.scale_max
                  and self._norm_growth_hold == 0):
                self.loss_scale = min(self.loss_scale * 8.0, self.scale_max)   # a tiny gradient: lift it (see _init_loss_scale)
                self._clean_steps = 0
            else:
                self._norm_growth_hold = max(0, self._norm_growth_hold - 1)
                self._clean_steps += self.scale_poll_every
                if self._clean_steps >= self.scale_grow_after:
                    self.loss_scale = min(self.loss_scale * 2.0, self.scale_max)
                    self._clean_steps = 0
        if self._poll is None and k % self.scale_poll_every == 0:
            with torch.cuda.stream(torch.cuda.current_stream(self.device)):
                self._skip_host.copy_(self._skipped, non_blocking=True)
                self._norm_host.copy_(self._sqnorm, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._poll = (k, ev)

    @property
    def packed_weights(self):
        return self._packed_weights

    @packed_weights.setter
    def packed_weights(self, v):
        """conv_nt3_kernel's fragment-order weight copies on / off.  Turning them on later (bench.py --tune, a test) re-derives every
        copy and drops what cached the decision (the folded-weight buffers, the pack argument tables): no stale or uninitialised
        packed operand can reach a kernel."""
        changed = bool(v) != getattr(self, "_packed_weights", None)
        self._packed_weights = bool(v)
        if changed and hasattr(self, "_plans"):   # (during __init__ the copies are made by init_params -> refresh_weights)
            self._drop_programs()   # recorded steps point into the fold buffers dropped below
            self._fold = {}
            self._pack_args = None
            self._wfp_stale = True
            self.refresh_weights()

    @property
    def side_priority(self):
        return self._side_priority

    @side_priority.setter
    def side_priority(self, v):
        """experiment (bench.py --tune side_priority=N): the HIP priority of the side stream the weight-gradient GEMMs run on"""
        if int(v) == self._side_priority:
            return
        self._side_priority = int(v)
        self._drop_programs()
        # 0: back on the process's shared stream; anything else is a private stream of that priority (an experiment: see _shared_stream
        # for what private streams cost the later engines of a process)
        self.side_stream = _shared_stream(self.device, "tower") if int(v) == 0 else torch.cuda.Stream(device=self.device, priority=int(v))

    def refresh_weights(self):
        """fp32 master conv kernels -> GEMM-layout copies in the storage dtype (wf: forward, wd: dgrad)."""
        nl = self.nb - 1
        if nl < 1:
            return
        if nl > 8:
            for i in range(1, self.nb):
                cin, cout = self.blocks[i - 1][1], self.blocks[i][1]
                self._call("vm_prep_conv_weights", _p(self.view(f"conv{i+1}.kernel")), cin, cout, self.dtype,
                           _p(self.wf[i]), _p(self.wd[i]), self.stream())
            self._pack_weights()
            return
        if getattr(self, "_prep_args", None) is None:  # the flat buffers never move: build the pointer tables once
            vp, ci = ctypes.c_void_p * nl, ctypes.c_int * nl
            self._prep_args = (vp(*[_p(self.view(f"conv{i+1}.kernel")) for i in range(1, self.nb)]),
                               ci(*[self.blocks[i - 1][1] for i in range(1, self.nb)]),
                               ci(*[self.blocks[i][1] for i in range(1, self.nb)]),
                               vp(*[_p(self.wf[i]) for i in range(1, self.nb)]), vp(*[_p(self.wd[i]) for i in range(1, self.nb)]),
                               vp(*[_p(self.wt[i]) for i in range(1, self.nb)]) if self.wt else None)
        w, cin, cout, wf, wd, wt = self._prep_args
        self._call("vm_prep_conv_weights_batch", nl, w, cin, cout, self.dtype, wf, wd, wt, self.stream())
        self._pack_weights()

    def _pack_weights(self):
        """wd -> its fragment-order copy after every refresh of the GEMM-layout copies (one launch for all layers); the packed wf is
        only read by the inference forward (vm_conv_fwd_pool) and is re-made lazily (``_ensure_wfp``)."""
        self._wfp_stale = True
        if not getattr(self, "packed_weights", False) or not self.wdp:
            return
        if getattr(self, "_pack_args", None) is None:
            ids = sorted(self.wdp)
            k = len(ids)
            vp, ci = ctypes.c_void_p * k, ctypes.c_int * k
            self._pack_args = (k, vp(*[_p(self.wd[i]) for i in ids]), ci(*[1] * k), ci(*[self.blocks[i - 1][1] for i in ids]),
                               ci(*[self.blocks[i][1] for i in ids]), vp(*[_p(self.wdp[i]) for i in ids]))
        k, bt, tw, rows, ac, out = self._pack_args
        if k <= 8:
            self._call("vm_pack_nt_weights_batch", k, bt, tw, rows, ac, self.dtype, out, self.stream())
        else:
            for i in sorted(self.wdp):
                self._call("vm_pack_nt_weights", _p(self.wd[i]), 1, self.blocks[i - 1][1], self.blocks[i][1], self.dtype, _p(self.wdp[i]), self.stream())

    def _ensure_wfp(self):
        """The packed forward weights of the inference path, re-made when the parameters changed since the last inference forward."""
        if not getattr(self, "_wfp_stale", True) or not self.packed_weights or not self.wfp:
            return
        ids = sorted(self.wfp)
        for j in range(0, len(ids), 8):
            part = ids[j:j + 8]
            k = len(part)
            vp, ci = ctypes.c_void_p * k, ctypes.c_int * k
            self._call("vm_pack_nt_weights_batch", k, vp(*[_p(self.wf[i]) for i in part]), ci(*[1] * k), ci(*[self.blocks[i][1] for i in part]),
                       ci(*[self.blocks[i - 1][1] for i in part]), self.dtype, vp(*[_p(self.wfp[i]) for i in part]), self.stream())
        self._wfp_stale = False

    # ------------------------------------------------------------------------------------------------
    def lengths(self, l0: int) -> List[int]:
        out = [l0]
        for (_, _, p) in self.blocks:
            out.append(out[-1] // p)
        return out

    def plan(self, n_windows: int, l0: int, training: bool) -> dict:
        key = (n_windows, l0, training)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        dev, tdt = self.device, self.tdt
        ls = self.lengths(l0)
        assert ls[-1] >= 1, "window too short for this encoder"
        pl = {"n": n_windows, "l0": l0, "L": ls, "training": training}
        f32 = torch.float32
        pl["x0"] = torch.zeros(n_windows, l0 + CONV1_HALO, dtype=f32, device=dev)
        prow = self.lib.query("vm_bn_part_rows")
        for i, (k, c, pool) in enumerate(self.blocks):
            L = ls[i]
            b = {}
            fused = (i == 0 and self.fuse_block1)
            if fused:
                b["e"] = torch.empty(n_windows, ls[1], c, dtype=tdt, device=dev)  # pooled extreme of relu(conv)
            else:
                b["z"] = torch.empty(n_windows, L, c, dtype=tdt, device=dev)
            b["act"] = torch.zeros(n_windows, ls[i + 1] + 2, c, dtype=tdt, device=dev)  # halo rows stay zero
            rows = self.lib.query("vm_conv1_stat_rows" if i == 0 else "vm_conv_stat_rows", L)
            b["stat_rows"] = rows
            for nm in ("mean", "invstd", "scale", "shift", "c1", "c2") + (("shift_c", "mean_c", "ctr") if i < self.nb - 1 else ()):
                b[nm] = torch.zeros(2, c, dtype=f32, device=dev)
            if training:
                b["ssum"] = torch.empty(n_windows * rows, c, dtype=f32, device=dev)
                b["ssq"] = torch.empty(n_windows * rows, c, dtype=f32, device=dev)
                b["dp"] = torch.empty(n_windows, ls[i + 1], c, dtype=tdt, device=dev)
                if not fused:
                    b["du"] = torch.zeros(n_windows, L + 2, c, dtype=tdt, device=dev)
                for nm in ("pa", "pb", "pdu"):
                    b[nm] = torch.empty(n_windows * prow, c, dtype=f32, device=dev)
            pl[i] = b
        cl = self.blocks[-1][1]
        pl["gmax_ws"] = torch.empty(self.lib.query("vm_bn_drop_pool_gmax_workspace_bytes", n_windows, cl) // 4, dtype=f32,
                                    device=dev)
        pl["gmax"] = torch.empty(n_windows, cl, dtype=f32, device=dev)
        pl["gidx"] = torch.empty(n_windows, cl, dtype=torch.int32, device=dev)
        pl["emb"] = torch.empty(n_windows, self.E, dtype=f32, device=dev)
        if training:
            pl["demb"] = torch.zeros(n_windows, self.E, dtype=f32, device=dev)
            pl["dgmax"] = torch.empty(n_windows, cl, dtype=f32, device=dev)
            ws = 0
            for i in range(1, self.nb):
                wsi = self.lib.query("vm_conv_wgrad_workspace_bytes", n_windows, ls[i], self.blocks[i - 1][1], self.blocks[i][1])
                pl[i]["wgrad_ws"] = torch.empty(wsi // 4 + 16, dtype=f32, device=dev)  # one per block: wgrads may overlap
                pl[i]["ev"] = torch.cuda.Event()
            ws = max(ws, self.lib.query("vm_conv1_wgrad_workspace_bytes", n_windows, self.blocks[0][1]))
            if self.fuse_block1:
                ws = max(ws, self.lib.query("vm_conv1_fused_bwd_workspace_bytes", n_windows, l0, self.blocks[0][1]))
            pl["wgrad_ws"] = torch.empty(ws // 4 + 16, dtype=f32, device=dev)
        pl["pred"] = torch.empty(max(n_windows // 2, 1), dtype=f32, device=dev)
        pl["loss_acc"] = torch.zeros(2, dtype=f32, device=dev)
        pl["head_ws"] = torch.empty(4 * max(n_windows // 2, 1), dtype=f32, device=dev)
        if self.head == "classifier":
            pl["logits"] = torch.empty(n_windows, self.num_classes, dtype=f32, device=dev)
            pl["prob"] = torch.empty_like(pl["logits"])
            pl["dlogits"] = torch.empty_like(pl["logits"])
            pl["cce_ws"] = torch.empty(2 * n_windows, dtype=f32, device=dev)
        pl["pre_ws"] = torch.empty(self.lib.query("vm_decimate_whiten_workspace_bytes", n_windows) // 8, dtype=torch.float64,
                                   device=dev)
        cmax = max(b[1] for b in self.blocks)
        pl["cr_ws"] = torch.empty(self.lib.query("vm_colreduce_workspace_bytes", 2, cmax) // 8, dtype=torch.float64, device=dev)
        pl["cr_ws_side"] = torch.empty_like(pl["cr_ws"])  # reductions enqueued on the side stream (backward)
        self._plans[key] = pl
        return pl

    # ------------------------------------------------------------------------------------------------
    def load_preprocessed(self, pl: dict, x: torch.Tensor):
        """x: (n_windows, L0) or (n_windows, L0, 1) already decimated + whitened (host path of the reference)."""
        x = x.reshape(pl["n"], pl["l0"]).to(self.device, torch.float32)
        pl["x0"][:, CONV1_HALO_L:CONV1_HALO_L + pl["l0"]].copy_(x)

    def preprocess(self, pl: dict, raw: torch.Tensor, downsampling: int, whitening: bool, windows_per_tower: int,
                   rms: float = 0.038021, offsets: Optional[torch.Tensor] = None, raw_len: Optional[int] = None):
        """voicemap/utils.py:22-34 + 88-101 on the GPU: raw (n_windows, raw_len) fp32 or int16 -> pl['x0'].
        With ``offsets`` (n_windows int64 on the device) ``raw`` is instead a resident 1-D buffer of decoded recordings
        (voicemap_amd/shards.py) and window n is the ``raw_len`` samples starting at raw[offsets[n]]: the crop of
        voicemap/librispeech.py:103-137 happens on the device and the host only chooses offsets."""
        is16 = raw.dtype == torch.int16
        if offsets is not None:
            assert raw.dim() == 1 and raw.is_contiguous() and raw_len is not None
            assert offsets.dtype == torch.int64 and offsets.numel() == pl["n"] and offsets.is_cuda
            assert (raw_len + downsampling - 1) // downsampling == pl["l0"]
            if not is16:
                assert raw.dtype == torch.float32
            self._call("vm_crop_decimate_whiten", self._dyn("raw", _p(raw)), int(is16), self._dyn("offsets", _p(offsets)), pl["n"], raw_len, downsampling,
                       int(whitening), rms, windows_per_tower, _p(pl["x0"]), _p(pl["pre_ws"]), self.stream())
            return
        raw = raw.reshape(pl["n"], -1).contiguous()
        if not is16:
            raw = raw.to(torch.float32)
        assert (raw.shape[1] + downsampling - 1) // downsampling == pl["l0"]
        self._call("vm_decimate_whiten", self._dyn("raw", _p(raw)), int(is16), pl["n"], raw.shape[1], downsampling, int(whitening), rms,
                      windows_per_tower, _p(pl["x0"]), _p(pl["pre_ws"]), self.stream())

    def _fold_ok(self, pl: dict, wpt: int, drop_masks) -> bool:
        """Can this training forward run blocks 2.. on the pool extremes with the BatchNorm affines folded into the weights?  Asked per
        call (the answer follows the dropout masks and vm_set_tuning)."""
        if not (self.fold_affine and pl["training"] and self.is16 and self.fuse_block1 and self.fused_bn_reduce
                and self.fused_sums_finalize and self.nb >= 2 and self.wt):
            return False
        if self.sync_bn:
            return False   # SyncBN goes through the two plain finalize funnels (vm_bn_finalize, vm_bn_bwd_finalize): the unfolded path
        if drop_masks is not None and any(m is not None for m in drop_masks):
            return False   # SpatialDropout1D scales per (window, channel): not a per-channel affine
        n, dt = pl["n"], self.dtype
        for i in range(1, self.nb):
            cin, c, L = self.blocks[i - 1][1], self.blocks[i][1], pl["L"][i]
            with_e = i < self.nb - 1
            if with_e and self.blocks[i][2] != 2:
                return False
            if not self.lib.query("vm_conv_fwd_fold_supported", n, L, cin, c, dt, int(with_e)):
                return False
            if not self.lib.query("vm_conv_dgrad_bnred_supported", n, L, cin, c, dt):
                return False
        return True

    def _fold_bufs(self, i: int):
        """Per tower: the folded forward weights of block i (0-based) and the per-tap constants hb (3, c_out)."""
        if i not in self._fold:
            cin, cout = self.blocks[i - 1][1], self.blocks[i][1]
            packed = self.packed_weights and self.lib.query("vm_pack_nt_weights_supported", cout, cin, self.dtype)
            self._fold[i] = (torch.empty(2, cout * 3 * cin, dtype=self.tdt, device=self.device),
                             torch.empty(2, 4, cout, dtype=torch.float32, device=self.device),   # rows 0..2 per tap, row 3 = bias + their sum
                             torch.empty(2, cout * 3 * cin, dtype=self.tdt, device=self.device) if packed else None)
        return self._fold[i]

    def _fold_plan(self, pl: dict):
        """Buffers of the folded forward / backward, on first use: padded pool extremes (zero halo rows), tap sums, wgrad slabs."""
        if "fold_ready" in pl:
            return
        n, ls, wpt = pl["n"], pl["L"], pl["wpt"]
        for i in range(self.nb - 1 if not (self.last_pairs and self.nb > 1) else self.nb):
            c = self.blocks[i][1]
            pl[i]["ep"] = torch.zeros(n, ls[i + 1] + 2, c, dtype=self.tdt, device=self.device)
            if i > 0:
                pl[i]["o"] = torch.empty(n, ls[i + 1], c, dtype=self.tdt, device=self.device)
        for i in range(1, self.nb):
            cin, c = self.blocks[i - 1][1], self.blocks[i][1]
            pl[i]["dsum"] = torch.empty(2, 3, c, dtype=torch.float32, device=self.device)
            ws = self.lib.query("vm_conv_wgrad_fold_workspace_bytes", n, wpt, ls[i], cin, c)
            pl[i]["wgrad_ws_fold"] = torch.empty(ws // 4 + 16, dtype=torch.float32, device=self.device)
        pl["fold_ready"] = wpt

    def forward(self, pl: dict, windows_per_tower: int, drop_masks: Optional[Sequence[Optional[torch.Tensor]]] = None,
                defer_tail: bool = False):
        """x0 -> embeddings (pl['emb']).  Training plans use batch statistics per tower and update the moving
        statistics; inference plans use the moving statistics (Keras learning phase 0).

        ``split_towers`` (training, two towers): the two encoder calls of the siamese model (voicemap/models.py:52-53) are
        ``defer_tail`` (the siamese training steps): the caller goes on to ``siamese_head(pl, y)`` and ``backward(pl)``; the finish of the
        global max, the dense layer, the head, the loss and their backward then run as one launch there (vm_tail_fwd_bwd) and
        pl['gmax'] / pl['emb'] are not valid before it.

        independent until the head, so tower 2's kernel sequence is enqueued on a second stream over the second half of the
        same buffers: a GEMM of one tower then runs next to a streaming BatchNorm pass of the other (8.6 % off the forward at
        cfg-A forward alone, 1.5 % off the step, tools/tower_pipeline_probe.py).  Every launch computes what its half of the
        one-launch form computes; the only difference is the order of a few fp32 partial sums whose chunking depends on the launch
        size (block 1), i.e. results agree to rounding and stay run-to-run bit-identical (tests/test_gpu_e2e.py)."""
        n = pl["n"]
        training = pl["training"]
        n_towers = n // windows_per_tower if training else 1
        assert (not training) or n % windows_per_tower == 0
        assert n_towers <= 2 or not training, "at most two towers per call"
        wpt = windows_per_tower if training else n
        pl["wpt"], pl["drop"] = wpt, drop_masks
        split = training and n_towers == 2 and self.split_towers
        fold = training and self._fold_ok(pl, wpt, drop_masks)
        pl["fold_now"] = fold
        cl = self.blocks[-1][1]
        tail = bool(defer_tail and training and n_towers == 2 and self.fused_tail and self.head in HEADS
                    and self.lib.query("vm_tail_fwd_bwd_supported", cl, self.E))
        pl["tail_pending"] = tail
        pl["tail_parts"] = tail   # the last block leaves segment partials (cleared by _forward_range where it runs vm_global_maxpool_fwd)
        if fold:
            if pl.get("fold_ready", wpt) != wpt:   # the slab split depends on the tower size
                del pl["fold_ready"]
                self._drop_programs()              # steps recorded for the other tower size point at the buffers re-made below
            self._fold_plan(pl)
        if training:
            self.bn_steps += 1
            self._bn_t = self.bn_steps
        else:
            self._ensure_wfp()
        if split:
            if "cr_ws_t2" not in pl:
                pl["cr_ws_t2"] = torch.empty_like(pl["cr_ws"])
                pl["gmax_ws_t2"] = torch.empty_like(pl["gmax_ws"])
                pl["mov_scratch"] = torch.empty(2 * max(b[1] for b in self.blocks), dtype=torch.float32, device=self.device)
                pl["tower_ev"] = [torch.cuda.Event() for _ in self.blocks]
            cur = torch.cuda.current_stream(self.device)
            self._join(self.tower_stream, cur)   # the pre-processed windows are ready
            pl["t2_stream"] = cur if self.tower_swap else self.tower_stream   # the stream the second tower's chain is on
            if self.tower_swap:
                # (round 6) the chain that is enqueued SECOND finishes last, and what follows the forward (tail, backward) is on the
                # current stream: with the first tower on the tower stream and the second on the current one, the hand-over at the
                # end of the forward waits for a chain that finished long ago instead of crossing queues behind the last kernel
                with self._on(self.tower_stream):
                    self._forward_range(pl, 0, wpt, 0, 1, wpt, drop_masks, pl["cr_ws"], pl["gmax_ws"], first_of_two=True, fold=fold)
                self._forward_range(pl, wpt, wpt, 1, 1, wpt, drop_masks, pl["cr_ws_t2"], pl["gmax_ws_t2"], second_of_two=True, fold=fold)
            else:
                self._forward_range(pl, 0, wpt, 0, 1, wpt, drop_masks, pl["cr_ws"], pl["gmax_ws"], first_of_two=True, fold=fold)
                if self.tower_stagger and "stagger_ev" in pl:
                    self._wait(self.tower_stream, pl["stagger_ev"])
                with self._on(self.tower_stream):
                    self._forward_range(pl, wpt, wpt, 1, 1, wpt, drop_masks, pl["cr_ws_t2"], pl["gmax_ws_t2"], second_of_two=True, fold=fold)
            self._join(cur, self.tower_stream)
        else:
            self._forward_range(pl, 0, n, 0, n_towers, wpt, drop_masks, pl["cr_ws"], pl["gmax_ws"], fold=fold)
        if not tail:
            self._call("vm_dense_fwd", _p(pl["gmax"]), _p(self.view("dense.kernel")), _p(self.view("dense.bias")), n, cl, self.E,
                       _p(pl["emb"]), self.stream())
        return pl["emb"]

    def _forward_range(self, pl: dict, w0: int, nw: int, tw0: int, ntw: int, wpt: int, drop_masks, cr_ws, gmax_ws,
                       first_of_two: bool = False, second_of_two: bool = False, fold: bool = False):
        """The encoder blocks for windows [w0, w0 + nw) = towers [tw0, tw0 + ntw) on the current stream."""
        st, dt, training = self.stream(), self.dtype, pl["training"]

        def W(t, per_window=1):   # rows of a per-window tensor that belong to this range
            return t[w0 * per_window:(w0 + nw) * per_window].data_ptr()

        def T(t):                 # rows of a per-tower (2, C) tensor
            return t[tw0:].data_ptr()

        fused_tail = False
        for i, (k, c, pool) in enumerate(self.blocks):
            b, L = pl[i], pl["L"][i]
            rows = b["stat_rows"]
            ssum = W(b["ssum"], rows) if training else None
            ssq = W(b["ssq"], rows) if training else None
            bias = _p(self.view(f"conv{i+1}.bias"))
            gam, bet = _p(self.view(f"bn{i+1}.gamma")), _p(self.view(f"bn{i+1}.beta"))
            mm, mv = _p(self.view(f"bn{i+1}.moving_mean")), _p(self.view(f"bn{i+1}.moving_variance"))
            dm = self._dyn(("drop", i, w0), W(drop_masks[i])) if (training and drop_masks is not None and drop_masks[i] is not None) else None

            def finalize():
                zd, zc = self._zd(i)
                zc = self._dyn("zc", zc)
                m_, v_ = mm, mv
                if zd is not None:
                    zd += tw0 * 2 * c * 4   # this tower's accumulators
                    if first_of_two:        # the moving statistic ends as the LAST tower's de-biased average: tower 1 of 2 only
                        m_ = pl["mov_scratch"].data_ptr()   # updates its accumulators
                        v_ = m_ + 4 * c
                elif second_of_two:         # plain average: the two updates are sequential -- after tower 1's
                    self._wait(pl.get("t2_stream", self.tower_stream), pl["tower_ev"][i])
                centred = i == 0 and fold and self.fuse_block1   # block 1's extreme is stored as e - max(bias, 0): the offset's constants
                s_sum, s_sq, s_rows, s_cnt = ssum, ssq, wpt * rows, float(wpt * L)
                if self.sync_bn:
                    s_sum, s_sq = self._sync_rows(("fwd", i, tw0), ssum, ssq, wpt * rows, ntw, c, 1, _p(cr_ws), st)
                    s_rows, s_cnt = 1, float(wpt * L) * self.sync_bn_world
                tile_ctr = bool(b.get("ctr_now")) and i > 0   # this block's tile (statistics and stored extreme) is centred: vm_conv_fwd_fold e_center
                adj = centred or tile_ctr
                self._call("vm_bn_finalize", s_sum, s_sq, s_rows, ntw, c, s_cnt, gam, bet, self.bn_eps, self.bn_momentum,
                           int(self.unbiased), m_, v_, T(b["mean"]), T(b["invstd"]), T(b["scale"]), T(b["shift"]), _p(cr_ws), zd, zc,
                           bias if centred else None, T(b["shift_c"]) if adj else None, T(b["mean_c"]) if adj else None,
                           T(b["ctr"]) if tile_ctr else None, st)
                if zd is None and first_of_two:
                    self._record(pl["tower_ev"][i])

            if i == 0 and self.fuse_block1:
                w1 = _p(self.view("conv1.kernel"))
                if training:
                    # fold: the extreme goes out as a padded activation tensor (mode 2) and block 2 reads it as it is
                    self._call("vm_conv1_fused_fwd", W(pl["x0"]), w1, bias, gam, None, nw, L, c, pool, 2 if fold else 0, dt,
                               W(b["ep"] if fold else b["e"]), ssum, ssq, st)
                    if first_of_two and self.tower_stagger == 1:
                        if "stagger_ev" not in pl:
                            pl["stagger_ev"] = torch.cuda.Event()
                        self._record(pl["stagger_ev"])
                    finalize()
                    if not fold:
                        self._call("vm_bn_drop_pool_fwd", W(b["e"]), T(b["scale"]), T(b["shift"]), dm, nw, wpt, pl["L"][1], c, 1, dt,
                                   W(b["act"]), st)
                    if first_of_two and self.tower_stagger == 2:
                        if "stagger_ev" not in pl:
                            pl["stagger_ev"] = torch.cuda.Event()
                        self._record(pl["stagger_ev"])
                else:
                    self._call("vm_bn_infer_affine", gam, bet, mm, mv, self.bn_eps, c, _p(b["scale"]), _p(b["shift"]), st)
                    self._call("vm_conv1_fused_fwd", W(pl["x0"]), w1, bias, _p(b["scale"]), _p(b["shift"]), nw, L, c, pool, 1, dt,
                               W(b["act"]), None, None, st)
                continue
            if i == 0:
                self._call("vm_conv1_fwd", W(pl["x0"]), _p(self.view("conv1.kernel")), bias, nw, L, c, dt, W(b["z"]), ssum, ssq, st)
            else:
                cin = self.blocks[i - 1][1]
                if (not training and self.fused_infer_pool and pool == 2
                        and self.lib.query("vm_conv_fwd_pool_supported", nw, L, cin, c, dt)):
                    # inference: conv + ReLU + BatchNorm affine + max-pool in one launch, z is never written (bit-identical to the
                    # two-kernel path below); the last block's GlobalMaxPool1D then runs on its pooled tensor
                    self._call("vm_bn_infer_affine", gam, bet, mm, mv, self.bn_eps, c, _p(b["scale"]), _p(b["shift"]), st)
                    self._call("vm_conv_fwd_pool", W(pl[i - 1]["act"]), _p(self.wf[i]), bias, _p(b["scale"]), _p(b["shift"]), nw, L, cin,
                               c, dt, W(b["act"]), _p(self.wfp.get(i)) if self.packed_weights else None, st)
                    continue
                b["e_now"] = b["pairs_now"] = b["ctr_now"] = False
                if fold:
                    # the BatchNorm affine of the block below (this tower's) goes into this block's weights, the conv reads that
                    # block's pool extreme and leaves its own; no pass in between
                    lo = pl[i - 1]
                    wfo, hbo, wfp = self._fold_bufs(i)
                    # (round 6) the LAST block in pair form too: its (e, o) feed the GlobalMaxPool1D pass (e alone: half of z's bytes on
                    # the forward -> backward turn-around) and the backward's sparse sums / apply pass
                    last_pairs = (i == self.nb - 1 and self.last_pairs and self.fold_pairs and pool == 2 and L % 2 == 0 and c % 8 == 0
                                  and bool(pl.get("tail_parts")) and self.fused_sums_finalize and not self.sync_bn and "o" in b)
                    with_e = i < self.nb - 1 or last_pairs
                    use_packed = wfp is not None and self.packed_weights
                    # (a block whose input extreme is stored CENTRED -- block 1's always, a centred tile's below -- folds the shift over
                    # the stored value)
                    lo_c = (i == 1 and self.fuse_block1) or bool(lo.get("ctr_now"))
                    pairs = with_e and self.fold_pairs
                    # f16: this block's own tile centred on the pedestal its accumulators start from (round 5)
                    ctr = T(b["ctr"]) if (pairs and dt == VM_F16 and i in self.center_blocks) else None
                    self._call("vm_fold_bn_weights", _p(self.wt[i]), T(lo["scale"]), T(lo["shift_c" if lo_c else "shift"]),
                               bias, ntw, cin, c, dt,
                               wfo[tw0].data_ptr(), wfp[tw0].data_ptr() if use_packed else None, hbo[tw0].data_ptr(), ctr, st)
                    self._call("vm_conv_fwd_fold", W(lo["ep"]), wfo[tw0].data_ptr(), bias, hbo[tw0].data_ptr(), gam if with_e else None,
                               nw, wpt, L, cin, c, dt, None if pairs else W(b["z"]), ssum, ssq, W(b["ep"]) if with_e else None,
                               W(b["o"]) if pairs else None, wfp[tw0].data_ptr() if use_packed else None, ctr, st)
                    b["e_now"], b["pairs_now"], b["ctr_now"] = with_e, pairs, ctr is not None
                    if last_pairs:
                        finalize()
                        parts, seg = pl["gmax_ws"], self._seg_rows
                        self._call("vm_bn_drop_pool_gmax_partials_e", W(b["ep"]), T(b["scale"]), T(b["shift"]), dm, nw, wpt, L // 2, c, dt,
                                   parts.data_ptr() + w0 * seg * c * 4, parts.data_ptr() + (pl["n"] + w0) * seg * c * 4, st)
                        fused_tail = True
                        continue
                    if with_e:
                        finalize()
                        continue
                elif (training and self.fused_pool_extreme and pool == 2 and i < self.nb - 1
                        and self.lib.query("vm_conv_fwd_e_supported", nw, L, cin, c, dt)):
                    # the conv epilogue also leaves the pool-window extreme of z (chosen by sign(gamma)): the BatchNorm / pool pass
                    # then reads that pooled-size tensor instead of z -- same bits -- and the backward sums are taken against it
                    if "e" not in b:
                        b["e"] = torch.empty(pl["n"], L // 2, c, dtype=self.tdt, device=self.device)
                    self._call("vm_conv_fwd_e", W(pl[i - 1]["act"]), _p(self.wf[i]), bias, gam, nw, L, cin, c, dt, W(b["z"]), ssum, ssq,
                               W(b["e"]), st)
                    finalize()
                    self._call("vm_bn_drop_pool_fwd", W(b["e"]), T(b["scale"]), T(b["shift"]), dm, nw, wpt, L // 2, c, 1, dt,
                               W(b["act"]), st)
                    b["e_now"] = True
                    continue
                else:
                    self._call("vm_conv_fwd", W(pl[i - 1]["act"]), _p(self.wf[i]), bias, nw, L, cin, c, dt, W(b["z"]), ssum, ssq, st)
            if training:
                finalize()
            else:
                self._call("vm_bn_infer_affine", gam, bet, mm, mv, self.bn_eps, c, _p(b["scale"]), _p(b["shift"]), st)
            sc, sh = (T(b["scale"]), T(b["shift"])) if training else (_p(b["scale"]), _p(b["shift"]))
            if i == self.nb - 1:
                # last block: BN apply + dropout + max-pool + GlobalMaxPool1D in one pass; its pooled tensor has no other
                # consumer and is never written
                if pl.get("tail_parts"):
                    # the fused tail finishes the segment partials inside its own launch: every range (tower) writes the rows of its
                    # windows into the one pair of arrays (value rows, then position rows)
                    parts, seg = pl["gmax_ws"], self._seg_rows
                    self._call("vm_bn_drop_pool_gmax_partials", W(b["z"]), sc, sh, dm, nw, wpt, L, c, pool, dt,
                               parts.data_ptr() + w0 * seg * c * 4, parts.data_ptr() + (pl["n"] + w0) * seg * c * 4, st)
                else:
                    self._call("vm_bn_drop_pool_gmax_fwd", W(b["z"]), sc, sh, dm, nw, wpt, L, c, pool, dt, W(pl["gmax"]), W(pl["gidx"]),
                               _p(gmax_ws), st)
                fused_tail = True
                continue
            self._call("vm_bn_drop_pool_fwd", W(b["z"]), sc, sh, dm, nw, wpt, L, c, pool, dt, W(b["act"]), st)
        if not fused_tail:
            pl["tail_parts"] = False   # (both ranges of a split forward take this branch or neither)
            cl, Ll = self.blocks[-1][1], pl["L"][-1]
            self._call("vm_global_maxpool_fwd", W(pl[self.nb - 1]["act"]), nw, Ll, cl, dt, W(pl["gmax"]), W(pl["gidx"]), st)

    def backward(self, pl: dict, sync_tail: bool = False):
        """pl['demb'] -> gradients of every encoder tensor in self.G (fixed summation order throughout).  ``sync_tail``
        (data parallelism, set by the train steps that go on to ``optimizer_step``): as soon as the gradients of blocks 2..n,
        the dense layer and the head are enqueued, hand G[conv2.kernel:] to ``grad_sync.begin_tail`` so that its all-reduce
        overlaps the block-2 dgrad and the whole block-1 backward (voicemap_amd/parallel.py)."""
        assert pl["training"]
        sync_tail = sync_tail and self.grad_sync is not None and hasattr(self.grad_sync, "begin_tail")
        lib, st, n, dt, wpt = self.lib, self.stream(), pl["n"], self.dtype, pl["wpt"]
        drop = pl["drop"]
        fold = bool(pl.get("fold_now"))   # the forward ran blocks 2.. on pool extremes with folded BatchNorm affines
        cl, Ll = self.blocks[-1][1], pl["L"][-1]
        G = self.G
        dense = (_p(pl["gmax"]), _p(self.view("dense.kernel")), _p(pl["demb"]), n, cl, self.E)
        assert not pl.get("tail_pending"), "forward(defer_tail=True): call siamese_head(pl, y) before backward"
        if pl.get("tail_grads_pending"):
            # vm_tail_fwd_bwd has left dgmax; what only the optimizer reads (loss, accuracy, the head's and the dense layer's parameter
            # gradients) is one launch, on the side stream where there is one
            pl["tail_grads_pending"] = False
            tail_args = (_p(pl["gmax"]), _p(pl["demb"]), _p(pl["emb"]), _p(pl["head_ws"]), n // 2, cl, self.E, HEADS[self.head],
                         _p(pl["loss_acc"]), _p(self.view("dense.kernel", G)), _p(self.view("dense.bias", G)),
                         _p(self.view("head.kernel", G)), _p(self.view("head.bias", G)))
            if self.overlap_wgrad:
                if "head_ev" not in pl:
                    pl["head_ev"] = torch.cuda.Event()
                self._record(pl["head_ev"])
                with self._on(self.side_stream):
                    self._wait(self.side_stream, pl["head_ev"])
                    self._call("vm_tail_param_grads", *tail_args, self.stream())
            else:
                self._call("vm_tail_param_grads", *tail_args, st)
        elif self.overlap_wgrad:
            # what only the optimizer reads -- the head's sums, the dense layer's parameter gradients -- goes to the side stream: the
            # main stream's chain to the first BatchNorm backward is four small launches shorter
            if "head_ev" not in pl:
                pl["head_ev"] = torch.cuda.Event()
            self._record(pl["head_ev"])
            with self._on(self.side_stream):
                self._wait(self.side_stream, pl["head_ev"])
                if pl.get("head_pending"):
                    self._call("vm_siamese_head_reduce", _p(pl["emb"]), _p(pl["head_ws"]), n // 2, self.E, HEADS[self.head], _p(pl["loss_acc"]),
                               _p(self.view("head.kernel", G)), _p(self.view("head.bias", G)), self.stream())
                    pl["head_pending"] = False
                self._call("vm_dense_bwd", *dense, _p(self.view("dense.kernel", G)), _p(self.view("dense.bias", G)), None, self.stream())
            self._call("vm_dense_bwd", *dense, None, None, _p(pl["dgmax"]), st)
        else:
            if pl.get("head_pending"):   # (the switch was flipped between the head and here)
                self._call("vm_siamese_head_reduce", _p(pl["emb"]), _p(pl["head_ws"]), n // 2, self.E, HEADS[self.head], _p(pl["loss_acc"]),
                           _p(self.view("head.kernel", G)), _p(self.view("head.bias", G)), st)
                pl["head_pending"] = False
            self._call("vm_dense_bwd", *dense, _p(self.view("dense.kernel", G)), _p(self.view("dense.bias", G)), _p(pl["dgmax"]), st)
        # GlobalMaxPool1D backward stays sparse (dgmax, gidx): the last block's BN-backward passes consume that form
        last = self.nb - 1
        wgrad_tails = []   # side stream: the small launches behind every weight-gradient GEMM, deferred behind the last one
        for i in range(self.nb - 1, -1, -1):
            k, c, pool = self.blocks[i]
            b, L = pl[i], pl["L"][i]
            dm = self._dyn(("dropb", i), _p(drop[i])) if drop is not None and drop[i] is not None else None
            if i == 0 and self.fuse_block1:
                Lq = pl["L"][1]
                if b.get("bnred_now") and self.fused_sums_finalize and not self.sync_bn:
                    # the sums of the dgrad epilogue straight into the column reduction: two small launches instead of three
                    self._call("vm_bn_bwd_from_sums_finalize", _p(b["rs0"]), _p(b["rs1"]), b["rs_rows"], None, _p(b["dp"]), _p(b["scale"]),
                               _p(b["shift"]), _p(b["mean_c" if fold else "mean"]), _p(b["invstd"]), dm, n, wpt, Lq, c, 1, dt, 0, float(wpt * L), _p(b["c1"]),
                               _p(b["c2"]), _p(self.view("bn1.gamma", G)), _p(self.view("bn1.beta", G)), _p(pl["cr_ws"]), st)
                else:
                    if b.get("bnred_now"):
                        self._call("vm_bn_bwd_from_sums", _p(b["rs0"]), _p(b["rs1"]), b["rs_rows"], None, _p(b["dp"]), _p(b["scale"]),
                                   _p(b["shift"]), _p(b["mean_c" if fold else "mean"]), _p(b["invstd"]), dm, n, wpt, Lq, c, 1, dt, 0, _p(b["pa"]),
                                   _p(b["pb"]), st)
                    else:
                        self._call("vm_bn_pool_bwd_reduce", _p(b["e"]), _p(b["dp"]), _p(b["scale"]), _p(b["shift"]), _p(b["mean"]),
                                   _p(b["invstd"]), dm, n, wpt, Lq, c, 1, dt, _p(b["pa"]), _p(b["pb"]), st)
                    self._bn_bwd_finalize(("bwd", 0), _p(b["pa"]), _p(b["pb"]), n, wpt, c, float(wpt * L), _p(b["c1"]), _p(b["c2"]),
                                          _p(self.view("bn1.gamma", G)), _p(self.view("bn1.beta", G)), _p(pl["cr_ws"]), st)
                self._call("vm_conv1_fused_bwd", _p(pl["x0"]), _p(self.view("conv1.kernel")), _p(self.view("conv1.bias")),
                           _p(b["dp"]), _p(b["scale"]), _p(b["mean"]), _p(b["invstd"]), dm, _p(b["c1"]), _p(b["c2"]), n, wpt, L,
                           c, pool, dt, _p(pl["wgrad_ws"]), _p(self.view("conv1.kernel", G)), _p(self.view("conv1.bias", G)), st)
                if "x0_free_ev" in pl:
                    self._record(pl["x0_free_ev"])   # x0's last reader of this step is enqueued: the next step's preprocessing may overwrite it
                continue
            sparse = (i == last and last > 0)
            if sparse:
                common = (_p(b["z"]), _p(pl["dgmax"]), _p(pl["gidx"]), _p(b["scale"]), _p(b["shift"]), _p(b["mean"]),
                          _p(b["invstd"]), dm)
            else:
                common = (_p(b["z"]), _p(b["dp"]), _p(b["scale"]), _p(b["shift"]), _p(b["mean"]), _p(b["invstd"]), dm)
            fused_fin = False
            if sparse and b.get("pairs_now"):
                # the last block left (e, o): the sparse sums gather e, the apply pass takes the pair form
                self._call("vm_bn_bwd_gmax_finalize_e", _p(b["ep"]), *common[1:], n, wpt, L // 2, c, dt, float(wpt * L), _p(b["c1"]), _p(b["c2"]),
                           _p(self.view(f"bn{i+1}.gamma", G)), _p(self.view(f"bn{i+1}.beta", G)), st)
                fused_fin = True
            elif sparse and self.fused_sums_finalize and not self.sync_bn and c % 8 == 0:
                # the gather of the sparse sums and their finalize in one launch (three dependent small launches of the forward ->
                # backward turn-around otherwise)
                self._call("vm_bn_bwd_gmax_finalize", *common, n, wpt, L, c, pool, dt, float(wpt * L), _p(b["c1"]), _p(b["c2"]),
                           _p(self.view(f"bn{i+1}.gamma", G)), _p(self.view(f"bn{i+1}.beta", G)), st)
                fused_fin = True
            elif sparse:
                self._call("vm_bn_pool_bwd_reduce_gmax", *common, n, wpt, L, c, pool, dt, _p(b["pa"]), _p(b["pb"]), st)
            elif b.get("bnred_now") and self.fused_sums_finalize and not self.sync_bn:
                self._call("vm_bn_bwd_from_sums_finalize", _p(b["rs0"]), _p(b["rs1"]), b["rs_rows"], None if b.get("pairs_now") else _p(b["z"]),
                           _p(b["dp"]), _p(b["scale"]),
                           _p(b["shift"]), _p(b["mean_c" if b.get("ctr_now") else "mean"]), _p(b["invstd"]), dm, n, wpt, L, c, pool, dt, 0 if b.get("e_now") else 1,
                           float(wpt * L), _p(b["c1"]), _p(b["c2"]), _p(self.view(f"bn{i+1}.gamma", G)), _p(self.view(f"bn{i+1}.beta", G)),
                           _p(pl["cr_ws"]), st)
                fused_fin = True
            elif b.get("bnred_now"):
                self._call("vm_bn_bwd_from_sums", _p(b["rs0"]), _p(b["rs1"]), b["rs_rows"], _p(b["z"]), _p(b["dp"]), _p(b["scale"]),
                           _p(b["shift"]), _p(b["mean_c" if b.get("ctr_now") else "mean"]), _p(b["invstd"]), dm, n, wpt, L, c, pool, dt, 0 if b.get("e_now") else 1,
                           _p(b["pa"]), _p(b["pb"]), st)
            elif self.pooled_reduce and L % pool == 0:
                # throughput mode: the pool-window extreme comes from this block's pooled output (= the next block's input),
                # so the pass reads two pooled-size tensors instead of z + dp
                self._call("vm_bn_pool_bwd_reduce_pooled", _p(b["z"]), _p(b["act"]), _p(b["dp"]), _p(b["scale"]), _p(b["shift"]),
                           _p(b["mean"]), _p(b["invstd"]), dm, n, wpt, L, c, pool, dt, _p(b["pa"]), _p(b["pb"]), st)
            else:
                self._call("vm_bn_pool_bwd_reduce", *common, n, wpt, L, c, pool, dt, _p(b["pa"]), _p(b["pb"]), st)
            if not fused_fin:
                self._bn_bwd_finalize(("bwd", i), _p(b["pa"]), _p(b["pb"]), n, wpt, c, float(wpt * L), _p(b["c1"]), _p(b["c2"]),
                                      _p(self.view(f"bn{i+1}.gamma", G)), _p(self.view(f"bn{i+1}.beta", G)), _p(pl["cr_ws"]), st)
            if b.get("pairs_now") and sparse:
                self._call("vm_bn_pool_bwd_apply_pairs_gmax", _p(b["ep"]), _p(b["o"]), *common[1:], _p(b["c1"]), _p(b["c2"]), n, wpt, L, c, dt,
                           _p(b["du"]), _p(b["pdu"]), st)
            elif b.get("pairs_now") and not sparse:
                self._call("vm_bn_pool_bwd_apply_pairs", _p(b["ep"]), _p(b["o"]), *common[1:], _p(b["c1"]), _p(b["c2"]), n, wpt, L, c, dt,
                           _p(b["du"]), _p(b["pdu"]), _p(b["ctr"]) if b.get("ctr_now") else None, st)
            else:
                self._call("vm_bn_pool_bwd_apply_gmax" if sparse else "vm_bn_pool_bwd_apply", *common, _p(b["c1"]), _p(b["c2"]), n,
                           wpt, L, c, pool, dt, _p(b["du"]), _p(b["pdu"]), st)
            side = self.overlap_wgrad and i > 0
            gb = _p(self.view(f"conv{i+1}.bias", G))
            gw = _p(self.view(f"conv{i+1}.kernel", G))

            def wgrad(stream, cr_ws, i=i, b=b, L=L, c=c, gb=gb, gw=gw):
                # the conv bias gradient = column sums of du (from the apply pass's partials), then the weight gradient
                cin = self.blocks[i - 1][1]
                if fold:
                    lo = pl[i - 1]
                    # the GEMM first (it needs e and du only), the tap sums behind it, then the per-tower slab sums with their factors
                    self._call("vm_conv_wgrad_fold", _p(lo["ep"]), _p(b["du"]), n, wpt, L, cin, c, dt, None, None, None,
                               _p(b["wgrad_ws_fold"]), None, stream)

                    def finish(st_):
                        self._call("vm_du_tower_sums", _p(b["pdu"]), _p(b["du"]), n, wpt, L, c, dt, gb, _p(b["dsum"]), _p(cr_ws), st_)
                        self._call("vm_conv_wgrad_fold_finish", _p(b["wgrad_ws_fold"]), n, wpt, L, cin, c, _p(lo["scale"]),
                                   _p(lo["shift_c" if ((i == 1 and self.fuse_block1) or lo.get("ctr_now")) else "shift"]), _p(b["dsum"]), gw, st_)
                    # (round 6 experiments, profiles/r06_wgrad_tail.txt) on the side stream these small launches sit BETWEEN two
                    # weight-gradient GEMMs; beside the memory-bound apply pass of the main stream their 1024-thread workgroups find
                    # no compute unit until that pass has drained (85-130 us for a 5 us reduction).  defer_wgrad_tail 1: all of them
                    # behind the LAST weight-gradient GEMM; 2: on a stream of their own behind their GEMM.
                    on_side = stream == self.side_stream.cuda_stream
                    if self.defer_wgrad_tail == 1 and on_side:
                        wgrad_tails.append(lambda: finish(stream))
                    elif self.defer_wgrad_tail == 2 and on_side:
                        self._join(self.misc_stream, self.side_stream)
                        with self._on(self.misc_stream):
                            finish(self.stream())
                    else:
                        finish(stream)
                else:
                    self._call("vm_conv_wgrad", _p(pl[i - 1]["act"]), _p(b["du"]), n, L, cin, c, dt, _p(b["wgrad_ws"]), gw, stream)
                    self._call("vm_colsum", _p(b["pdu"]), b["pdu"].shape[0], c, gb, _p(cr_ws), stream)

            if i == 0:
                self._call("vm_colsum", _p(b["pdu"]), b["pdu"].shape[0], c, gb, _p(pl["cr_ws"]), st)
                self._call("vm_conv1_wgrad", _p(pl["x0"]), _p(b["du"]), n, L, c, dt, _p(pl["wgrad_ws"]), gw, st)
                if "x0_free_ev" in pl:
                    self._record(pl["x0_free_ev"])
            else:
                cin = self.blocks[i - 1][1]

                def side_wgrad():
                    self._record(b["ev"])
                    with self._on(self.side_stream):
                        self._wait(self.side_stream, b["ev"])
                        # (the bias gradient is nobody's input until the optimizer: off the main stream, with its own workspace)
                        wgrad(self.stream(), pl["cr_ws_side"])
                        if i == 1:   # the last weight-gradient GEMM of the step is enqueued: now the deferred sums / slab folds
                            for fin in wgrad_tails:
                                fin()
                            del wgrad_tails[:]
                            if self.defer_wgrad_tail == 2:   # what waits for the side stream from here on waits for the small launches too
                                self._join(self.side_stream, self.misc_stream)
                    if i == 1 and sync_tail:
                        self._begin_grad_tail(pl)

                late = self.overlap_wgrad and self.wgrad_after_dgrad and not (i == 1 and sync_tail)
                if self.overlap_wgrad and not late:
                    side_wgrad()
                elif not self.overlap_wgrad:
                    wgrad(st, pl["cr_ws"])
                    if i == 1 and sync_tail:
                        self._begin_grad_tail(pl)
                lo = pl[i - 1]
                lo["bnred_now"] = self._bnred_plan(pl, i)
                if lo["bnred_now"]:
                    use_e = (i == 1 and self.fuse_block1) or bool(lo.get("e_now"))   # the extreme itself, else the pooled output
                    red_a, padded = (lo["ep"], 1) if fold else ((lo["e"], 0) if use_e else (lo["act"], 1))
                    self._call("vm_conv_dgrad_bnred", _p(b["du"]), _p(self.wd[i]), n, L, cin, c, dt, _p(lo["dp"]), _p(red_a), padded,
                               _p(lo["rs0"]), _p(lo["rs1"]), _p(self.wdp.get(i)) if self.packed_weights else None, st)
                else:
                    self._call("vm_conv_dgrad", _p(b["du"]), _p(self.wd[i]), n, L, cin, c, dt, _p(lo["dp"]), st)
                if late:
                    side_wgrad()   # experiment: the weight-gradient GEMM beside the memory-bound passes of the block below
        if self.overlap_wgrad:
            self._join(torch.cuda.current_stream(self.device), self.side_stream)

    def _bnred_plan(self, pl: dict, i: int) -> bool:
        """Does block i's dgrad also reduce block i-1's BatchNorm-backward sums?  Asked per call (the answer follows
        vm_set_tuning); allocates the partial rows on first use."""
        if not self.fused_bn_reduce or i < 1:
            return False
        lo, n, L = pl[i - 1], pl["n"], pl["L"][i]
        cin, c = self.blocks[i - 1][1], self.blocks[i][1]
        if not (i == 1 and self.fuse_block1) and not lo.get("e_now"):
            if not (self.pooled_reduce and pl["L"][i - 1] % self.blocks[i - 1][2] == 0):
                return False
        if not self.lib.query("vm_conv_dgrad_bnred_supported", n, L, cin, c, self.dtype):
            return False
        if "rs0" not in lo:
            lo["rs_rows"] = self.lib.query("vm_conv_dgrad_bnred_rows", L)
            for nm in ("rs0", "rs1"):
                lo[nm] = torch.empty(n * lo["rs_rows"], cin, dtype=torch.float32, device=self.device)
        return True

    # ------------------------------------------------------------------------------------------------
    def siamese_head(self, pl: dict, y: Optional[torch.Tensor], loss: str = "contrastive"):
        """Twin distance -> Dense(1, sigmoid) -> loss (+ backward into pl['demb'] and the head gradients)."""
        assert self.head in HEADS, "engine was built without a siamese head"
        pairs = pl["n"] // 2
        G = self.G
        train = y is not None
        # training with the side stream on: only the per-pair pass here; the fixed-order sums (loss, accuracy, head gradients --
        # nobody's input before the optimizer) are enqueued on the side stream by backward()
        if pl.get("tail_pending"):
            assert train, "forward(defer_tail=True) must be followed by siamese_head with labels"
            cl, parts, seg = self.blocks[-1][1], pl["gmax_ws"], self._seg_rows
            pv = parts.data_ptr() if pl["tail_parts"] else None
            pi = parts.data_ptr() + pl["n"] * seg * cl * 4 if pl["tail_parts"] else None
            self._call("vm_tail_fwd_bwd", pv, pi, seg, _p(pl["gmax"]), _p(pl["gidx"]), _p(self.view("dense.kernel")), _p(self.view("dense.bias")),
                       _p(self.view("head.kernel")), _p(self.view("head.bias")), self._dyn("y", _p(y)), pairs, cl, self.E, HEADS[self.head],
                       LOSSES[loss], self._dyn("loss_scale", float(self.loss_scale)), _p(pl["emb"]), _p(pl["pred"]), _p(pl["demb"]), _p(pl["dgmax"]), _p(pl["head_ws"]), self.stream())
            pl["tail_pending"], pl["tail_grads_pending"], pl["head_pending"] = False, True, False
            return pl["pred"][:pairs]
        defer = train and self.overlap_wgrad and pl["training"] and getattr(self, "defer_head_reduce", False)
        self._call("vm_siamese_head_loss", _p(pl["emb"]), _p(self.view("head.kernel")), _p(self.view("head.bias")),
                      self._dyn("y", _p(y)), pairs, self.E, HEADS[self.head], LOSSES[loss], self._dyn("loss_scale", float(self.loss_scale)),
                      _p(pl["pred"]),
                      _p(pl["loss_acc"]) if (train and not defer) else None, _p(pl["demb"]) if train else None,
                      _p(self.view("head.kernel", G)) if train else None, _p(self.view("head.bias", G)) if train else None,
                      _p(pl["head_ws"]), self.stream())
        pl["head_pending"] = defer
        return pl["pred"][:pairs]

    def classifier_head(self, pl: dict, labels: Optional[torch.Tensor]):
        """Dense(num_classes, softmax) + categorical CE (+ backward into pl['demb'] and the head gradients)."""
        assert self.head == "classifier"
        lib, st, n = self.lib, self.stream(), pl["n"]
        self._call("vm_dense_fwd", _p(pl["emb"]), _p(self.view("head.kernel")), _p(self.view("head.bias")), n, self.E,
                 self.num_classes, _p(pl["logits"]), st)
        train = labels is not None
        self._call("vm_softmax_cce", _p(pl["logits"]), self._dyn("y", _p(labels)), n, self.num_classes,
                 self._dyn("loss_scale", float(self.loss_scale)), _p(pl["prob"]),
                 _p(pl["loss_acc"]) if train else None, _p(pl["dlogits"]) if train else None, _p(pl["cce_ws"]), st)
        if train:
            G = self.G
            self._call("vm_dense_bwd", _p(pl["emb"]), _p(self.view("head.kernel")), _p(pl["dlogits"]), n, self.E,
                     self.num_classes, _p(self.view("head.kernel", G)), _p(self.view("head.bias", G)), _p(pl["demb"]), st)
        return pl["prob"]

    def _adam_scalars(self):
        """(t, lr_t) of the Adam step about to run (Keras: lr decays with the iteration count BEFORE this step)."""
        lr = self.lr
        if self.decay > 0:
            lr = lr * (1.0 / (1.0 + self.decay * self.iterations))
        t = self.iterations + 1
        return t, lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def optimizer_step(self):
        """Keras Adam(clipnorm) on the flat buffers (after the optional data-parallel gradient sum)."""
        if self.grad_sync is not None:
            gs, G = self.grad_sync, self.G
            self._host_call(lambda: gs(G))
        lib, st = self.lib, self.stream()
        skip = self.loss_scaled   # loss-scaled storage: a non-finite gradient norm skips the update on the device
        if (self.clipnorm and self.clipnorm > 0) or skip:
            self._call("vm_grad_sqnorm", _p(self.G), self.n_flat, _p(self._sq_ws), None, st)   # partials; the optimizer kernel adds them
        t, lr_t = self._adam_scalars()
        self._call("vm_adam_clip_step", _p(self.P), _p(self.G), _p(self.M), _p(self.V), self.n_flat, self._dyn("lr_t", float(lr_t)), self.beta_1,
                 self.beta_2, self.adam_eps, float(self.clipnorm or 0.0),
                 self._dyn("gpre", float(self.grad_prescale) / float(self.loss_scale)),
                 _p(self._sqnorm), _p(self._sq_ws), int(skip), _p(self._skipped) if skip else None, st)
        self.iterations = t
        if skip and self._rec is None:
            self._poll_loss_scale()   # after the launch: the copy it may enqueue sees this step's count
        self.refresh_weights()
        if skip and self._rec is not None:
            self._poll_loss_scale()   # a step being recorded: torch's copies stay out of the recorded list and go behind it, as in a replay

    # ------------------------------------------------------------------------------------------------
    def make_drop_masks(self, n_windows: int, generator: Optional[torch.Generator] = None):
        """SpatialDropout1D keep masks (n_windows, C) / (1 - rate) per block; None when rate == 0.  With the engine's own generator
        (the training loops) the masks of all blocks are slices of ONE persistent buffer filled by three launches -- one uniform draw,
        one compare, one scale -- instead of sixteen, at fixed addresses; they are valid until the next call.  With a caller's
        generator: fresh tensors, one draw per block (tests hand them to the oracle as well)."""
        if self.dropout <= 0.0:
            return None
        if generator is None:
            bufs = self._drop_bufs.get(n_windows)
            if bufs is None:
                tot = sum(n_windows * c for (_, c, _) in self.blocks)
                u = torch.empty(tot, dtype=torch.float32, device=self.device)
                m = torch.empty_like(u)
                views, o = [], 0
                for (_, c, _) in self.blocks:
                    views.append(m[o:o + n_windows * c].view(n_windows, c))
                    o += n_windows * c
                bufs = self._drop_bufs[n_windows] = (u, torch.empty(tot, dtype=torch.bool, device=self.device), m, views)
            u, keep, m, views = bufs
            u.uniform_(generator=self._drop_gen)
            torch.ge(u, self.dropout, out=keep)
            torch.mul(keep, 1.0 / (1.0 - self.dropout), out=m)
            return views
        out = []
        for (_, c, _) in self.blocks:
            u = torch.rand(n_windows, c, device=self.device, generator=generator)
            out.append((u >= self.dropout).to(torch.float32) / (1.0 - self.dropout))
        return out

    # ---- one training step, eager or replayed -----------------------------------------------------------------------------------
    def _step_flags(self):
        return (self.fold_affine, self.fuse_block1, self.fused_bn_reduce, self.fused_sums_finalize, self.split_towers, self.tower_stagger, self.tower_swap, self.last_pairs,
                self.overlap_wgrad, self.wgrad_after_dgrad, self.defer_wgrad_tail, self.fused_pool_extreme, self.fold_pairs, self.pooled_reduce,
                self.packed_weights, self.fused_tail, self.defer_head_reduce, self.unbiased, self.clipnorm, self.loss_scaled,
                self.bn_zero_debias, self.bn_eps, self.bn_momentum, self.beta_1, self.beta_2, self.adam_eps, self._side_priority,
                self.grad_prescale, self.fused_infer_pool, self.center_blocks, self.pre_overlap, id(self.grad_sync), self.lib.tuning_epoch, id(self.side_stream), id(self.tower_stream), id(self.misc_stream))

    def _train_step(self, pl: dict, wpt: int, target: torch.Tensor, loss: Optional[str], drop_masks, apply_update: bool, pre,
                    input_ready: bool = False):
        """preprocess (``pre`` = None | ("raw", tensor, downsampling, whitening) | ("offsets", audio, offsets, raw_len, downsampling,
        whitening)) -> forward -> head (``loss`` None: the classifier's) -> backward -> optimizer.  The SECOND time a configuration
        is seen its enqueue sequence is recorded (_Program), from the third on it is replayed: same launches, same arguments, same
        stream ordering -- only the input / label / mask pointers and four scalars are patched in."""
        # (a gradient hook that declares itself ``replayable`` -- parallel.GradAllReduce: its two collectives are host calls kept in
        # the program, its stream ordering goes through the engine -- no longer forces the eager path; SyncBN still does)
        usable = (self.replay and (self.grad_sync is None or getattr(self.grad_sync, "replayable", False)) and not self.timed
                  and not self.sync_bn)
        prog = key = None
        if usable:
            # (the offsets path never reads the audio buffer's length: its shape is not part of what a program depends on)
            sig = None if pre is None else ((pre[0], pre[1].dtype, tuple(pre[1].shape)) + tuple(pre[-2:]) if pre[0] == "raw"
                                            else (pre[0], pre[1].dtype, pre[3]) + tuple(pre[-2:]))
            masks = None if drop_masks is None else tuple(m is not None for m in drop_masks)
            key = (id(pl), wpt, loss, apply_update, self.stream(), sig, masks, bool(input_ready), self._step_flags())
            prog = self._programs.get(key)
            if isinstance(prog, _Program):
                self._programs[key] = self._programs.pop(key)   # most recently used last
                self._x0_handover(pl, prog)
                self._replay_step(prog, pl, wpt, target, drop_masks, apply_update, pre)
                return
            if prog is None:
                self._programs[key] = 1            # first sighting: run it (lazy buffers get allocated), record the next one
                while len(self._programs) > 64:    # bounded: the least recently used configuration goes (with its events)
                    old = next(iter(self._programs))
                    if isinstance(self._programs[old], _Program):
                        torch.cuda.synchronize(self.device)
                    self._destroy_program(self._programs.pop(old))
            else:
                self._rec = _Program()
        self._x0_handover(pl, "eager")
        try:
            if pre is not None:
                ahead = bool(input_ready and self.pre_overlap and not self.timed and pl["training"])
                if ahead:
                    # on the tower stream, behind the previous step's last reader of x0 only (x0_free_ev: recorded by backward())
                    cur = torch.cuda.current_stream(self.device)
                    if "x0_free_ev" not in pl:
                        pl["x0_free_ev"] = torch.cuda.Event()
                    pre_cm = self._on(self.tower_stream)
                    pre_cm.__enter__()
                    self._wait(self.tower_stream, pl["x0_free_ev"])
                try:
                    if pre[0] == "raw":
                        self.preprocess(pl, pre[1], pre[2], pre[3], wpt)
                    else:
                        self.preprocess(pl, pre[1], pre[4], pre[5], wpt, offsets=pre[2], raw_len=pre[3])
                finally:
                    if ahead:
                        pre_cm.__exit__(None, None, None)
                if ahead:
                    self._join(cur, self.tower_stream)
            if loss is None:
                self.forward(pl, wpt, drop_masks)
                self.classifier_head(pl, target)
            else:
                self.forward(pl, wpt, drop_masks, defer_tail=True)
                self.siamese_head(pl, target, loss)
            self.backward(pl, sync_tail=apply_update)
            if apply_update:
                self.optimizer_step()
        finally:
            rec, self._rec = self._rec, None
        if rec is not None:
            self._programs[key] = self._finish_program(rec)

    def _x0_handover(self, pl: dict, who):
        """x0_free_ev is a torch event in an eager step and the program's own event in a replayed one: when the step before this one
        on the same plan was enqueued by somebody else (eager <-> replay, another program) the wait inside this step would see an event
        that step never recorded -- the tower stream then simply waits for everything the main stream holds (no overlap, once)."""
        if pl.get("x0_owner") is not who:
            if pl.get("x0_owner") is not None:
                self.tower_stream.wait_stream(torch.cuda.current_stream(self.device))
            pl["x0_owner"] = who

    def train_step_resident(self, pl: dict, windows_per_tower: int, target: torch.Tensor, loss: Optional[str] = "contrastive", raw=None,
                            downsampling: int = 4, whitening: bool = True, drop_masks="auto", apply_update: bool = True,
                            input_ready: bool = False):
        """One training step on tensors that are already on the device: ``raw`` (n_windows, samples) fp32 / int16 windows (None: the
        plan's input was loaded with load_preprocessed), ``target`` the labels (fp32 (pairs,) for the siamese losses, int32
        (n_windows,) with ``loss=None`` for the classifier).  What siamese_train_step / classifier_train_step run after their
        host-to-device copies; bench.py times this.  ``input_ready``: the caller guarantees that ``raw`` is complete in memory
        whatever the current stream still holds (a resident corpus, synthetic data) -- the preprocessing then runs on the tower stream
        beside the previous step's optimizer tail (``pre_overlap``) instead of behind it."""
        if isinstance(drop_masks, str):
            drop_masks = self.make_drop_masks(pl["n"])
        self._train_step(pl, windows_per_tower, target, loss, drop_masks, apply_update,
                         None if raw is None else ("raw", raw, downsampling, whitening), input_ready=input_ready)
        return pl

    def _replay_step(self, prog: _Program, pl: dict, wpt: int, target, drop_masks, apply_update: bool, pre):
        dyn = {"y": target.data_ptr(), "loss_scale": float(self.loss_scale)}
        keep = [target]
        if pre is not None:
            raw = pre[1]
            if pre[0] == "raw":
                raw = raw.reshape(pl["n"], -1).contiguous()
                if raw.dtype != torch.int16:
                    raw = raw.to(torch.float32)
            else:
                dyn["offsets"] = pre[2].data_ptr()
                keep.append(pre[2])
            dyn["raw"] = raw.data_ptr()
            keep.append(raw)
        if drop_masks is not None:
            for i, m in enumerate(drop_masks):
                if m is not None:
                    c = m.shape[1]
                    dyn[("dropb", i)] = m.data_ptr()
                    dyn[("drop", i, 0)] = m.data_ptr()
                    dyn[("drop", i, wpt)] = m.data_ptr() + wpt * c * 4
        # the host-side counters the eager path advances inside forward() and optimizer_step()
        self.bn_steps += 1
        self._bn_t = self.bn_steps
        dyn["zc"] = 1.0 / (1.0 - self.bn_momentum ** self._bn_t) if self.bn_zero_debias else 0.0
        pl["wpt"], pl["drop"], pl["replay_keep"] = wpt, drop_masks, keep   # alive until the next step, like the eager path's references
        if apply_update:
            t, lr_t = self._adam_scalars()
            dyn["lr_t"], dyn["gpre"] = float(lr_t), float(self.grad_prescale) / float(self.loss_scale)
        self._run_program(prog, dyn)
        if apply_update:
            self._wfp_stale = True   # what refresh_weights() / _pack_weights() note on the eager path: the inference copies are old
            self.iterations = t
            if self.loss_scaled:
                self._poll_loss_scale()

    def siamese_train_step(self, x1, x2, y, loss: str = "contrastive", preprocessed: bool = True, downsampling: int = 4,
                           whitening: bool = True, drop_masks="auto", apply_update: bool = True):
        """One ``train_on_batch`` of the siamese scripts.  x1/x2: (pairs, L, 1) windows (already decimated+whitened if
        ``preprocessed``, else raw fp32/int16 16 kHz windows that are decimated+whitened on the GPU, each tower
        separately like voicemap/utils.py:59-60).  y: (pairs, 1) labels, 0 = same speaker."""
        x1 = torch.as_tensor(x1)
        x2 = torch.as_tensor(x2)
        pairs = x1.shape[0]
        x = torch.cat([x1.reshape(pairs, -1), x2.reshape(pairs, -1)], 0).to(self.device)
        l0 = x.shape[1] if preprocessed else (x.shape[1] + downsampling - 1) // downsampling
        pl = self.plan(2 * pairs, l0, True)
        if preprocessed:
            self.load_preprocessed(pl, x)
        if isinstance(drop_masks, str):
            drop_masks = self.make_drop_masks(2 * pairs)
        yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).to(self.device).contiguous()
        self._train_step(pl, pairs, yd, loss, drop_masks, apply_update, None if preprocessed else ("raw", x, downsampling, whitening))
        return pl

    def siamese_train_step_from_offsets(self, audio: torch.Tensor, offsets_1, offsets_2, y,
                                        raw_len: int, loss: str = "contrastive", downsampling: int = 4, whitening: bool = True,
                                        drop_masks="auto", apply_update: bool = True):
        """``siamese_train_step`` fed from a device-resident recording buffer: ``audio`` 1-D int16/fp32 on the device,
        ``offsets_k`` (pairs,) int64 start samples of the windows of tower k (shards.ShardedSpeechDataset chooses them the way
        LibriSpeechDataset.__getitem__ / build_verification_batch do).  Host numpy offsets with host labels (what fit_generator
        hands over) go up in one asynchronous copy (_stage_offsets_and_labels); tensors are taken as they are."""
        host = isinstance(offsets_1, np.ndarray) and isinstance(offsets_2, np.ndarray)
        pairs = int(offsets_1.size) if host else int(offsets_1.numel())
        pl = self.plan(2 * pairs, (raw_len + downsampling - 1) // downsampling, True)
        staged = ready = False
        if host and not torch.is_tensor(y):
            offs, yd, ready = self._stage_offsets_and_labels(pl, offsets_1, offsets_2, y, pairs)
            staged = True
        else:
            offs = torch.cat([torch.as_tensor(offsets_1).reshape(-1), torch.as_tensor(offsets_2).reshape(-1)]).to(self.device, torch.int64).contiguous()
            yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).to(self.device).contiguous()
        if isinstance(drop_masks, str):
            drop_masks = self.make_drop_masks(2 * pairs)
        self._train_step(pl, pairs, yd, loss, drop_masks, apply_update, ("offsets", audio, offs, raw_len, downsampling, whitening),
                         input_ready=ready)
        if staged:
            self._staged_step_enqueued(pl)
        return pl

    def _stage_offsets_and_labels(self, pl: dict, o1: "np.ndarray", o2: "np.ndarray", y, pairs: int):
        """Host arrays -> device buffers (2 * pairs int64 offsets, pairs fp32 labels) in ONE asynchronous copy from a ring of pinned
        staging buffers.  torch's ``.to(device)`` of a pageable array is a blocking copy that is ordered behind everything already
        enqueued on the stream: three of them per step made the host wait for the GPU to drain, then left the GPU idle while the host
        prepared the next step (fit_generator at 64 pairs: 1.67 ms per step against 1.45 ms of GPU work).  Round 6: the device side is
        a ring as well (a step reads ITS slot, so the next step's copy needs no ordering against this step's kernels) and, with
        ``pre_overlap``, the copy goes to the tower stream in front of the preprocessing that runs there ahead of the main stream.
        A slot is reused after 32 steps; the step that last read it has its end-of-enqueue event waited for first (long done).
        Returns (offsets, labels, input_ready)."""
        st = pl.get("h2d")
        if st is None:
            nbytes = 2 * pairs * 8 + pairs * 4
            st = pl["h2d"] = {"dev": [torch.empty(nbytes, dtype=torch.uint8, device=self.device) for _ in range(32)],
                              "pin": [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(32)],
                              "ev": [None] * 32, "k": 0}
        k = st["k"] % 32
        st["k"] += 1
        if st["ev"][k] is not None:
            st["ev"][k].synchronize()
        buf = st["pin"][k].numpy()
        buf[:pairs * 8] = np.ascontiguousarray(o1, dtype=np.int64).reshape(-1).view(np.uint8)
        buf[pairs * 8:2 * pairs * 8] = np.ascontiguousarray(o2, dtype=np.int64).reshape(-1).view(np.uint8)
        buf[2 * pairs * 8:] = np.ascontiguousarray(np.asarray(y, dtype=np.float32).reshape(pairs)).view(np.uint8)
        dev = st["dev"][k]
        ahead = bool(self.pre_overlap and not self.timed)
        if ahead:
            with torch.cuda.stream(self.tower_stream):
                dev.copy_(st["pin"][k], non_blocking=True)
        else:
            dev.copy_(st["pin"][k], non_blocking=True)
        st["cur"] = k
        return dev[:2 * pairs * 8].view(torch.int64), dev[2 * pairs * 8:].view(torch.float32), ahead

    def _staged_step_enqueued(self, pl: dict):
        """The step that reads the staging slot is enqueued: its slot may be refilled once everything enqueued so far has run."""
        st = pl["h2d"]
        k = st["cur"]
        ev = st["ev"][k] = st["ev"][k] or torch.cuda.Event()
        ev.record()

    def classifier_train_step(self, x, labels, preprocessed: bool = True, downsampling: int = 4, whitening: bool = True,
                              drop_masks="auto", apply_update: bool = True):
        """One ``train_on_batch`` of experiments/train_classifier.py (labels: int class ids)."""
        x = torch.as_tensor(x)
        n = x.shape[0]
        x = x.reshape(n, -1).to(self.device)
        l0 = x.shape[1] if preprocessed else (x.shape[1] + downsampling - 1) // downsampling
        pl = self.plan(n, l0, True)
        if preprocessed:
            self.load_preprocessed(pl, x)
        if isinstance(drop_masks, str):
            drop_masks = self.make_drop_masks(n)
        lab = torch.as_tensor(labels).reshape(n).to(self.device, torch.int32).contiguous()
        self._train_step(pl, n, lab, None, drop_masks, apply_update, None if preprocessed else ("raw", x, downsampling, whitening))
        return pl

    def embed(self, x, preprocessed: bool = True, downsampling: int = 4, whitening: bool = True,
              windows_per_tower: Optional[int] = None) -> torch.Tensor:
        """Inference-mode embeddings (encoder.predict): x (n, L[, 1]) -> (n, E) fp32 device tensor."""
        x = torch.as_tensor(x)
        n = x.shape[0]
        x = x.reshape(n, -1).to(self.device)
        l0 = x.shape[1] if preprocessed else (x.shape[1] + downsampling - 1) // downsampling
        pl = self.plan(n, l0, False)
        self.last_infer_l0 = l0
        if preprocessed:
            self.load_preprocessed(pl, x)
        else:
            self.preprocess(pl, x, downsampling, whitening, windows_per_tower or n)
        return self.forward(pl, n, None)

    def embed_from_offsets(self, audio: torch.Tensor, offsets: torch.Tensor, raw_len: int, downsampling: int = 4,
                           whitening: bool = True, windows_per_tower: Optional[int] = None) -> torch.Tensor:
        """``embed`` fed from a device-resident recording buffer: window i is the ``raw_len`` samples at audio[offsets[i]]
        (voicemap_amd/shards.py); the crop happens inside the preprocessing kernel."""
        n = int(offsets.numel())
        l0 = (raw_len + downsampling - 1) // downsampling
        pl = self.plan(n, l0, False)
        self.last_infer_l0 = l0
        self.preprocess(pl, audio, downsampling, whitening, windows_per_tower or n,
                        offsets=offsets.to(self.device, torch.int64).contiguous(), raw_len=raw_len)
        return self.forward(pl, n, None)

    def siamese_eval(self, x1, x2, y, loss: str = "contrastive", preprocessed: bool = True, downsampling: int = 4,
                     whitening: bool = True):
        """test_on_batch of the siamese model: inference-mode forward + loss / accuracy (no gradients are used; the
        head kernel writes its backward outputs into scratch)."""
        x1 = torch.as_tensor(x1)
        x2 = torch.as_tensor(x2)
        pairs = x1.shape[0]
        x = torch.cat([x1.reshape(pairs, -1), x2.reshape(pairs, -1)], 0)
        self.embed(x, preprocessed, downsampling, whitening, windows_per_tower=pairs)
        pl = self.plan(2 * pairs, self.last_infer_l0, False)
        yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).to(self.device).contiguous()
        if "scratch" not in pl:
            pl["scratch"] = torch.empty(2 * pairs * self.E + self.E + 8, dtype=torch.float32, device=self.device)
        sc = pl["scratch"]
        off = 2 * pairs * self.E
        self._call("vm_siamese_head_loss", _p(pl["emb"]), _p(self.view("head.kernel")), _p(self.view("head.bias")), _p(yd),
                   pairs, self.E, HEADS[self.head], LOSSES[loss], 1.0, _p(pl["pred"]), _p(pl["loss_acc"]), _p(sc),
                   sc.data_ptr() + 4 * off, sc.data_ptr() + 4 * (off + self.E), _p(pl["head_ws"]), self.stream())
        return pl

    def classifier_head_eval(self, pl: dict, labels: torch.Tensor):
        """Dense(num_classes, softmax) + categorical CE / accuracy on the embeddings of an inference plan."""
        n = pl["n"]
        self._call("vm_dense_fwd", _p(pl["emb"]), _p(self.view("head.kernel")), _p(self.view("head.bias")), n, self.E,
                   self.num_classes, _p(pl["logits"]), self.stream())
        self._call("vm_softmax_cce", _p(pl["logits"]), _p(labels), n, self.num_classes, 1.0, _p(pl["prob"]), _p(pl["loss_acc"]),
                   None, _p(pl["cce_ws"]), self.stream())
        return pl["prob"]

    def siamese_predict(self, x1, x2, preprocessed: bool = True, downsampling: int = 4, whitening: bool = True):
        """siamese.predict([x1, x2]) -> (pairs, 1) probabilities (inference-mode BN)."""
        x1 = torch.as_tensor(x1)
        x2 = torch.as_tensor(x2)
        pairs = x1.shape[0]
        x = torch.cat([x1.reshape(pairs, -1), x2.reshape(pairs, -1)], 0)
        self.embed(x, preprocessed, downsampling, whitening, windows_per_tower=pairs)
        l0 = x.shape[1] if preprocessed else (x.shape[1] + downsampling - 1) // downsampling
        pl = self.plan(2 * pairs, l0, False)
        return self.siamese_head(pl, None).reshape(pairs, 1)
