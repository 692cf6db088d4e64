"""Host-side driver of the log-mel + 2-D CNN encoder variant (BASELINE.json config 4).

The variant is NOT in the reference.  It keeps the reference encoder's shape -- 4 x [conv -> ReLU -> BatchNorm ->
spatial dropout -> max-pool] -> global max-pool -> Dense(E), channels F, 2F, 3F, 4F (voicemap/models.py:13-39) -- over a (T, M)
log-mel image instead of the waveform: Conv2D(3 x 3, SAME), MaxPool2D(2, 2), GlobalMaxPool2D.

Everything runs through the C ABI (include/voicemap_hip.h).  A clip is kept as M "windows" (one per mel band) of T positions, so
* ``vm_stft_logmel`` writes the network input directly in that layout,
* a Conv2D(3 x 3) is ``vm_stack_windows`` (three neighbouring bands side by side in the channel dimension) followed by the
  library's k = 3 implicit-GEMM convolution along T -- forward, dgrad and wgrad are the 1-D entry points, W2d[kt][km][ci][co] is
  W1d[kt][km * C + ci][co]; ``vm_fold_windows`` is the adjoint of the stacking,
* BatchNorm (+ dropout + the time half of the pooling) are the 1-D kernels, the mel half of the pooling is
  ``vm_pool_windows_fwd / _bwd`` and ``vm_clip_max_fwd / _bwd``.
Parameters live in the same flat fp32 buffers as the 1-D engine's (one clip-norm reduction, one Adam pass, one all-reduce).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, spectro
from .engine import HEADS, LOSSES, FlatState, HipEncoderEngine, _DT, _TORCH_DT, _align, _p, _shared_stream


class HipSpectrogramEncoderEngine(HipEncoderEngine):
    """filters F -> channels F, 2F, 3F, 4F; head: None | 'uniform_euclidean' | 'weighted_l1'."""

    def __init__(self, filters: int, embedding_dimension: int, dropout: float = 0.05, head: Optional[str] = None, dtype: str = "f16",
                 device="cuda", bn_eps: float = 1e-3, bn_momentum: float = 0.99, unbiased_moving_variance: bool = True,
                 seed: Optional[int] = None, n_mels: int = spectro.N_MELS, win_length: int = spectro.WIN_LENGTH,
                 hop: int = spectro.HOP, log_floor: float = spectro.LOG_FLOOR):
        if not torch.cuda.is_available():
            raise RuntimeError("HipSpectrogramEncoderEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
        if head not in (None, "uniform_euclidean", "weighted_l1"):
            raise NotImplementedError(head)
        self.lib = _lib.lib()
        self.timed = {}
        self._call("vm_check_device")
        self.filters = int(filters)
        self.chan = [self.filters * (i + 1) for i in range(4)]
        assert all(c % 8 == 0 for c in self.chan), "filters must be a multiple of 8"
        self.nb = 4
        self.E = int(embedding_dimension)
        self.dropout = float(dropout)
        self.head = head
        self.num_classes = 0
        self.dtype = _DT[dtype]
        self.tdt = _TORCH_DT[self.dtype]
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._views = {}
        self.bn_eps, self.bn_momentum = float(bn_eps), float(bn_momentum)
        self.unbiased = bool(unbiased_moving_variance)
        self.n_mels, self.win_length, self.hop, self.log_floor = int(n_mels), int(win_length), int(hop), float(log_floor)
        # stacked channel counts of the four convolutions: 3 bands x C_in, padded to a multiple of 8 (block 1: 3 -> 8)
        self.cin = [1] + self.chan[:-1]
        self.cs = [_align(3 * c, 8) for c in self.cin]
        # 1-D view of the architecture for the shared flat layout: k = 3 kernels of shape (3, Cs_i, C_i)
        self.blocks = [(3, c, 2) for c in self.chan]
        spec = []
        for i, c in enumerate(self.chan):
            spec += [(f"conv{i+1}.kernel", (3, self.cs[i], c)), (f"conv{i+1}.bias", (c,)), (f"bn{i+1}.gamma", (c,)), (f"bn{i+1}.beta", (c,))]
        spec += [("dense.kernel", (self.chan[-1], self.E)), ("dense.bias", (self.E,))]
        if head == "uniform_euclidean":
            spec += [("head.kernel", (1, 1)), ("head.bias", (1,))]
        elif head == "weighted_l1":
            spec += [("head.kernel", (self.E, 1)), ("head.bias", (1,))]
        st = FlatState.__new__(FlatState)
        st.spec = spec
        st.offsets = OrderedDict()
        off = 0
        for name, shape in spec:
            n = int(np.prod(shape))
            st.offsets[name] = (off, n, shape)
            off += _align(n)
        self.spec, self.offsets, self.n_flat = spec, st.offsets, off
        self.n_params = sum(int(np.prod(self.public_shape(n_))) for n_, _ in spec)
        dev = self.device
        self.P = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        self.G, self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.nt_off: Dict[str, tuple] = OrderedDict()
        off = 0
        for i, c in enumerate(self.chan):
            self.nt_off[f"bn{i+1}.moving_mean"] = (off, c)
            off += _align(c)
            self.nt_off[f"bn{i+1}.moving_variance"] = (off, c)
            off += _align(c)
        self.NT = torch.zeros(off, dtype=torch.float32, device=dev)
        self.wf = {i: torch.empty(self.chan[i] * 3 * self.cs[i], dtype=self.tdt, device=dev) for i in range(4)}
        self.wd = {i: torch.empty(self.cs[i] * 3 * self.chan[i], dtype=self.tdt, device=dev) for i in range(4)}
        self._sq_ws = torch.empty(self.lib.query("vm_sqnorm_workspace_bytes", self.n_flat) // 8, dtype=torch.float64, device=dev)
        self._sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.beta_1, self.beta_2, self.adam_eps, self.decay, self.clipnorm = 1e-3, 0.9, 0.999, 1e-7, 0.0, 1.0
        self.iterations = 0
        self.last_infer_l0 = 0
        self.overlap_wgrad = True
        # block 1 (one input channel) on the vector ALUs (vm_conv2d_first_fwd / _wgrad) instead of as a band-stacked GEMM
        self.first_layer_direct = bool(self.lib.query("vm_conv2d_first_supported", self.chan[0], self.dtype))
        self.flat_dgrad = True   # dgrad over the concatenated windows (see backward)
        self.flat_fwd = True     # ... and the forward of the GEMM-shaped layers (vm_conv_fwd_flat)
        self.towers_concurrent = True   # forward: the two towers' GEMM launches of the small blocks side by side on two streams
        self.fuse_boundary = True   # BatchNorm + 2 x 2 pooling + band stacking as one pass per block boundary, and its adjoint
        self.fused_bn_sums = True   # ... which also leaves the two BatchNorm-backward sums of the block below (no reduce pass)
        self.side_stream = _shared_stream(self.device, "side")
        self.grad_sync = None
        self.grad_prescale = 1.0
        self._plans: Dict[tuple, dict] = {}
        self.is16 = self.dtype in (_lib.VM_BF16, _lib.VM_F16)
        self._init_loss_scale()
        self._init_zero_debias()
        self.basis = torch.from_numpy(spectro.dft_basis(self.win_length)).to(dev)
        self.melw = torch.from_numpy(spectro.mel_filterbank(self.n_mels)).to(dev)
        # 16-bit storage: the DFT on the f16 matrix pipe from hi / lo halves of basis and samples (vm_stft_logmel_f16s)
        self.stft_split = self.is16
        # (round 6) 16-bit storage: the log-mel image as TWO planes of the storage type (hi + what hi dropped), both multiplied by the first
        # convolution (filters split the same way inside): the one tensor of the variant with a pedestal and a single channel no longer
        # enters at 11 bits -- its rounding alone was 7.7e-4 of config 4's 1.27e-3 (tools/probe/config4_storage_sites.py)
        self.split_image = self.is16 and self.first_layer_direct and self.chan[0] % 32 == 0
        self.split_z = True   # ... and its own output z1 likewise (the largest remaining site: 5.0e-4), read back as z + z_lo by the boundary pass
        # ... or not read back at all: the boundary pass recomputes the nine-tap convolution from the image (cheaper than 312 MB of z per
        # 256 clips) and normalises the fp32 accumulator.  False: z_lo is stored and read (vm_bn_pool2d_stack_fwd_split)
        self.recompute_boundary = True
        self.basis16 = torch.empty(self.lib.query("vm_stft_split_basis_bytes", self.win_length) // 2, dtype=torch.float16, device=dev)
        self._call("vm_stft_split_basis", _p(self.basis), self.win_length, _p(self.basis16), self.stream())
        self.init_params(seed)

    # ---- parameters: the flat store keeps Conv2D kernels as (3, Cs, C_out) = (kT, [kM x C_in, zero padding], C_out) ----------------
    def public_shape(self, name):
        if name.startswith("conv") and name.endswith(".kernel"):
            i = int(name[4]) - 1
            return (3, 3, self.cin[i], self.chan[i])
        return self.offsets[name][2]

    def init_params(self, seed: Optional[int] = None):
        """Keras defaults: glorot_uniform Conv2D kernels (fan = 9 C), zero biases, gamma 1, beta 0, moving 0 / 1."""
        import os
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.seed = int(seed)
        g = torch.Generator().manual_seed(self.seed)
        self._drop_gen = torch.Generator(device=self.device)
        self._drop_gen.manual_seed(self.seed * 1000003 + 7919 * (int(os.environ.get("RANK", "0")) + 1))
        params = OrderedDict()
        for name, (o, n, shape) in self.offsets.items():
            if name.endswith(".kernel"):
                ps = self.public_shape(name)
                if len(ps) == 4:
                    fan_in, fan_out = 9 * ps[2], 9 * ps[3]
                else:
                    fan_in, fan_out = ps
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                params[name] = ((torch.rand(ps, generator=g, dtype=torch.float64) * 2 - 1) * lim).numpy()
            elif name.endswith(".gamma"):
                params[name] = np.ones(shape)
            else:
                params[name] = np.zeros(shape)
        for name, (o, n) in self.nt_off.items():
            self.NT[o:o + n] = 1.0 if name.endswith("moving_variance") else 0.0
        self.M.zero_()
        self.V.zero_()
        self.iterations = 0
        if hasattr(self, "ZD"):
            self.ZD.zero_()
            self.bn_steps = 0
        self.set_params(params)

    def set_params(self, params):
        for name, val in params.items():
            a = np.asarray(val, dtype=np.float32)
            v = self.view(name)
            if name.startswith("conv") and name.endswith(".kernel") and a.ndim == 4:   # (kT, kM, C_in, C_out) -> (kT, Cs, C_out)
                i = int(name[4]) - 1
                buf = np.zeros((3, self.cs[i], self.chan[i]), np.float32)
                buf[:, :3 * self.cin[i], :] = a.reshape(3, 3 * self.cin[i], self.chan[i])
                a = buf
            v.copy_(torch.as_tensor(a).to(self.device).reshape(v.shape))
        self.refresh_weights()

    def _public(self, name, t: torch.Tensor) -> np.ndarray:
        a = t.detach().cpu().numpy().copy()
        if name.startswith("conv") and name.endswith(".kernel"):
            i = int(name[4]) - 1
            a = a[:, :3 * self.cin[i], :].reshape(3, 3, self.cin[i], self.chan[i])
        return a

    def get_params(self):
        return OrderedDict((name, self._public(name, self.view(name))) for name in list(self.offsets) + list(self.nt_off))

    def get_grads(self):
        inv = np.float32(1.0 / float(self.loss_scale))   # G holds loss_scale x the gradients with f16 storage
        return OrderedDict((name, self._public(name, self.view(name, self.G)) * inv) for name in self.offsets)

    def refresh_weights(self):
        for i in range(4):
            self._call("vm_prep_conv_weights", _p(self.view(f"conv{i+1}.kernel")), self.cs[i], self.chan[i], self.dtype, _p(self.wf[i]),
                       _p(self.wd[i]), self.stream())

    def _fused_boundary(self, i: int) -> bool:
        """Is the boundary between blocks i and i + 1 one pass each way (vm_bn_pool2d_stack_fwd / vm_fold_pool_windows_bwd)?"""
        if not self.fuse_boundary or i < 0 or i >= 3:
            return False
        vec = 4 if self.dtype in (_lib.VM_F32, _lib.VM_F32S) else 8
        return self.chan[i] % vec == 0 and self.cs[i + 1] % vec == 0

    # ---- geometry --------------------------------------------------------------------------------------------------------------
    def _split_z(self, i: int) -> bool:
        """block 1's conv output on two planes of the storage type (z + what its rounding dropped; the boundary pass adds them)."""
        return i == 0 and self.split_image and self.split_z and self.stft_split and self._fused_boundary(0)

    def geometry(self, raw_len: int):
        T = spectro.n_frames(raw_len, self.win_length, self.hop)
        Ts, Ms = [T], [self.n_mels]
        for _ in range(4):
            Ts.append(Ts[-1] // 2)
            Ms.append(Ms[-1] // 2)
        assert Ts[3] >= 2 and Ms[3] >= 2, "clip too short / too few mel bands for four 2 x 2 poolings"
        return Ts, Ms

    def plan(self, n_clips: int, raw_len: int, training: bool) -> dict:
        key = (n_clips, raw_len, training)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        dev, tdt, f32 = self.device, self.tdt, torch.float32
        Ts, Ms = self.geometry(raw_len)
        pl = {"n": n_clips, "raw_len": raw_len, "T": Ts, "M": Ms, "training": training, "l0": raw_len}
        prow = self.lib.query("vm_bn_part_rows")
        for i, c in enumerate(self.chan):
            nw, L = n_clips * Ms[i], Ts[i]
            b = {"nw": nw}
            b["in"] = torch.zeros(nw, L + 2, self.cin[i], dtype=tdt, device=dev)       # block input, halo rows stay zero
            if i == 0 and self.split_image:
                b["in_lo"] = torch.zeros_like(b["in"])                                  # ... and what its storage type dropped
                b["z_lo"] = torch.empty(nw, L, c, dtype=tdt, device=dev)
            b["xs"] = torch.zeros(nw, L + 2, self.cs[i], dtype=tdt, device=dev)        # band-stacked
            b["z"] = torch.empty(nw, L, c, dtype=tdt, device=dev)
            if i < 3:
                b["q"] = torch.zeros(nw, Ts[i + 1] + 2, c, dtype=tdt, device=dev)      # pooled along T
            b["stat_rows"] = self.lib.query("vm_conv_stat_rows", L)
            for nm in ("mean", "invstd", "scale", "shift", "c1", "c2"):
                b[nm] = torch.zeros(2, c, dtype=f32, device=dev)
            if training:
                b["ssum"] = torch.empty(nw * b["stat_rows"], c, dtype=f32, device=dev)
                b["ssq"] = torch.empty(nw * b["stat_rows"], c, dtype=f32, device=dev)
                b["du"] = torch.zeros(nw, L + 2, c, dtype=tdt, device=dev)
                if i < 3:
                    b["dp"] = torch.empty(nw, Ts[i + 1], c, dtype=tdt, device=dev)    # gradient of q (un-padded)
                    if self._fused_boundary(i):   # the BatchNorm-backward sums vm_fold_pool_windows_bwd leaves per (window, workgroup row)
                        b["fs_rows"] = self.lib.query("vm_fold_pool_windows_rows", Ts[i + 1], c, self.cs[i + 1], self.dtype)
                        b["fs0"] = torch.empty(nw * b["fs_rows"], c, dtype=f32, device=dev)
                        b["fsa"] = torch.empty_like(b["fs0"])
                for nm in ("pa", "pb", "pdu"):
                    b[nm] = torch.empty(nw * prow, c, dtype=f32, device=dev)
                b["pdu_used"] = self.lib.query("vm_bn_part_rows_used", L, c, 2, self.dtype)
                # the weight gradient is a sum over positions, and consecutive windows -- each with its own zero halo rows in xs and
                # du -- are one window of g (L + 2) - 2 positions for vm_conv_wgrad (a halo row multiplies by zero, no tap reaches
                # across two of them): the same sum (1e-6, the order of fp32 partial sums) from fewer, longer reductions.  g = 8 is
                # the best of tools/probe/wgrad_merge_probe.py on the three GEMM-shaped layers (452 -> 372 us together)
                g = 8 if nw % 8 == 0 else 1
                b["wg_n"], b["wg_L"] = nw // g, g * (L + 2) - 2
                b["wgrad_ws"] = torch.empty(self.lib.query("vm_conv_wgrad_workspace_bytes", b["wg_n"], b["wg_L"], self.cs[i], c) // 4 + 16,
                                            dtype=f32, device=dev)
                b["ev"] = torch.cuda.Event()
                if i > 0:
                    b["dxs"] = torch.empty(nw, L + 2 * int(self.flat_dgrad), self.cs[i], dtype=tdt, device=dev)   # dgrad output (gradient of xs)
                    b["din"] = torch.empty(nw, L, self.cin[i], dtype=tdt, device=dev)  # folded: gradient of the block input
            pl[i] = b
        cl, nwl = self.chan[-1], n_clips * Ms[3]
        pl["gmax_ws"] = torch.empty(self.lib.query("vm_bn_drop_pool_gmax_workspace_bytes", nwl, cl) // 4, dtype=f32, device=dev)
        pl["gmax_w"] = torch.empty(nwl, cl, dtype=f32, device=dev)
        pl["gidx"] = torch.empty(nwl, cl, dtype=torch.int32, device=dev)
        pl["gmax"] = torch.empty(n_clips, cl, dtype=f32, device=dev)
        pl["widx"] = torch.empty(n_clips, cl, dtype=torch.int32, device=dev)
        pl["emb"] = torch.empty(n_clips, self.E, dtype=f32, device=dev)
        if training:
            pl["demb"] = torch.zeros(n_clips, self.E, dtype=f32, device=dev)
            pl["dgmax"] = torch.empty(n_clips, cl, dtype=f32, device=dev)
            pl["dgmax_w"] = torch.empty(nwl, cl, dtype=f32, device=dev)
        pl["pred"] = torch.empty(max(n_clips // 2, 1), dtype=f32, device=dev)
        pl["loss_acc"] = torch.zeros(2, dtype=f32, device=dev)
        pl["head_ws"] = torch.empty(4 * max(n_clips // 2, 1), dtype=f32, device=dev)
        cmax = max(self.chan)
        pl["cr_ws"] = torch.empty(self.lib.query("vm_colreduce_workspace_bytes", 2, cmax) // 8, dtype=torch.float64, device=dev)
        self._plans[key] = pl
        return pl

    # ---- the path --------------------------------------------------------------------------------------------------------------
    def features(self, pl: dict, raw: torch.Tensor):
        """raw (n_clips, raw_len) fp32 or int16 on the device -> pl[0]['in'] (the log-mel image as n_clips * n_mels windows)."""
        raw = raw.reshape(pl["n"], -1).contiguous()
        is16 = raw.dtype == torch.int16
        if not is16:
            raw = raw.to(torch.float32)
        assert raw.shape[1] == pl["raw_len"]
        if self.stft_split and self.split_image:
            self._call("vm_stft_logmel_f16s_split", _p(raw), int(is16), pl["n"], pl["raw_len"], self.win_length, self.hop, _p(self.basis16),
                       _p(self.melw), self.n_mels, self.log_floor, self.dtype, _p(pl[0]["in"]), _p(pl[0]["in_lo"]), self.stream())
        elif self.stft_split:
            self._call("vm_stft_logmel_f16s", _p(raw), int(is16), pl["n"], pl["raw_len"], self.win_length, self.hop, _p(self.basis16),
                       _p(self.melw), self.n_mels, self.log_floor, self.dtype, _p(pl[0]["in"]), self.stream())
        else:
            self._call("vm_stft_logmel", _p(raw), int(is16), pl["n"], pl["raw_len"], self.win_length, self.hop, _p(self.basis), _p(self.melw),
                       self.n_mels, self.log_floor, self.dtype, _p(pl[0]["in"]), self.stream())
        pl["_raw_keepalive"] = raw

    def make_drop_masks(self, n_clips: int, generator: Optional[torch.Generator] = None):
        """SpatialDropout2D keep masks (n_clips, C) / (1 - rate) per block; None when rate == 0."""
        if self.dropout <= 0.0:
            return None
        out = []
        for c in self.chan:
            u = torch.rand(n_clips, c, device=self.device, generator=generator if generator is not None else self._drop_gen)
            out.append((u >= self.dropout).to(torch.float32) / (1.0 - self.dropout))
        return out

    def forward(self, pl: dict, clips_per_tower: int, drop_masks=None):
        st, n, dt = self.stream(), pl["n"], self.dtype
        training = pl["training"]
        n_towers = n // clips_per_tower if training else 1
        assert (not training) or (n % clips_per_tower == 0 and n_towers <= 2)
        cpt = clips_per_tower if training else n
        pl["cpt"] = cpt
        pl["drop_w"] = [None] * 4
        if training:
            self.bn_steps += 1
            self._bn_t = self.bn_steps
        for i, c in enumerate(self.chan):
            b, L, Mi = pl[i], pl["T"][i], pl["M"][i]
            nw, wpt = b["nw"], cpt * Mi
            ssum = _p(b["ssum"]) if training else None
            ssq = _p(b["ssq"]) if training else None
            stat_rows_per_tower = wpt * b["stat_rows"]
            if i == 0 and self.first_layer_direct and self.split_image and self.stft_split:
                self._call("vm_conv2d_first_fwd_split", _p(b["in"]), _p(b["in_lo"]), _p(self.view("conv1.kernel")), _p(self.view("conv1.bias")), n,
                           Mi, L, self.cs[0], c, dt, _p(b["z"]), _p(b["z_lo"]) if (self._split_z(i) and not self.recompute_boundary) else None,
                           ssum, ssq, st)
            elif i == 0 and self.first_layer_direct:
                self._call("vm_conv2d_first_fwd", _p(b["in"]), _p(self.view("conv1.kernel")), _p(self.view("conv1.bias")), n, Mi, L,
                           self.cs[0], c, dt, _p(b["z"]), ssum, ssq, st)
            else:
                if not self._fused_boundary(i - 1):   # (else the previous block's pooling pass wrote xs itself)
                    self._call("vm_stack_windows", _p(b["in"]), n, Mi, L + 2, self.cin[i], self.cs[i], dt, _p(b["xs"]), st)
                flat = self.flat_fwd and wpt * (L + 2) * max(c, self.cs[i]) < 2 ** 31
                if flat:
                    # a tower's windows -- each with its own zero halo rows in xs -- as ONE sequence on full 128-row tiles; the epilogue
                    # drops the halo positions and writes the same un-padded z.  The statistics rows are then per tile of the
                    # concatenation: one launch per tower keeps them per tower
                    srows = self.lib.query("vm_conv_flat_stat_rows", wpt, L)
                    if training and b.get("flat_rows") != (n_towers, srows):
                        b["ssum_f"] = torch.empty(n_towers * srows, c, dtype=torch.float32, device=self.device)
                        b["ssq_f"] = torch.empty_like(b["ssum_f"])
                        b["flat_rows"] = (n_towers, srows)
                    # the second tower's launch goes to the side stream where one tower does not fill the chip (blocks 3, 4: a few
                    # hundred 128-row tiles a tower)
                    two = self.towers_concurrent and training and n_towers == 2 and wpt * (L + 2) < 128 * 2048
                    if two:
                        b["ev"].record()   # the stacked input is complete here
                    for tw in reversed(range(n_towers)):
                        args = (b["xs"][tw * wpt:].data_ptr(), _p(self.wf[i]), _p(self.view(f"conv{i+1}.bias")), wpt, L,
                                self.cs[i], c, dt, b["z"][tw * wpt:].data_ptr(),
                                b["ssum_f"][tw * srows:].data_ptr() if training else None,
                                b["ssq_f"][tw * srows:].data_ptr() if training else None)
                        if two and tw == 1:
                            with torch.cuda.stream(self.side_stream):
                                self.side_stream.wait_event(b["ev"])
                                self._call("vm_conv_fwd_flat", *args, self.stream())
                        else:
                            self._call("vm_conv_fwd_flat", *args, st)
                    if two:
                        torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
                    if training:
                        ssum, ssq, stat_rows_per_tower = _p(b["ssum_f"]), _p(b["ssq_f"]), srows
                else:
                    self._call("vm_conv_fwd", _p(b["xs"]), _p(self.wf[i]), _p(self.view(f"conv{i+1}.bias")), nw, L, self.cs[i], c, dt, _p(b["z"]),
                               ssum, ssq, st)
            gam, bet = _p(self.view(f"bn{i+1}.gamma")), _p(self.view(f"bn{i+1}.beta"))
            mm, mv = _p(self.view(f"bn{i+1}.moving_mean")), _p(self.view(f"bn{i+1}.moving_variance"))
            if training:
                zd, zc = self._zd(i)
                self._call("vm_bn_finalize", ssum, ssq, stat_rows_per_tower, n_towers, c, float(wpt * L), gam, bet, self.bn_eps,
                           self.bn_momentum, int(self.unbiased), mm, mv, _p(b["mean"]), _p(b["invstd"]), _p(b["scale"]), _p(b["shift"]),
                           _p(pl["cr_ws"]), zd, zc, None, None, None, None, st)
            else:
                self._call("vm_bn_infer_affine", gam, bet, mm, mv, self.bn_eps, c, _p(b["scale"]), _p(b["shift"]), st)
            dm = None
            if training and drop_masks is not None and drop_masks[i] is not None:
                dm = drop_masks[i].repeat_interleave(Mi, dim=0).contiguous()   # one mask per clip -> one row per window
                pl["drop_w"][i] = dm
            if i == 3:
                self._call("vm_bn_drop_pool_gmax_fwd", _p(b["z"]), _p(b["scale"]), _p(b["shift"]), _p(dm), nw, wpt, L, c, 2, dt,
                           _p(pl["gmax_w"]), _p(pl["gidx"]), _p(pl["gmax_ws"]), st)
                self._call("vm_clip_max_fwd", _p(pl["gmax_w"]), n, Mi, 2 * (Mi // 2), c, _p(pl["gmax"]), _p(pl["widx"]), st)
            elif self._split_z(i) and self.recompute_boundary:
                # block 1's boundary without the round trip of z: the convolution redone on the two-plane image, affine + pool + stacking
                # out of the fp32 accumulator (z, left by the launch above, is the backward's)
                self._call("vm_conv2d_first_bn_pool_stack", _p(b["in"]), _p(b["in_lo"]), _p(self.view("conv1.kernel")), _p(self.view("conv1.bias")),
                           _p(b["scale"]), _p(b["shift"]), _p(dm), n, Mi, cpt, L, self.cs[0], c, self.cs[i + 1], dt, _p(b["q"]),
                           _p(pl[i + 1]["xs"]), st)
            elif self._split_z(i):
                # block 1's z on two planes (round 6): the affine sees their sum
                self._call("vm_bn_pool2d_stack_fwd_split", _p(b["z"]), _p(b["z_lo"]), _p(b["scale"]), _p(b["shift"]), _p(dm), n, Mi, cpt, L, c,
                           self.cs[i + 1], dt, _p(b["q"]), _p(pl[i + 1]["xs"]), st)
            elif self._fused_boundary(i):
                # BatchNorm affine + dropout + MaxPool2D(2, 2) + the next block's band stacking in one pass over z
                self._call("vm_bn_pool2d_stack_fwd", _p(b["z"]), _p(b["scale"]), _p(b["shift"]), _p(dm), n, Mi, cpt, L, c, self.cs[i + 1], dt,
                           _p(b["q"]), _p(pl[i + 1]["xs"]), st)
            else:
                self._call("vm_bn_drop_pool_fwd", _p(b["z"]), _p(b["scale"]), _p(b["shift"]), _p(dm), nw, wpt, L, c, 2, dt, _p(b["q"]), st)
                self._call("vm_pool_windows_fwd", _p(b["q"]), n, Mi, pl["T"][i + 1] + 2, c, dt, _p(pl[i + 1]["in"]), st)
        self._call("vm_dense_fwd", _p(pl["gmax"]), _p(self.view("dense.kernel")), _p(self.view("dense.bias")), n, self.chan[-1], self.E,
                   _p(pl["emb"]), st)
        return pl["emb"]

    def backward(self, pl: dict, sync_tail: bool = False):
        assert pl["training"]
        st, n, dt, cpt, G = self.stream(), pl["n"], self.dtype, pl["cpt"], self.G
        sync_tail = sync_tail and self.grad_sync is not None and hasattr(self.grad_sync, "begin_tail")
        cl = self.chan[-1]
        self._call("vm_dense_bwd", _p(pl["gmax"]), _p(self.view("dense.kernel")), _p(pl["demb"]), n, cl, self.E,
                   _p(self.view("dense.kernel", G)), _p(self.view("dense.bias", G)), _p(pl["dgmax"]), st)
        self._call("vm_clip_max_bwd", _p(pl["dgmax"]), _p(pl["widx"]), n, pl["M"][3], cl, _p(pl["dgmax_w"]), st)
        for i in range(3, -1, -1):
            c, b, L, Mi = self.chan[i], pl[i], pl["T"][i], pl["M"][i]
            nw, wpt = b["nw"], cpt * Mi
            dm = _p(pl["drop_w"][i])
            bn_grads = (_p(self.view(f"bn{i+1}.gamma", G)), _p(self.view(f"bn{i+1}.beta", G)))
            if i == 3:
                head = (_p(b["z"]), _p(pl["dgmax_w"]), _p(pl["gidx"]))
                self._call("vm_bn_pool_bwd_reduce_gmax", *head, _p(b["scale"]), _p(b["shift"]), _p(b["mean"]), _p(b["invstd"]), dm, nw, wpt,
                           L, c, 2, dt, _p(b["pa"]), _p(b["pb"]), st)
            else:
                head = (_p(b["z"]), _p(b["dp"]))
                if not (self.fused_bn_sums and "fs0" in b):
                    self._call("vm_bn_pool_bwd_reduce", *head, _p(b["scale"]), _p(b["shift"]), _p(b["mean"]), _p(b["invstd"]), dm, nw, wpt, L,
                               c, 2, dt, _p(b["pa"]), _p(b["pb"]), st)
            if i < 3 and self.fused_bn_sums and "fs0" in b:
                # sum dp and sum dp * q came out of vm_fold_pool_windows_bwd (q = this block's pooled BatchNorm output): no pass over (z, dp)
                self._call("vm_bn_bwd_from_sums_finalize", _p(b["fs0"]), _p(b["fsa"]), b["fs_rows"], _p(b["z"]), _p(b["dp"]), _p(b["scale"]),
                           _p(b["shift"]), _p(b["mean"]), _p(b["invstd"]), dm, nw, wpt, L, c, 2, dt, 1, float(wpt * L), _p(b["c1"]),
                           _p(b["c2"]), *bn_grads, _p(pl["cr_ws"]), st)
            else:
                self._call("vm_bn_bwd_finalize", _p(b["pa"]), _p(b["pb"]), nw, wpt, c, float(wpt * L), _p(b["c1"]), _p(b["c2"]), *bn_grads,
                           _p(pl["cr_ws"]), st)
            self._call("vm_bn_pool_bwd_apply_gmax" if i == 3 else "vm_bn_pool_bwd_apply", *head, _p(b["scale"]), _p(b["shift"]),
                       _p(b["mean"]), _p(b["invstd"]), dm, _p(b["c1"]), _p(b["c2"]), nw, wpt, L, c, 2, dt, _p(b["du"]), _p(b["pdu"]), st)
            if b["pdu_used"] == 1:   # short windows: one live partial row per window, the other seven are zeros
                self._call("vm_colsum_strided", _p(b["pdu"]), nw, b["pdu"].shape[0] // nw, c, _p(self.view(f"conv{i+1}.bias", G)),
                           _p(pl["cr_ws"]), st)
            else:
                self._call("vm_colsum", _p(b["pdu"]), b["pdu"].shape[0], c, _p(self.view(f"conv{i+1}.bias", G)), _p(pl["cr_ws"]), st)
            gw = _p(self.view(f"conv{i+1}.kernel", G))

            def wgrad(stream):
                if i == 0 and self.first_layer_direct:
                    if "c2f_ws" not in b:
                        b["c2f_ws"] = torch.empty(self.lib.query("vm_conv2d_first_wgrad_workspace_bytes", n, Mi, c) // 4 + 16,
                                                  dtype=torch.float32, device=self.device)
                    self._call("vm_conv2d_first_wgrad", _p(b["in"]), _p(b["du"]), n, Mi, L, self.cs[0], c, dt, _p(b["c2f_ws"]), gw, stream)
                else:
                    self._call("vm_conv_wgrad", _p(b["xs"]), _p(b["du"]), b["wg_n"], b["wg_L"], self.cs[i], c, dt, _p(b["wgrad_ws"]), gw, stream)

            if self.overlap_wgrad:
                b["ev"].record()
                with torch.cuda.stream(self.side_stream):
                    self.side_stream.wait_event(b["ev"])
                    wgrad(self.stream())
            else:
                wgrad(st)
            if i == 1 and sync_tail:
                if "sync_ev" not in pl:
                    pl["sync_ev"] = torch.cuda.Event()
                pl["sync_ev"].record()
                self.grad_sync.begin_tail(self, pl["sync_ev"])
            if i > 0:
                flat = self.flat_dgrad and nw * (L + 2) * max(c, self.cs[i]) < 2 ** 31
                if flat:
                    # all windows, each with its own zero halo rows in du, are ONE window of nw (L + 2) - 2 positions for the k = 3
                    # dgrad: full 128-row tiles instead of windows of 149 .. 37 rows.  Output row p lands on padded row p + 1 of dxs,
                    # so a window's rows 1 .. L are what the per-window launch computes and its halo rows take the junk
                    self._call("vm_conv_dgrad", _p(b["du"]), _p(self.wd[i]), 1, nw * (L + 2) - 2, self.cs[i], c, dt,
                               b["dxs"].data_ptr() + self.cs[i] * b["dxs"].element_size(), st)
                else:   # per window, un-padded rows (the buffer is large enough either way)
                    self._call("vm_conv_dgrad", _p(b["du"]), _p(self.wd[i]), nw, L, self.cs[i], c, dt, _p(b["dxs"]), st)
                if self._fused_boundary(i - 1):
                    # adjoint of the stacking + the mel half of the pooling backward in one pass: dxs -> gradient of the previous block's q
                    sums = self.fused_bn_sums and "fs0" in pl[i - 1]
                    self._call("vm_fold_pool_windows_bwd", _p(b["dxs"]), _p(pl[i - 1]["q"]), n, pl["M"][i - 1], L, self.cin[i], self.cs[i],
                               int(flat), dt, _p(pl[i - 1]["dp"]), _p(pl[i - 1]["fs0"]) if sums else None,
                               _p(pl[i - 1]["fsa"]) if sums else None, st)
                else:
                    self._call("vm_fold_windows", _p(b["dxs"]), n, Mi, L, self.cin[i], self.cs[i], int(flat), dt, _p(b["din"]), st)
                    # gradient of the previous block's pooled output -> gradient of its time-pooled tensor q
                    self._call("vm_pool_windows_bwd", _p(pl[i - 1]["q"]), _p(b["din"]), n, pl["M"][i - 1], L, self.cin[i], dt,
                               _p(pl[i - 1]["dp"]), st)
        if self.overlap_wgrad:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)

    # ---- steps (raw 16 kHz windows in; ``preprocessed`` / ``downsampling`` / ``whitening`` are the 1-D engine's and ignored) --------
    def _raw(self, x) -> torch.Tensor:
        x = torch.as_tensor(x)
        x = x.reshape(x.shape[0], -1)
        return x.to(self.device) if x.dtype == torch.int16 else x.to(self.device, torch.float32)

    def siamese_train_step(self, x1, x2, y, loss: str = "contrastive", preprocessed: bool = False, downsampling: int = 1,
                           whitening: bool = False, drop_masks="auto", apply_update: bool = True):
        a, b2 = self._raw(x1), self._raw(x2)
        pairs = a.shape[0]
        x = torch.cat([a, b2], 0)
        pl = self.plan(2 * pairs, x.shape[1], True)
        self.features(pl, x)
        if isinstance(drop_masks, str):
            drop_masks = self.make_drop_masks(2 * pairs)
        yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).to(self.device).contiguous()
        self.forward(pl, pairs, drop_masks)
        self.siamese_head(pl, yd, loss)
        self.backward(pl, sync_tail=apply_update)
        if apply_update:
            self.optimizer_step()
        return pl

    def embed(self, x, preprocessed: bool = False, downsampling: int = 1, whitening: bool = False, windows_per_tower=None) -> torch.Tensor:
        x = self._raw(x)
        pl = self.plan(x.shape[0], x.shape[1], False)
        self.last_infer_l0 = x.shape[1]
        self.features(pl, x)
        return self.forward(pl, x.shape[0], None)

    def siamese_eval(self, x1, x2, y, loss: str = "contrastive", preprocessed: bool = False, downsampling: int = 1, whitening: bool = False):
        a, b2 = self._raw(x1), self._raw(x2)
        pairs = a.shape[0]
        self.embed(torch.cat([a, b2], 0))
        pl = self.plan(2 * pairs, self.last_infer_l0, False)
        yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).to(self.device).contiguous()
        if "scratch" not in pl:
            pl["scratch"] = torch.empty(2 * pairs * self.E + self.E + 8, dtype=torch.float32, device=self.device)
        sc = pl["scratch"]
        off = 2 * pairs * self.E
        self._call("vm_siamese_head_loss", _p(pl["emb"]), _p(self.view("head.kernel")), _p(self.view("head.bias")), _p(yd), pairs, self.E,
                   HEADS[self.head], LOSSES[loss], 1.0, _p(pl["pred"]), _p(pl["loss_acc"]), _p(sc), sc.data_ptr() + 4 * off,
                   sc.data_ptr() + 4 * (off + self.E), _p(pl["head_ws"]), self.stream())
        return pl

    def siamese_predict(self, x1, x2, preprocessed: bool = False, downsampling: int = 1, whitening: bool = False):
        a, b2 = self._raw(x1), self._raw(x2)
        pairs = a.shape[0]
        self.embed(torch.cat([a, b2], 0))
        pl = self.plan(2 * pairs, self.last_infer_l0, False)
        return self.siamese_head(pl, None).reshape(pairs, 1)

    # the 1-D engine's data paths that have no meaning here
    def preprocess(self, *a, **k):
        raise NotImplementedError("the spectrogram engine takes raw windows; its front-end is vm_stft_logmel")

    load_preprocessed = siamese_train_step_from_offsets = embed_from_offsets = classifier_train_step = preprocess
