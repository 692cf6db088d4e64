"""ctypes binding of libvoicemap_hip.so (the C ABI declared in include/voicemap_hip.h).

There is NO fallback: if the shared library is missing (and cannot be built because hipcc is absent) the
import of the product path fails loudly.  The oracle under ``oracle/`` is never imported from here.
"""
import ctypes
import os
import re

# torch must be imported BEFORE the shared library is loaded: the wheel bundles its own libamdhip64.so / HSA
# runtime, and the process must end up with ONE HIP runtime (ours then binds to the copy torch already mapped,
# so torch's streams and allocations are valid in our launches).  Loading ours first leaves two runtimes and
# hipGetDevice fails with "no ROCm-capable device is detected".
import torch  # noqa: F401
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOICEMAP_HIP_LIB") or os.path.join(_HERE, "lib", "libvoicemap_hip.so")  # env: A/B experiments
HEADER_PATH = os.path.join(_HERE, "..", "include", "voicemap_hip.h")

VM_F32, VM_BF16, VM_F32S, VM_F16 = 0, 1, 2, 3
ABI_VERSION = 11  # include/voicemap_hip.h vm_abi_version(): checked when the library is loaded
VM_LOSS_CONTRASTIVE, VM_LOSS_BCE = 0, 1
VM_HEAD_UNIFORM_EUCLIDEAN, VM_HEAD_WEIGHTED_L1 = 0, 1
VM_DIST_EUCLIDEAN, VM_DIST_COSINE, VM_DIST_DOT = 0, 1, 2

P, I, L, F, D = c_void_p, c_int, c_int64, c_float, c_double

# name -> (restype, argtypes); must list every function the header declares (tests/test_abi.py checks).
SIGNATURES = {
    "vm_last_error": (c_char_p, []),
    "vm_abi_version": (I, []),
    "vm_check_device": (I, []),
    "vm_fill_zero": (I, [P, L, P]),
    "vm_event_create": (I, [ctypes.POINTER(c_void_p)]),
    "vm_event_destroy": (I, [P]),
    "vm_event_record": (I, [P, P]),
    "vm_stream_wait_event": (I, [P, P]),
    "vm_program_run": (I, [P, L, P]),
    "vm_program_table_hash": (L, []),
    "vm_set_tuning": (I, [c_char_p, I]),
    "vm_mfma_rate_probe": (I, [I, I, P, P]),
    "vm_mfma_rate_probe_flops": (L, [I]),
    "vm_decimate_whiten_workspace_bytes": (L, [L]),
    "vm_decimate_whiten": (I, [P, I, L, L, I, I, F, L, P, P, P]),
    "vm_crop_decimate_whiten": (I, [P, I, P, L, L, I, I, F, L, P, P, P]),
    "vm_conv1_stat_rows": (L, [L]),
    "vm_conv1_fwd": (I, [P, P, P, L, L, I, I, P, P, P, P]),
    "vm_conv1_wgrad_workspace_bytes": (L, [L, I]),
    "vm_conv1_wgrad": (I, [P, P, L, L, I, I, P, P, P]),
    "vm_conv1_fused_fwd": (I, [P, P, P, P, P, L, L, I, I, I, I, P, P, P, P]),
    "vm_conv1_fused_bwd_workspace_bytes": (L, [L, L, I]),
    "vm_conv1_fused_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P, P]),
    "vm_conv_stat_rows": (L, [L]),
    "vm_conv_fwd": (I, [P, P, P, L, L, I, I, I, P, P, P, P]),
    "vm_conv_dgrad": (I, [P, P, L, L, I, I, I, P, P]),
    "vm_conv_flat_stat_rows": (L, [L, L]),
    "vm_conv_fwd_flat": (I, [P, P, P, L, L, I, I, I, P, P, P, P]),
    "vm_conv_fwd_e_supported": (I, [L, L, I, I, I]),
    "vm_conv_fwd_e": (I, [P, P, P, P, L, L, I, I, I, P, P, P, P, P]),
    "vm_fold_bn_weights": (I, [P, P, P, P, I, I, I, I, P, P, P, P, P]),
    "vm_conv_fwd_fold_supported": (I, [L, L, I, I, I, I]),
    "vm_conv_fwd_fold": (I, [P, P, P, P, P, L, L, L, I, I, I, P, P, P, P, P, P, P, P]),
    "vm_pack_nt_weights_supported": (I, [I, I, I]),
    "vm_pack_nt_weights": (I, [P, I, I, I, I, P, P]),
    "vm_pack_nt_weights_batch": (I, [I, P, P, P, P, I, P, P]),
    "vm_conv_fwd_pool_supported": (I, [L, L, I, I, I]),
    "vm_conv_fwd_pool": (I, [P, P, P, P, P, L, L, I, I, I, P, P, P]),
    "vm_conv_dgrad_bnred_rows": (L, [L]),
    "vm_conv_dgrad_bnred_supported": (I, [L, L, I, I, I]),
    "vm_conv_dgrad_bnred": (I, [P, P, L, L, I, I, I, P, P, I, P, P, P, P]),
    "vm_prep_conv_weights_batch": (I, [I, P, P, P, I, P, P, P, P]),
    "vm_conv_wgrad_splits": (I, [L, L, I, I]),
    "vm_conv_wgrad_workspace_bytes": (L, [L, L, I, I]),
    "vm_conv_wgrad": (I, [P, P, L, L, I, I, I, P, P, P]),
    "vm_conv_wgrad_fold_workspace_bytes": (L, [L, L, L, I, I]),
    "vm_conv_wgrad_fold": (I, [P, P, L, L, L, I, I, I, P, P, P, P, P, P]),
    "vm_conv_wgrad_fold_finish": (I, [P, L, L, L, I, I, P, P, P, P, P]),
    "vm_prep_conv_weights": (I, [P, I, I, I, P, P, P]),
    "vm_colreduce_workspace_bytes": (L, [I, I]),
    "vm_bn_finalize": (I, [P, P, L, I, I, D, P, P, F, F, I, P, P, P, P, P, P, P, P, F, P, P, P, P, P]),
    "vm_bn_infer_affine": (I, [P, P, P, P, F, I, P, P, P]),
    "vm_bn_drop_pool_fwd": (I, [P, P, P, P, L, L, L, I, I, I, P, P]),
    "vm_bn_part_rows": (I, []),
    "vm_bn_pool_bwd_reduce": (I, [P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_pool_bwd_reduce_pooled": (I, [P, P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_bwd_from_sums": (I, [P, P, L, P, P, P, P, P, P, P, L, L, L, I, I, I, I, P, P, P]),
    "vm_bn_bwd_from_sums_finalize": (I, [P, P, L, P, P, P, P, P, P, P, L, L, L, I, I, I, I, D, P, P, P, P, P, P]),
    "vm_bn_bwd_finalize": (I, [P, P, L, L, I, D, P, P, P, P, P, P]),
    "vm_bn_pool_bwd_apply": (I, [P, P, P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_pool_bwd_reduce_gmax": (I, [P, P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_bwd_gmax_finalize": (I, [P, P, P, P, P, P, P, P, L, L, L, I, I, I, D, P, P, P, P, P]),
    "vm_bn_pool_bwd_apply_gmax": (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_pool_bwd_apply_pairs": (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, I, I, P, P, P, P]),
    "vm_colsum": (I, [P, L, I, P, P, P]),
    "vm_bn_part_rows_used": (I, [L, I, I, I]),
    "vm_colsum_strided": (I, [P, L, I, I, P, P, P]),
    "vm_du_tower_sums": (I, [P, P, L, L, L, I, I, P, P, P, P]),
    "vm_bn_drop_pool_gmax_workspace_bytes": (L, [L, I]),
    "vm_bn_drop_pool_gmax_fwd": (I, [P, P, P, P, L, L, L, I, I, I, P, P, P, P]),
    "vm_bn_drop_pool_gmax_partials": (I, [P, P, P, P, L, L, L, I, I, I, P, P, P]),
    "vm_bn_drop_pool_gmax_partials_e": (I, [P, P, P, P, L, L, L, I, I, P, P, P]),
    "vm_bn_bwd_gmax_finalize_e": (I, [P, P, P, P, P, P, P, P, L, L, L, I, I, D, P, P, P, P, P]),
    "vm_bn_pool_bwd_apply_pairs_gmax": (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, L, I, I, P, P, P]),
    "vm_global_maxpool_fwd": (I, [P, L, L, I, I, P, P, P]),
    "vm_global_maxpool_bwd": (I, [P, P, L, L, I, I, P, P]),
    "vm_dense_fwd": (I, [P, P, P, L, I, I, P, P]),
    "vm_dense_bwd": (I, [P, P, P, L, I, I, P, P, P, P]),
    "vm_siamese_head_loss": (I, [P, P, P, P, L, I, I, I, F, P, P, P, P, P, P, P]),
    "vm_siamese_head_reduce": (I, [P, P, L, I, I, P, P, P, P]),
    "vm_tail_fwd_bwd_supported": (I, [I, I]),
    "vm_tail_fwd_bwd": (I, [P, P, I, P, P, P, P, P, P, P, L, I, I, I, I, F, P, P, P, P, P, P]),
    "vm_tail_param_grads": (I, [P, P, P, P, L, I, I, I, P, P, P, P, P, P]),
    "vm_softmax_cce": (I, [P, P, L, I, F, P, P, P, P, P]),
    "vm_sqnorm_workspace_bytes": (L, [L]),
    "vm_grad_sqnorm": (I, [P, L, P, P, P]),
    "vm_adam_clip_step": (I, [P, P, P, P, L, F, F, F, F, F, F, P, P, I, P, P]),
    "vm_nshot_distances": (I, [P, P, L, I, I, I, I, P, P, P]),
    "vm_nshot_indexed": (I, [P, L, P, P, L, I, I, I, I, P, P, P]),
    "vm_pairdist_workspace_bytes": (L, [L, L]),
    "vm_pairdist_argmin": (I, [P, P, L, L, I, I, L, P, P, P, P, P]),
    "vm_stft_frames": (L, [L, I, I]),
    "vm_stft_logmel": (I, [P, I, L, L, I, I, P, P, I, F, I, P, P]),
    "vm_stft_split_basis_bytes": (L, [I]),
    "vm_stft_split_basis": (I, [P, I, P, P]),
    "vm_stft_logmel_f16s": (I, [P, I, L, L, I, I, P, P, I, F, I, P, P]),
    "vm_stft_logmel_f16s_split": (I, [P, I, L, L, I, I, P, P, I, F, I, P, P, P]),
    "vm_conv2d_first_supported": (I, [I, I]),
    "vm_conv2d_first_fwd": (I, [P, P, P, L, I, L, I, I, I, P, P, P, P]),
    "vm_conv2d_first_fwd_split": (I, [P, P, P, P, L, I, L, I, I, I, P, P, P, P, P]),
    "vm_conv2d_first_bn_pool_stack": (I, [P, P, P, P, P, P, P, L, I, L, L, I, I, I, I, P, P, P]),
    "vm_conv2d_first_wgrad_workspace_bytes": (L, [L, I, I]),
    "vm_conv2d_first_wgrad": (I, [P, P, L, I, L, I, I, I, P, P, P]),
    "vm_stack_windows": (I, [P, L, I, L, I, I, I, P, P]),
    "vm_fold_windows": (I, [P, L, I, L, I, I, I, I, P, P]),
    "vm_bn_pool2d_stack_fwd": (I, [P, P, P, P, L, I, L, L, I, I, I, P, P, P]),
    "vm_bn_pool2d_stack_fwd_split": (I, [P, P, P, P, P, L, I, L, L, I, I, I, P, P, P]),
    "vm_fold_pool_windows_rows": (L, [L, I, I, I]),
    "vm_fold_pool_windows_bwd": (I, [P, P, L, I, L, I, I, I, I, P, P, P, P]),
    "vm_pool_windows_fwd": (I, [P, L, I, L, I, I, P, P]),
    "vm_pool_windows_bwd": (I, [P, P, L, I, L, I, I, P, P]),
    "vm_clip_max_fwd": (I, [P, L, I, I, I, P, P, P]),
    "vm_clip_max_bwd": (I, [P, P, L, I, I, P, P]),
}


def header_functions(path=HEADER_PATH):
    """Names of all functions declared in include/voicemap_hip.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vm_[a-z0-9_]+)\s*\(", src)))


class VoicemapHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            from . import build as _build
            _build.build(verbose=False)  # raises if hipcc is missing too
        self.cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        self.tuning_epoch = 0   # bumped by every vm_set_tuning through call(): recorded launch sequences (engine.py) are keyed on it
        self.abi = self.cdll.vm_abi_version()
        if self.abi != ABI_VERSION:  # a stale prebuilt library: its entry points take different arguments
            raise VoicemapHipError("%s reports ABI %d, this package binds ABI %d -- rebuild it (python -m voicemap_amd.build --force)"
                                   % (LIB_PATH, self.abi, ABI_VERSION))
        # experiments: kernel-selection knobs for a whole process (a pytest run under another knob value): VOICEMAP_TUNE="key=value,..."
        for kv in [t for t in os.environ.get("VOICEMAP_TUNE", "").split(",") if t.strip()]:
            k, v = kv.split("=")
            self.call("vm_set_tuning", k.strip().encode(), int(v))

    def call(self, name, *args):
        """Call an int-returning entry point; raise with vm_last_error() on failure."""
        if name == "vm_set_tuning":
            self.tuning_epoch += 1
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.vm_last_error()
            raise VoicemapHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else ""))

    def query(self, name, *args):
        return getattr(self.cdll, name)(*args)


_LIB = None
_PROGRAM_TABLE = False


def program_table():
    """(name -> function id, name -> "PILFD" argument types) of vm_program_run, or None when the loaded library was generated from
    another table than this binding's (tools/gen_program_run.py: ids are positions in the name-sorted list of int-returning entry
    points whose arguments are pointers / int / int64 / float / double)."""
    global _PROGRAM_TABLE
    if _PROGRAM_TABLE is False:
        import zlib
        code = {P: "P", I: "I", L: "L", F: "F", D: "D"}
        tab = [(n, "".join(code[a] for a in SIGNATURES[n][1])) for n in sorted(SIGNATURES)
               if SIGNATURES[n][0] is I and n != "vm_program_run" and all(a in code for a in SIGNATURES[n][1])]
        h = zlib.crc32(";".join("%s:%s" % t for t in tab).encode()) & 0x7FFFFFFF
        ok = lib().cdll.vm_program_table_hash() == h
        _PROGRAM_TABLE = ({n: k for k, (n, _) in enumerate(tab)}, dict(tab)) if ok else None
    return _PROGRAM_TABLE


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
