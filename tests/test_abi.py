"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly what include/*.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os

from voicemap_amd import _lib


def test_header_and_binding_table_agree():
    declared = _lib.header_functions()
    assert declared, "no functions parsed from include/voicemap_hip.h"
    assert sorted(_lib.SIGNATURES) == declared


def test_library_builds_loads_and_exports_every_symbol():
    from voicemap_amd import build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    cdll = ctypes.CDLL(path)
    for name in _lib.header_functions():
        assert hasattr(cdll, name), name
    lib = _lib.lib()
    assert lib.abi == _lib.ABI_VERSION == 10
    assert lib.query("vm_bn_part_rows") > 0
    assert lib.query("vm_conv_stat_rows", 3000) == 24
    assert lib.query("vm_conv_wgrad_splits", 256, 3000, 128, 256) >= 1


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.lib()
    try:
        lib.call("vm_conv_fwd", None, None, None, 1, 1, 8, 8, 0, None, None, None, None)
    except _lib.VoicemapHipError as e:
        assert "null pointer" in str(e)
    else:
        raise AssertionError("expected VoicemapHipError")


def test_engine_refuses_to_run_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from voicemap_amd.engine import HipEncoderEngine
    with pytest.raises(RuntimeError):
        HipEncoderEngine([(32, 16, 4), (3, 32, 2)], 8)


def test_tuning_table_is_small_and_rejects_unknown_keys():
    """vm_set_tuning only selects among kernels that compute the same result (six keys, include/voicemap_hip.h); the experiment
    switches of rounds 1-2 (ablations, ring / phase / skew variants) are gone from the shipped library.  Host-only call: no GPU."""
    from voicemap_amd import _lib
    lib = _lib.lib()
    for key, good, bad in ((b"nt_n2", 3, 4), (b"nt_glds", 1, 2), (b"tn_x", 1, 2), (b"tn_tile", 256, 64), (b"f1_blocks", 1024, 0),
                           (b"f1_fwd_blocks", 1024, -1), (b"apply_order", 2, 3), (b"f1_products", 2, 0)):   # (good = the defaults)
        assert lib.cdll.vm_set_tuning(key, bad) != 0, key
        assert lib.cdll.vm_set_tuning(key, good) == 0, key
    for gone in (b"nt_ablate", b"nt_ring", b"nt_p8", b"nt_w4", b"nt_n2r", b"gemm_kb", b"no_such_knob"):
        assert lib.cdll.vm_set_tuning(gone, 1) != 0, gone
    assert lib.cdll.vm_set_tuning(None, 1) != 0
