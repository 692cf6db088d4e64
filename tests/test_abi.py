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
    assert lib.abi == 1
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


def test_shipped_build_refuses_result_changing_tuning():
    """vm_set_tuning only selects among kernels that compute the same result; the ablation switches (wrong results, timing
    experiments) exist only in -DVM_ENABLE_ABLATION builds (include/voicemap_hip.h).  Host-only call: no GPU needed."""
    from voicemap_amd import _lib
    lib = _lib.lib()
    assert lib.cdll.vm_set_tuning(b"nt_ablate", 0) == 0
    assert lib.cdll.vm_set_tuning(b"nt_ablate", 2) == -3 and b"VM_ENABLE_ABLATION" in lib.cdll.vm_last_error()
    assert lib.cdll.vm_set_tuning(b"no_such_knob", 1) != 0
