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
    assert lib.abi == _lib.ABI_VERSION == 11
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


def test_program_runner_is_generated_from_the_binding_table_and_dispatches():
    """csrc/program_run.hip is what tools/gen_program_run.py prints for the current binding table; the loaded library reports that
    table's hash; and vm_program_run walks a word list in order, stopping at the first non-zero return code with its position --
    checked with entry points that need no GPU (a `_supported` query that answers 0, then vm_bn_part_rows, whose non-zero answer is
    taken for a failure code)."""
    import ctypes
    import importlib.util
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_program_run", os.path.join(root, "tools", "gen_program_run.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(root, "voicemap_amd", "csrc", "program_run.hip")) as f:
        assert f.read() == gen.render(), "run python tools/gen_program_run.py"
    tab = _lib.program_table()
    assert tab is not None, "the library's vm_program_table_hash() is not the binding's"
    ids, sigs = tab
    assert sigs["vm_event_record"] == "PP" and sigs["vm_adam_clip_step"].count("F") >= 5
    lib = _lib.lib()
    rows = lib.cdll.vm_bn_part_rows()
    assert rows > 0
    words = np.array([ids["vm_conv_fwd_fold_supported"], 6, 1, 7, 3, 5, 1, 0,      # an odd shape: not served -> 0 -> the run goes on
                      ids["vm_bn_part_rows"], 0,
                      ids["vm_abi_version"], 0], dtype=np.int64)
    fail = ctypes.c_int64(-1)
    rc = lib.cdll.vm_program_run(words.ctypes.data, words.size, ctypes.byref(fail))
    assert rc == rows and fail.value == 8
    # a wrong argument count and an unknown id are refused at their position
    bad = np.array([ids["vm_bn_part_rows"], 1, 0], dtype=np.int64)
    assert lib.cdll.vm_program_run(bad.ctypes.data, bad.size, ctypes.byref(fail)) == -1 and fail.value == 0
    bad = np.array([10 ** 6, 0], dtype=np.int64)
    assert lib.cdll.vm_program_run(bad.ctypes.data, bad.size, ctypes.byref(fail)) == -1 and fail.value == 0
