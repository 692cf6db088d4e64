"""CPU check of the recorded-step machinery of voicemap_amd/engine.py (no GPU: the C ABI is a stub that logs what it is called with):
a recorded list of C-ABI calls, event records and waits is replayed in order, with only the marked per-step arguments patched, and
every event key mapped onto one event of the program's own.  The GPU side -- a replayed training step is bit-identical to the eager
one -- is tests/test_gpu_replay.py."""
import ctypes

from voicemap_amd import engine as E


class _StubLib:
    def __init__(self):
        self.log, self.cdll, self.tuning_epoch, self._next_event = [], self, 0, 1000

    def call(self, name, *args):
        rc = getattr(self, name)(*args)
        assert rc == 0
        return rc

    def __getattr__(self, name):
        if not name.startswith("vm_"):
            raise AttributeError(name)

        def fn(*args):
            if name == "vm_event_create":
                self._next_event += 1
                args[0]._obj.value = self._next_event
            self.log.append((name,) + tuple(a for a in args if not hasattr(a, "_obj")))
            return 0
        return fn


def _bare_engine():
    eng = E.HipEncoderEngine.__new__(E.HipEncoderEngine)      # no GPU, no __init__: only the funnel under test
    eng.lib, eng.timed, eng._stream_stack = _StubLib(), {}, [7]
    return eng


def test_recorded_calls_are_replayed_in_order_with_only_the_dynamic_slots_patched():
    eng = _bare_engine()
    eng._rec = E._Program()
    assert eng._dyn("y", None) is None                         # a NULL argument is part of the configuration, not a slot
    eng._call("vm_first", 11, eng._dyn("y", 123), 5, eng._dyn("lr_t", 0.5))
    eng._rec.cmds.append([1, "ev-a", 7])                        # record(ev-a) on stream 7 ...
    eng._rec.cmds.append([2, 9, "ev-a"])                        # ... stream 9 waits for it
    eng._call("vm_second", eng._dyn(("drop", 2, 64), 4096), None)
    eng._rec.cmds.append([1, ("join", 4), 9])
    eng._rec.cmds.append([2, 7, ("join", 4)])
    rec, eng._rec = eng._rec, None
    assert [(c, a, k) for c, a, k in rec.patches] == [(0, 1, "y"), (0, 3, "lr_t"), (3, 0, ("drop", 2, 64))]
    assert type(rec.cmds[0][2][1]) is int and type(rec.cmds[0][2][3]) is float      # plain numbers are stored, not the markers
    eager = list(eng.lib.log)
    assert eager == [("vm_first", 11, 123, 5, 0.5), ("vm_second", 4096, None)]
    prog = eng._finish_program(rec)
    assert len(prog.events) == 2 and len(set(prog.events.values())) == 2
    eng.lib.log.clear()
    eng._run_program(prog, {"y": 999, "lr_t": 0.25, ("drop", 2, 64): 8192, "unused": 1})
    ea, ej = prog.events["ev-a"], prog.events[("join", 4)]
    assert eng.lib.log == [("vm_first", 11, 999, 5, 0.25), ("vm_event_record", ea, 7), ("vm_stream_wait_event", 9, ea),
                           ("vm_second", 8192, None), ("vm_event_record", ej, 9), ("vm_stream_wait_event", 7, ej)]
    eng._run_program(prog, {"y": 1, "lr_t": 2.0, ("drop", 2, 64): 3})               # a second replay patches the same slots again
    assert eng.lib.log[-6] == ("vm_first", 11, 1, 5, 2.0) and eng.lib.log[-3] == ("vm_second", 3, None)


def test_outside_a_recording_the_markers_are_plain_values_and_nothing_is_logged():
    eng = _bare_engine()
    assert eng._rec is None and eng._dyn("y", 5) == 5 and type(eng._dyn("y", 5)) is int
    eng._call("vm_only", 1, 2)
    assert eng.lib.log == [("vm_only", 1, 2)]


def test_a_failing_call_in_a_replay_raises_with_its_name():
    import pytest
    from voicemap_amd import _lib
    eng = _bare_engine()
    eng._rec = E._Program()
    eng._call("vm_ok", 1)
    prog = eng._finish_program(eng._rec)
    eng._rec = None
    prog.cmds[0][1] = lambda *a: -1
    eng.lib.vm_last_error = lambda: b"boom"
    with pytest.raises(_lib.VoicemapHipError, match="vm_ok.*boom"):
        eng._run_program(prog, {})


def test_host_calls_keep_their_place_in_a_recorded_step():
    """Round 6: a step's host-side calls that are not C-ABI entry points (the two gradient collectives of data parallelism,
    voicemap_amd/parallel.py) are slots of the program: run when recorded and again, at the same position, in every replay."""
    eng = _bare_engine()
    eng._rec = E._Program()
    seen = []
    eng._call("vm_before", 1)
    eng._host_call(lambda: seen.append(len(eng.lib.log)))
    eng._call("vm_after", 2)
    rec, eng._rec = eng._rec, None
    assert seen == [1] and [c[0] for c in rec.cmds] == [0, 3, 0]
    prog = eng._finish_program(rec)
    assert prog.events == {}                                   # a host call is not an event key
    eng.lib.log.clear()
    eng._run_program(prog, {})
    assert seen == [1, 1] and eng.lib.log == [("vm_before", 1), ("vm_after", 2)]
    # outside a recording the call simply runs
    eng._host_call(lambda: seen.append(-1))
    assert seen[-1] == -1
