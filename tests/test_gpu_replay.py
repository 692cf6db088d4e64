"""-m gpu: a REPLAYED training step is the eager step.  engine._train_step records the enqueue sequence of a configuration the
second time it sees it (C-ABI calls with their arguments, event records and waits between the main, tower and side streams) and
replays that list from the third step on, patching only the per-step values (input / label / mask pointers, the BatchNorm
zero-debias factor, the loss scale, Adam's lr_t).  The reference's train_on_batch (experiments/train_siamese.py:65-94,
siamese_contrastive_loss.py:70-99, train_classifier.py:117-127) is the same computation every step, so two engines fed the same
batches -- one with replay off -- must hold bit-identical weights, optimizer slots, moving statistics and losses after every step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BLOCKS = [(32, 16, 4), (3, 32, 2), (3, 48, 2), (3, 64, 2)]
DEFAULT_FUSE = 17  # the library's default vm_set_tuning("fuse_finalize") mask


def _engines(head, dtype, dropout, **kw):
    from voicemap_amd.engine import HipEncoderEngine
    a = HipEncoderEngine(BLOCKS, 32, dropout=dropout, head=head, dtype=dtype, seed=5, **kw)
    b = HipEncoderEngine(BLOCKS, 32, dropout=dropout, head=head, dtype=dtype, seed=5, **kw)
    b.replay = False
    return a, b


def _same_state(a, b, what):
    for nm in ("P", "M", "V", "NT", "ZD", "G"):
        u, v = getattr(a, nm), getattr(b, nm)
        assert torch.equal(u.view(torch.int32), v.view(torch.int32)), (what, nm)      # bit patterns: an overflowed f16 step leaves NaNs in G
    assert a.iterations == b.iterations and a.bn_steps == b.bn_steps and a.loss_scale == b.loss_scale, what


def _programs(eng):
    from voicemap_amd.engine import _Program
    return [p for p in eng._programs.values() if isinstance(p, _Program)]


@pytest.mark.parametrize("dtype,dropout,loss", [("f16", 0.0, "contrastive"), ("f16", 0.05, "bce"), ("bf16", 0.0, "bce"), ("f32", 0.05, "contrastive")])
def test_replayed_siamese_steps_are_the_eager_steps(dtype, dropout, loss):
    a, b = _engines("uniform_euclidean", dtype, dropout)
    r = np.random.default_rng(3)
    pairs, raw_len = 6, 4800
    for step in range(7):
        x1 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        x2 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        y = (r.random((pairs, 1)) > 0.5).astype(np.float32)
        out = []
        for eng in (a, b):
            pl = eng.siamese_train_step(x1, x2, y, loss=loss, preprocessed=False, downsampling=4)     # masks: the engine's own generator
            torch.cuda.synchronize()
            out.append((pl["loss_acc"].clone(), pl["emb"].clone(), pl["pred"].clone()))
        for u, v in zip(*out):
            assert torch.equal(u, v), step
        _same_state(a, b, step)
    progs = _programs(a)
    assert len(progs) == 1 and not _programs(b)
    # what a program holds: the step's C-ABI calls and its stream ordering, and only a handful of patched slots
    kinds = [c[0] for c in progs[0].cmds]
    assert kinds.count(0) > 30 and kinds.count(1) >= 3 and kinds.count(2) >= 3
    assert {k if not isinstance(k, tuple) else k[0] for _, _, k in progs[0].patches} <= {"raw", "y", "loss_scale", "zc", "lr_t", "gpre", "drop", "dropb"}
    # ... and (round 6) the same list as int64 words for the library's own runner, which is what replayed it above
    assert a.native_replay and progs[0].native is not None
    segs, slots = progs[0].native
    assert len(segs) == 1 and len(slots) == len(progs[0].patches) and segs[0].dtype == np.int64


@pytest.mark.parametrize("dtype,dropout", [("f16", 0.05), ("f32", 0.0)])
def test_the_native_runner_and_the_python_loop_replay_the_same_steps(dtype, dropout):
    """vm_program_run (the recorded step as int64 words, called once per step) against the Python loop of ctypes calls it replaces:
    the same launches with the same arguments -- bit-identical state after every step, float and double slots (lr_t, the zero-debias
    factor, the loss scale) included."""
    from voicemap_amd.engine import HipEncoderEngine
    a = HipEncoderEngine(BLOCKS, 32, dropout=dropout, head="weighted_l1", dtype=dtype, seed=11)
    b = HipEncoderEngine(BLOCKS, 32, dropout=dropout, head="weighted_l1", dtype=dtype, seed=11)
    b.native_replay = False
    r = np.random.default_rng(4)
    pairs, raw_len = 6, 4800
    for step in range(8):
        x1 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        x2 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        y = (r.random((pairs, 1)) > 0.5).astype(np.float32)
        for eng in (a, b):
            eng.siamese_train_step(x1, x2, y, loss="bce", preprocessed=False, downsampling=4)
        torch.cuda.synchronize()
        _same_state(a, b, step)
    assert _programs(a)[0].native is not None and len(_programs(b)) == 1


def test_replay_follows_the_configuration():
    """A different loss, batch size, an injected mask set, apply_update=False or a flipped engine switch is a different program (or
    the eager path), never a stale replay."""
    a, b = _engines("weighted_l1", "f16", 0.0)
    r = np.random.default_rng(4)

    def batch(pairs):
        return (r.normal(0, 0.05, (pairs, 4800, 1)).astype(np.float32), r.normal(0, 0.05, (pairs, 4800, 1)).astype(np.float32),
                (r.random((pairs, 1)) > 0.5).astype(np.float32))
    seq = [(4, "bce", True)] * 4 + [(4, "contrastive", True)] * 3 + [(6, "bce", True)] * 3 + [(4, "bce", False)] * 3 + [(4, "bce", True)] * 2
    for k, (pairs, loss, upd) in enumerate(seq):
        x1, x2, y = batch(pairs)
        if k == 14:
            a.overlap_wgrad = b.overlap_wgrad = False       # a switch that changes the stream structure
        for eng in (a, b):
            eng.siamese_train_step(x1, x2, y, loss=loss, preprocessed=False, apply_update=upd)
        torch.cuda.synchronize()
        _same_state(a, b, k)
    assert len(_programs(a)) == 4


def test_replayed_steps_from_a_resident_corpus_and_the_classifier():
    from voicemap_amd.engine import HipEncoderEngine
    a, b = _engines("uniform_euclidean", "f16", 0.05)
    g = torch.Generator(device="cuda").manual_seed(1)
    audio = (torch.randn(400000, device="cuda", generator=g) * 0.05 * 32767).clamp(-32767, 32767).to(torch.int16)
    r = np.random.default_rng(6)
    for step in range(6):
        o1, o2 = torch.as_tensor(r.integers(0, 390000, 5)), torch.as_tensor(r.integers(0, 390000, 5))
        y = (r.random(5) > 0.5).astype(np.float32)
        for eng in (a, b):
            eng.siamese_train_step_from_offsets(audio, o1, o2, y, raw_len=4800, loss="bce")
        torch.cuda.synchronize()
        _same_state(a, b, step)
    assert len(_programs(a)) == 1
    c = HipEncoderEngine(BLOCKS, 32, dropout=0.05, head="classifier", num_classes=10, dtype="f16", seed=2)
    d = HipEncoderEngine(BLOCKS, 32, dropout=0.05, head="classifier", num_classes=10, dtype="f16", seed=2)
    d.replay = False
    for step in range(6):
        x = r.normal(0, 0.05, (8, 4800, 1)).astype(np.float32)
        lab = r.integers(0, 10, 8)
        outs = []
        for eng in (c, d):
            pl = eng.classifier_train_step(x, lab, preprocessed=False)
            torch.cuda.synchronize()
            outs.append(pl["loss_acc"].clone())
        assert torch.equal(*outs), step
        _same_state(c, d, step)
    assert len(_programs(c)) == 1


def test_a_replayed_f16_run_keeps_its_loss_scale_logic():
    """The loss scale is patched into every replay: force a skip (a scale that overflows) and watch both engines halve it at the
    same step and end with the same weights."""
    a, b = _engines("uniform_euclidean", "f16", 0.0)
    for eng in (a, b):
        eng.scale_poll_every, eng.scale_poll_lag = 2, 1
    r = np.random.default_rng(8)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for step in range(12):
            if step == 5:
                a.loss_scale = b.loss_scale = 2.0 ** 40
            x1 = r.normal(0, 0.05, (4, 4800, 1)).astype(np.float32)
            x2 = r.normal(0, 0.05, (4, 4800, 1)).astype(np.float32)
            y = (r.random((4, 1)) > 0.5).astype(np.float32)
            for eng in (a, b):
                eng.siamese_train_step(x1, x2, y, loss="contrastive", preprocessed=False)
            torch.cuda.synchronize()
            _same_state(a, b, step)
    assert a.skipped_steps() == b.skipped_steps() >= 1 and a.loss_scale < 2.0 ** 40


@pytest.mark.parametrize("dtype,dropout,filters", [("f16", 0.0, 128), ("f16", 0.05, 32), ("bf16", 0.0, 64), ("f32", 0.05, 16)])
def test_last_arriver_finalize_equals_the_two_launch_form(dtype, dropout, filters):
    """vm_set_tuning("fuse_finalize", 1): the two-stage column reductions (BatchNorm statistics, BatchNorm-backward sums, bias-gradient
    column sums, the folded weight gradient's tap sums) finish in the stage-1 launch -- the last workgroup of a channel block to arrive
    runs the finalize body.  Same partials, same butterfly: training with it on and off must agree bit for bit (folded path at
    filters 128 / 64 with dropout 0, the unfolded passes with dropout)."""
    from voicemap_amd import _lib
    from voicemap_amd.engine import HipEncoderEngine
    blocks = [(32, filters, 4), (3, 2 * filters, 2), (3, 3 * filters, 2), (3, 4 * filters, 2)]
    lib = _lib.lib()
    r = np.random.default_rng(12)
    pairs, raw_len = 4, 9600
    batches = [(r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32), r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32),
                (r.random((pairs, 1)) > 0.5).astype(np.float32)) for _ in range(5)]
    states = []
    try:
        for fuse in (0, 15):
            lib.call("vm_set_tuning", b"fuse_finalize", fuse)
            eng = HipEncoderEngine(blocks, 32, dropout=dropout, head="uniform_euclidean", dtype=dtype, seed=9)
            outs = []
            for x1, x2, y in batches:
                pl = eng.siamese_train_step(x1, x2, y, loss="bce", preprocessed=False)
                torch.cuda.synchronize()
                outs.append(pl["loss_acc"].clone())
            states.append((eng, outs))
    finally:
        lib.call("vm_set_tuning", b"fuse_finalize", DEFAULT_FUSE)
    (a, oa), (b, ob) = states
    for u, v in zip(oa, ob):
        assert torch.equal(u, v)
    _same_state(a, b, "fuse_finalize")


def test_inference_between_replayed_steps_sees_the_current_weights():
    """a replayed optimizer step must mark the packed inference weights stale like the eager one does -- the ordinary
    train / evaluate / train / evaluate loop of experiments/train_siamese.py:65-94 (validation and the n-shot callback per epoch)."""
    a, b = _engines("uniform_euclidean", "f16", 0.0)
    assert a.packed_weights
    r = np.random.default_rng(11)
    pairs, raw_len = 6, 4800
    probe = r.normal(0, 0.05, (5, raw_len // 4)).astype(np.float32)
    seen = []
    for step in range(12):
        x1 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        x2 = r.normal(0, 0.05, (pairs, raw_len, 1)).astype(np.float32)
        y = (r.random((pairs, 1)) > 0.5).astype(np.float32)
        for eng in (a, b):
            eng.siamese_train_step(x1, x2, y, loss="contrastive", preprocessed=False, downsampling=4)
        if step in (3, 6, 7, 11):      # the first embed clears the stale flag; the later ones follow REPLAYED steps
            ea, eb = a.embed(probe).clone(), b.embed(probe).clone()
            torch.cuda.synchronize()
            assert torch.equal(ea, eb), step
            seen.append(ea)
    assert len(_programs(a)) == 1
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[2], seen[3])   # the weights did move in between


def test_recorded_steps_are_dropped_when_their_buffers_go():
    """programs hold raw device pointers; toggling packed_weights (bench.py --tune, tests) re-makes the fold buffers and a
    plan used with another tower size re-makes its fold plan -- neither may leave a recorded step behind."""
    a, b = _engines("uniform_euclidean", "f16", 0.0)
    r = np.random.default_rng(12)

    def steps(pairs, k):
        for _ in range(k):
            x1 = r.normal(0, 0.05, (pairs, 4800, 1)).astype(np.float32)
            x2 = r.normal(0, 0.05, (pairs, 4800, 1)).astype(np.float32)
            y = (r.random((pairs, 1)) > 0.5).astype(np.float32)
            for eng in (a, b):
                eng.siamese_train_step(x1, x2, y, loss="bce", preprocessed=False)
        torch.cuda.synchronize()
    steps(4, 4)
    assert len(_programs(a)) == 1
    for v in (False, True):
        a.packed_weights = b.packed_weights = v
        assert not a._programs
        steps(4, 4)
        _same_state(a, b, v)


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_preprocessing_beside_the_previous_steps_tail_is_the_same_step(dtype):
    """Round 6 (engine.pre_overlap): with input_ready the decimate / whiten pass of step k + 1 runs on the tower stream as soon as
    step k's block-1 backward has read x0 (voicemap/utils.py:22-34, 88-101 is the same function of the batch wherever it runs).  A
    different batch every step: a preprocessing that overwrote x0 too early, or a forward that started too early, changes the weights.
    Eager, recorded and replayed steps follow one another on one plan."""
    a, b = _engines("uniform_euclidean", dtype, 0.0)
    b.replay = True
    b.pre_overlap = False
    r = np.random.default_rng(21)
    pairs, raw_len = 16, 12000
    raws = [torch.from_numpy(r.normal(0, 0.05, (2 * pairs, raw_len)).astype(np.float32)).cuda() for _ in range(4)]
    y = torch.from_numpy((r.random(pairs) > 0.5).astype(np.float32)).cuda()
    torch.cuda.synchronize()
    pls = [e.plan(2 * pairs, raw_len // 4, True) for e in (a, b)]
    for step in range(14):
        if step == 9:
            a.replay = False        # back to eager steps behind replayed ones
        for e, pl in zip((a, b), pls):
            e.train_step_resident(pl, pairs, y, "contrastive" if step % 5 else "bce", raw=raws[step % 4], drop_masks=None, input_ready=True)
        if step % 3 == 2:
            torch.cuda.synchronize()
            _same_state(a, b, step)
    torch.cuda.synchronize()
    _same_state(a, b, "end")
    assert _programs(a) and "x0_free_ev" in pls[0] and "x0_free_ev" not in pls[1]


def test_staged_offsets_on_the_tower_stream_are_the_same_steps():
    """The resident-corpus loop (shards.py; reference: LibriSpeechDataset.__getitem__ crops on the host, voicemap/librispeech.py:103-137):
    offsets and labels go up on the tower stream in front of the preprocessing that runs there."""
    a, b = _engines("uniform_euclidean", "f16", 0.0)
    b.replay = True
    b.pre_overlap = False
    r = np.random.default_rng(22)
    pairs, raw_len = 8, 12000
    audio = torch.from_numpy((r.normal(0, 0.05, 400000) * 32767).astype(np.int16)).cuda()
    for step in range(40):          # more steps than staging slots: a slot is refilled
        o1 = r.integers(0, 400000 - raw_len, pairs).astype(np.int64)
        o2 = r.integers(0, 400000 - raw_len, pairs).astype(np.int64)
        y = (r.random(pairs) > 0.5).astype(np.float32)
        for e in (a, b):
            e.siamese_train_step_from_offsets(audio, o1, o2, y, raw_len, loss="contrastive", drop_masks=None)
        if step % 13 == 12:
            torch.cuda.synchronize()
            _same_state(a, b, step)
    torch.cuda.synchronize()
    _same_state(a, b, "end")


def test_engines_of_a_process_share_their_streams():
    """One tower stream per device for the whole process, and the weight-gradient chain rides it (round 6): private streams per engine
    put the 4th and 6th engine of a process on shared hardware queues (+ 7.5 % on their step, profiles/r06_stream_aliasing.txt), and one
    stream beside the main one beat two at every batch size (profiles/r06_stream_merge.txt).  A priority experiment gets a private
    stream and 0 returns to the shared one."""
    from voicemap_amd.engine import HipEncoderEngine
    a = HipEncoderEngine(BLOCKS, 32, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=1)
    b = HipEncoderEngine(BLOCKS, 32, dropout=0.0, head=None, dtype="bf16", seed=2)
    assert a.tower_stream is b.tower_stream and a.misc_stream is b.misc_stream
    assert a.side_stream is a.tower_stream and b.side_stream is a.tower_stream
    a.side_priority = -1
    assert a.side_stream is not a.tower_stream and b.side_stream is b.tower_stream
    a.side_priority = 0
    assert a.side_stream is a.tower_stream
