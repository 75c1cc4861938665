"""The step arena (engine._Arena / _Run) and the collector settling, without a GPU: what the round-4 headline fix rests on.

  * a lease bumps a pointer through chunks, hands out 256-byte-aligned views, and a step that needed several chunks leaves ONE
    chunk with headroom -- the next step of the same size makes no allocator call at all;
  * what outlives a step never comes from the arena: cached weight images (`_Run.cached`), outputs and gradients
    (`_Run.out_empty`);
  * two steps in flight at once (forward, forward, backward, backward) get two arenas; a dropped context hands its lease back;
  * a whole validate-only training step (every host path of engine._forward / _backward, kernels off) leaves the arena idle,
    its outputs and gradients outside it, and a second step of the same shape grows nothing;
  * engine.settle_gc(): one collection, then the survivors are frozen out of later collections.
"""
import gc
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd import engine, native
from tacotron2_amd.hparams import create_hparams


def _inside(t, arena):
    a = t.data_ptr()
    return any(c.data_ptr() <= a < c.data_ptr() + c.numel() for c in arena.chunks)


def test_arena_bump_alignment_growth_and_consolidation(monkeypatch):
    monkeypatch.setattr(engine, 'ARENA_CHUNK', 1 << 16)
    a = engine._Arena(torch.device('cpu'))
    v1 = a.alloc(100)
    v2 = a.alloc(300)
    assert v1.numel() == 100 and v2.numel() == 300
    assert (v2.data_ptr() - v1.data_ptr()) == 256 and v1.data_ptr() % 64 == 0
    big = a.alloc(3 << 16)                       # larger than a chunk: gets a chunk of its own
    assert big.numel() == 3 << 16 and len(a.chunks) == 2
    a.alloc(60000)                               # does not fit behind `big`: a third chunk
    assert len(a.chunks) == 3 and a.growths == 3
    used = a.used
    a.busy = True
    a.release()
    assert not a.busy and len(a.chunks) == 3 and a.peak == used       # consolidation waits for the next lease
    a.consolidate()
    assert len(a.chunks) == 1
    assert a.capacity() >= int(used * engine.ARENA_HEADROOM) and a.capacity() % (1 << 28) == 0
    g = a.growths
    for _ in range(3):                           # the same step again: no allocator call
        a.busy = True
        a.alloc(100); a.alloc(300); a.alloc(3 << 16); a.alloc(60000)
        a.release()
        a.consolidate()
    assert a.growths == g and len(a.chunks) == 1
    # an empty shape is served like torch.empty serves it (a 1-byte arena view cannot be viewed as float32)
    run = engine._Run(torch.device('cpu'), 'fp32', {}, a)
    z = run.empty(0, 5)
    assert z.shape == (0, 5) and z.dtype == torch.float32


def test_run_keeps_what_outlives_the_step_out_of_the_arena():
    dev = torch.device('cpu')
    a = engine._arena_acquire(dev)
    try:
        run = engine._Run(dev, 'bf16', {}, a)
        x = run.empty(3, 5)
        h = run.empty16(7)
        m = run.empty8(9)
        i = run.empty_i32(2)
        assert x.dtype == torch.float32 and x.shape == (3, 5) and h.dtype == torch.bfloat16 and m.dtype == torch.uint8 and i.dtype == torch.int32
        assert all(_inside(t, a) for t in (x, h, m, i))
        x.fill_(1.0); h.fill_(2.0)               # views are writable and do not overlap
        assert float(x.sum()) == 15.0 and float(h.float().sum()) == 14.0
        w = torch.nn.Parameter(torch.ones(4))
        img = run.cached('image', [w], lambda: run.empty16(4))          # what run.cast16 allocates for a weight image
        assert not _inside(img, a) and run.cached('image', [w], lambda: 1 / 0) is img
        assert not _inside(run.out_empty(4, 4), a)
        assert _inside(run.ws(8), a)
    finally:
        engine._arena_release(a)
    assert not a.busy


def test_overlapping_steps_get_their_own_arena_and_dropped_contexts_give_theirs_back():
    dev = torch.device('cpu')
    c1, c2 = engine._Ctx(), engine._Ctx()
    c1.arena = engine._arena_acquire(dev)
    c2.arena = engine._arena_acquire(dev)
    assert c1.arena is not c2.arena and c1.arena.busy and c2.arena.busy
    first = c1.arena
    c1.release()
    assert not first.busy and c1.arena is None
    c3 = engine._Ctx()
    c3.arena = engine._arena_acquire(dev)
    assert c3.arena is first                     # reused as it is
    second = c2.arena
    del c2                                       # a forward whose graph is dropped without a backward
    gc.collect()
    assert not second.busy
    c3.release()
    c3.release()                                 # idempotent


def test_validate_only_training_step_runs_in_the_arena(native_lib):
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.loss_function import Tacotron2Loss
    native.set_validate_only(True)
    try:
        hp = create_hparams(gu.TINY_HP)
        m = Tacotron2(hp).train()
        m.precision = 'bf16'
        batch = gu.make_train_batch([12, 9, 5], [20, 16, 11], hp.n_mel_channels, 1)
        stats = []
        for _ in range(3):
            m.zero_grad()
            x, y = m.parse_batch(batch)
            out = m(x)
            pool = [a for pool in engine._ARENAS.values() for a in pool]
            busy = [a for a in pool if a.busy]
            assert len(busy) == 1                # the forward's lease, held for the backward
            assert not any(_inside(o, busy[0]) for o in out)
            Tacotron2Loss()(out, y).backward()
            assert not busy[0].busy
            assert all(p.grad is not None and not _inside(p.grad, busy[0]) for p in m.parameters())
            stats.append((busy[0].growths, len(busy[0].chunks)))
        assert stats[1] == stats[2] and stats[2][1] == 1          # steady state: one chunk, no further allocator call
        # an eval-mode forward (validation) gives its lease back at once; so does a forward under no_grad
        m.eval()
        m(m.parse_batch(batch)[0])
        assert not any(a.busy for pool in engine._ARENAS.values() for a in pool)
        m.train()
        with torch.no_grad():
            m(m.parse_batch(batch)[0])
        gc.collect()
        assert not any(a.busy for pool in engine._ARENAS.values() for a in pool)
    finally:
        native.set_validate_only(False)


def test_arena_can_be_switched_off(monkeypatch):
    monkeypatch.setattr(engine, 'ARENA', False)
    assert engine._arena_acquire(torch.device('cpu')) is None
    run = engine._Run(torch.device('cpu'), 'fp32', {}, None)
    assert run.empty(2, 2).shape == (2, 2)       # plain torch.empty


def test_settle_gc_freezes_once():
    was = dict(engine._gc_state)
    frozen_before = gc.get_freeze_count()
    try:
        engine._gc_state['frozen'] = False
        assert engine.settle_gc(quiet=True) is True
        assert gc.get_freeze_count() > frozen_before
        n = gc.get_freeze_count()
        assert engine.settle_gc(quiet=True) is False and gc.get_freeze_count() == n       # only once per process
        assert engine.GC_FREEZE is (os.environ.get('T2AMD_GC_FREEZE', '0') == '1')    # implicit form is opt-in
    finally:
        gc.unfreeze()
        engine._gc_state.update(was)


def test_demoted_forms_are_reselected_after_clean_steps(native_lib, monkeypatch):
    """a give-up demotes the process to the chains; after TRAIN_FWD_REPROMOTE_AFTER clean training steps
    what was selected before is selected again, and a repeated give-up doubles the interval.  (Host logic only: the counters
    the kernels raise are stood in for.)"""
    said = []
    state = dict(engine._DEMOTION)
    flags = (engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT)
    fold = native.get_bptt_cell_fold()
    monkeypatch.setattr(engine, 'TRAIN_FWD_REPROMOTE_AFTER', 3)
    pending = {'attn': 2, 'enc': 0}
    monkeypatch.setattr(native, 'attn_handoff_timeouts', lambda reset=True: pending.pop('attn', 0) if reset else pending.get('attn', 0))
    monkeypatch.setattr(native, 'encoder_handoff_timeouts', lambda reset=True: pending.pop('enc', 0) if reset else pending.get('enc', 0))
    try:
        engine._DEMOTION.update(active=False, count=0, clean=0, need=0, saved=None, repromotions=0, explicit=False, probation=0, given_up=False)
        engine.TRAIN_FWD_PERSISTENT = True
        native.set_attn_fwd_fused(-1); native.set_attn_bwd_fused(-1); native.set_bptt_cell_fold(1)
        assert engine.handle_nonfinite_step(log=said.append) == 2
        assert engine.TRAIN_FWD_PERSISTENT is False
        assert native.get_attn_fwd_fused() == 0 and native.get_attn_bwd_fused() == 0 and native.get_bptt_cell_fold() == 0
        assert engine.give_up_counters()['demoted_now'] and 're-selected after 3 clean steps' in said[-1]
        assert [engine._note_training_step(said.append) for _ in range(4)] == [False, False, False, True]
        assert engine.TRAIN_FWD_PERSISTENT is True
        assert native.get_attn_fwd_fused() == -1 and native.get_attn_bwd_fused() == -1 and native.get_bptt_cell_fold() == 1
        assert not engine.give_up_counters()['demoted_now'] and engine.give_up_counters()['repromotions'] == 1
        assert engine._note_training_step(said.append) is False            # nothing to do while not demoted
        pending['attn'] = 1                                                  # it gives up again: the interval doubles
        assert engine.handle_nonfinite_step(log=said.append) == 1 and engine._DEMOTION['need'] == 6
        assert engine.handle_nonfinite_step(log=said.append) == 0          # a non-finite step with another cause changes nothing
        assert engine._DEMOTION['need'] == 6 and engine._DEMOTION['count'] == 2
        monkeypatch.setattr(engine, 'TRAIN_FWD_REPROMOTE_AFTER', 0)         # 0 = never
        assert not any(engine._note_training_step(said.append) for _ in range(20))
    finally:
        engine._DEMOTION.clear(); engine._DEMOTION.update(state)
        engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT = flags
        native.set_attn_fwd_fused(-1); native.set_attn_bwd_fused(-1); native.set_bptt_cell_fold(fold)
