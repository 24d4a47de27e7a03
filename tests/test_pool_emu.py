"""CPU tests of the decode pool (csrc/engine.hip: concurrent generate() calls share their decode steps) through the
emulator: requests of different batch sizes, prompt lengths, EOS / stop / sampling parameters in flight together get
exactly the ids their own session loop produces."""
import threading

import numpy as np
import pytest

import e2e_cases
import kernel_cases as kc


@pytest.fixture(scope="module")
def emu_lib():
    return kc.EmuBackend().lib


def session_loop_ids(eng, ids, imgs, segs, deps, n_new):
    """greedy ids from the session's own loop (vc_prefill + vc_decode_step): never pooled"""
    last, _, _ = eng.prefill(ids, imgs, segs, deps, reserve=n_new)
    toks = [np.argmax(last, -1).astype(np.int32)]
    for _ in range(n_new - 1):
        _, nxt = eng.decode_step(toks[-1], want_logits=False)
        toks.append(nxt)
    return np.stack(toks, 1)


def test_concurrent_requests_share_steps_and_keep_their_ids(emu_lib):
    names = ["ds_img_depth_seg", "ds_img_only", "ds_img_seg"]            # B = 2 / 1 / 1, three spliced lengths
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    cases, refs = [], []
    for n in names:
        g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs(n)
        cases.append((ids, imgs, segs, deps))
        refs.append(session_loop_ids(root, ids, imgs, segs, deps, 6))
    sessions = [root, root.fork(), root.fork()]
    outs = [[None] * 3 for _ in sessions]
    errs = []
    # (a pool left behind by an earlier test in another mode — split, profiling — is rebuilt by the first call, which restarts its
    # histogram: one warm call first)
    root.generate_greedy(*cases[1], max_new_tokens=2)
    steps0 = root.pool_step_counts()

    def work(si):
        try:
            for j in range(3):
                ci = (si + j) % 3                                          # every session meets every case
                outs[si][j] = (ci, sessions[si].generate_greedy(*cases[ci], max_new_tokens=6))
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for si in range(3):
        for ci, got in outs[si]:
            assert np.array_equal(got, refs[ci]), f"session {si} case {names[ci]}: pooled ids differ from the session loop"
    # the pool's step histogram (what bench.py weights its kernel timings by): nine requests of 5 cached steps each ran in
    # at most 45 steps (fewer whenever two of them were in flight together: how often depends on thread timing); all of
    # them over the first 8-row span (at most 4 rows in flight)
    steps = [a - b for a, b in zip(root.pool_step_counts(), steps0)]
    assert steps[1:] == [0, 0, 0] and 5 <= steps[0] <= 45, steps
    assert sessions[1].pool_step_counts() == root.pool_step_counts()
    for s in sessions[1:]:
        s.close()


def test_pool_profile_counts_every_launch_and_changes_no_id(emu_lib):
    """vc_pool_profile: the step graphs re-captured with timing slots give the same ids, and the per-(span, kind) sums hold exactly
    one launch per kind and layer for every pool step (lm_head: one per step) with non-zero durations"""
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    plain = root.generate_greedy(ids, imgs, segs, deps, max_new_tokens=6)
    assert all(v["launches"] == 0 for v in root.pool_profile_read()[8].values()), "profiling is off by default"
    root.pool_profile(True)
    try:
        other = root.fork()
        outs, errs = [None, None], []

        def work(i, eng):
            try:
                outs[i] = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=6)
            except BaseException as e:
                errs.append(e)

        ths = [threading.Thread(target=work, args=(i, e)) for i, e in enumerate((root, other))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        assert np.array_equal(outs[0], plain) and np.array_equal(outs[1], plain)
        steps = root.pool_step_counts()   # (the pool was rebuilt for the slots: its histogram starts at zero)
        prof = root.pool_profile_read()
        L = root.cfg.num_hidden_layers
        for s_, rows in enumerate((8, 16, 24, 32)):
            for kind, v in prof[rows].items():
                want = steps[s_] * (1 if kind == "lm_head" else L)
                assert v["launches"] == want, (rows, kind, v, steps)
                assert (v["us"] > 0) == (want > 0) and (v["exec_us"] > 0) == (want > 0), (rows, kind, v)
        assert sum(steps) >= 5
        assert all(v["launches"] == 0 for r in root.pool_profile_read().values() for v in r.values()), "read(reset=True) zeroes the sums"
        other.close()
    finally:
        root.pool_profile(False)
    assert np.array_equal(root.generate_greedy(ids, imgs, segs, deps, max_new_tokens=6), plain)   # rebuilt without the slots


def test_pool_hold_policy_changes_scheduling_not_ids(emu_lib):
    """vc_pool_set_hold: with the policy on (default) the pool waits for a call that holds rows and is still prefilling instead of
    stepping without it — fewer, fuller steps; off, it steps whatever is active.  Either way every request gets the ids of its own
    session loop."""
    names = ["ds_img_depth_seg", "ds_img_only", "ds_img_seg"]
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    cases, refs = [], []
    for n in names:
        g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs(n)
        cases.append((ids, imgs, segs, deps))
        refs.append(session_loop_ids(root, ids, imgs, segs, deps, 5))
    sessions = [root, root.fork(), root.fork()]
    steps = {}
    try:
        for hold in (True, False):
            root.pool_set_hold(hold)
            s0 = root.pool_step_counts()
            outs, errs = [None] * 3, []

            def work(i):
                try:
                    outs[i] = sessions[i].generate_greedy(*cases[i], max_new_tokens=5)
                except BaseException as e:
                    errs.append(e)
            ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            assert not errs, errs
            for i in range(3):
                assert np.array_equal(outs[i], refs[i]), f"hold={hold}: request {i} differs from its session loop"
            steps[hold] = sum(a - b for a, b in zip(root.pool_step_counts(), s0))
        # three requests of 4 cached steps each: 4 steps when they all share every step, up to 12 when none do
        assert 4 <= steps[True] <= 12 and 4 <= steps[False] <= 12, steps
    finally:
        root.pool_set_hold(True)
        for s in sessions[1:]:
            s.close()


def test_pool_mixes_eos_stops_and_sampling(emu_lib):
    """Rows of one step with different generation parameters: an EOS request that ends early, a keyword-stop request, a
    sampled request and a plain greedy one, all in flight together; each equals its lone run."""
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    base = root.generate_greedy(ids, imgs, segs, deps, max_new_tokens=8)
    eos = int(base[0, 2])
    kws = [
        dict(max_new_tokens=8),
        dict(max_new_tokens=8, eos_token_id=eos, pad_token_id=0),
        dict(max_new_tokens=8, stop_sequences=[[int(base[1, 3])]], pad_token_id=0),
        dict(max_new_tokens=8, do_sample=True, temperature=0.9, top_k=20, top_p=0.95, seed=11),
    ]
    lone = [root.generate(ids, imgs, segs, deps, **kw) for kw in kws]
    assert lone[1].shape[1] <= 8 and (lone[1][0, 3:] == 0).all()            # row 0 pads after its EOS
    sessions = [root] + [root.fork() for _ in range(3)]
    outs, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(2):
                outs[i] = sessions[i].generate(ids, imgs, segs, deps, **kws[i])
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(outs[i], lone[i]), f"request {i} ({kws[i]}) changed when pooled with the others"
    for s in sessions[1:]:
        s.close()


def test_pool_queues_requests_beyond_its_rows(emu_lib):
    """5 concurrent requests of 8 rows want 40 rows of a 32-row pool: the fifth waits for rows and still gets its ids."""
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_only")
    from vcoder_amd import synth

    ids8 = np.concatenate([ids] * 8, axis=0)
    pix, _, _ = synth.synth_batch(8, cfg.vit_image_size)
    ref = session_loop_ids(root, ids8, pix, None, None, 4)
    sessions = [root] + [root.fork() for _ in range(4)]
    outs, errs = [None] * 5, []

    def work(i):
        try:
            outs[i] = sessions[i].generate_greedy(ids8, pix, None, None, max_new_tokens=4)
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(5)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for o in outs:
        assert np.array_equal(o, ref)
    for s in sessions[1:]:
        s.close()


def test_pool_random_schedule_stress(emu_lib, seed: int = 2024):
    """A randomised schedule: five sessions, each issuing two generate() calls with random case, length, EOS / stop-sequence /
    sampling parameters, padded masks and start delays — more rows wanted than the pool's first span holds at times, requests
    joining and leaving mid-flight, one call failing on purpose (unequal lengths).  Every call's ids equal the ids of the same
    call made alone afterwards, and the failing call fails the same way without disturbing the others."""
    import time

    rng = np.random.RandomState(seed)
    root = e2e_cases.engine_for("vcoder_ds", emu_lib)
    names = ["ds_img_depth_seg", "ds_img_only", "ds_img_seg"]
    cases = []
    for n in names:
        g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs(n)
        cases.append((ids, imgs, segs, deps))
    probe = root.generate_greedy(*cases[0], max_new_tokens=10)
    sessions = [root] + [root.fork() for _ in range(4)]
    plans = []
    for si in range(len(sessions)):
        calls = []
        for j in range(2):
            ci = int(rng.randint(3))
            kw = dict(max_new_tokens=int(rng.randint(2, 11)))
            r = rng.rand()
            if r < 0.25:
                kw.update(eos_token_id=int(probe[0, int(rng.randint(1, 6))]), pad_token_id=0)
            elif r < 0.45:
                kw.update(stop_sequences=[[int(probe[-1, int(rng.randint(1, 6))])]], pad_token_id=0)
            elif r < 0.7:
                kw.update(do_sample=True, temperature=float(rng.choice([0.7, 1.0, 1.3])), top_k=int(rng.choice([0, 10, 40])),
                          top_p=float(rng.choice([1.0, 0.9])), seed=int(rng.randint(1 << 30)))
            if rng.rand() < 0.3:        # a padded prompt: right padding hidden by the mask
                ids = cases[ci][0]
                mask = np.ones_like(ids)
                mask[:, -1] = 0
                kw["attention_mask"] = mask
            calls.append((ci, kw, float(rng.rand() * 0.05)))
        plans.append(calls)
    bad_ids = np.concatenate([cases[0][0][:1], np.where(cases[0][0][:1] < 0, 5, cases[0][0][:1])], 0)   # row 1 lost its placeholders
    plans[2][1] = ("bad", dict(max_new_tokens=4), 0.0)
    outs = [[None] * 2 for _ in sessions]
    errs = []

    def run(eng, ci, kw):
        if ci == "bad":
            imgs, segs, deps = cases[0][1:]
            return eng.generate(bad_ids, imgs, segs, deps, **kw)
        return eng.generate(*cases[ci], **kw)

    def work(si):
        try:
            for j, (ci, kw, delay) in enumerate(plans[si]):
                time.sleep(delay)
                try:
                    outs[si][j] = ("ok", run(sessions[si], ci, kw))
                except UnboundLocalError as e:
                    outs[si][j] = ("UnboundLocalError", str(e))
        except BaseException as e:
            errs.append((si, e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(sessions))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert outs[2][1][0] == "UnboundLocalError"
    for si in range(len(sessions)):
        for j, (ci, kw, _) in enumerate(plans[si]):
            if ci == "bad":
                continue
            kind, got = outs[si][j]
            want = run(root, ci, kw)                         # the same call, alone
            assert kind == "ok" and got.shape == want.shape and np.array_equal(got, want), \
                f"session {si} call {j} ({names[ci]}, {kw}): pooled ids differ from the lone call"
    for s in sessions[1:]:
        s.close()
