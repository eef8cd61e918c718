"""crl_ppo_iterate_async / crl_ppo_drain (the pipelined read-back behind ppo() / train(): ppo.jl:147-165,246-248 logged without stalling the GPU) against the
synchronous read-back: the same records, one update late — loss records, episode statistics and the per-episode ring, for the fused path, the six-wave shard
regime, the layer-wise path, a guard window that closes by itself and one that is REPLAYED (γ = 0, critic bias 5: the speculation fails every iteration)."""
import numpy as np
import pytest

import oraclelib as O
from test_gpu_parity import IT_LOSS, loss_close, make_agent

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1
    return crl


def _sync_reports(h, n, ring):
    out = []
    for _ in range(n):
        st = h.iterate(1)
        ep = h.episode_stats()
        recs, tot = h.episode_records() if ring else ([], int(ep["episodes"]))
        out.append({"iteration": h.iteration - 1, "stats": st, "episodes": ep, "records": recs, "n_episodes": tot})
    return out


def _async_reports(h, n):
    out = []
    for _ in range(n):
        r = h.iterate_async()
        if r is not None:
            out.append(r)
    r = h.drain()
    assert r is not None
    out.append(r)
    assert h.drain() is None, "nothing is pending after a drain"
    return out


@pytest.mark.parametrize("kind,nt,ring", [("fused", 8, 64), ("fused", 70, 0), ("fused-4096", 4096, 65536), ("wide", 64, 128)])
def test_pipelined_reports_equal_the_synchronous_read_back(crl, kind, nt, ring, monkeypatch):
    L = crl._lib
    n_it = 10 if nt <= 70 else 4                       # ten iterations: a guard window of eight closes by itself in between
    # (the ring keeps the episodes that ARRIVE first — an atomic counter — so two runs store the same records only while every episode fits: 65536 at 4096 envs)
    reports = {}
    for mode in ("sync", "async"):
        cfg = crl.PPOConfig(num_envs=nt, num_steps=128, total_timesteps=nt * 128 * 20)
        shape = dict(obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC) if kind == "wide" else {}
        agent = crl.Agent(cfg, seed=7, init_seed=3, **shape)
        h = agent.handle
        if ring:
            h.episode_ring_enable(ring)
        h.env_reset()
        reports[mode] = _sync_reports(h, n_it, ring) if mode == "sync" else _async_reports(h, n_it)
        assert h.iteration == n_it
        reports[mode + "_params"] = h.read(L.F_PARAMS)
        agent.close()
    a, s = reports["async"], reports["sync"]
    assert [r["iteration"] for r in a] == list(range(n_it)) == [r["iteration"] for r in s]
    for ra, rs in zip(a, s):
        assert ra["stats"] == rs["stats"], ra["iteration"]                        # the same kernels on the same inputs: bit-equal records
        assert ra["episodes"] == rs["episodes"] and ra["records"] == rs["records"] and ra["n_episodes"] == rs["n_episodes"], ra["iteration"]
    assert np.array_equal(reports["async_params"], reports["sync_params"])
    assert any(r["n_episodes"] > 0 for r in a) and all(r["n_episodes"] == len(r["records"]) for r in a if ring)


def test_pipelined_read_back_replays_a_failed_speculation_before_it_reports(crl):
    """γ = 0 and a critic bias of 5 make u = mean(v − R²) > 0 in every minibatch: every slot arrives with the speculation flag up, the guard window is repeated
    exactly and the slot re-staged — the reports must be the ORACLE's, not the speculative pass's."""
    nt, k = 8, 128
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, gamma=0.0)
    agent = crl.Agent(cfg, shuffle_mode=crl._lib.SHUFFLE_FISHER_YATES)
    cfgo = O.make_config(num_envs=nt, num_steps=k, gamma=0.0)
    params = agent.get_params()
    params[O.param_offsets(cfgo)[11]] = 5.0
    agent.set_params(params)
    h = agent.handle
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h.env_reset()
    got = _async_reports(h, 3)
    assert h.exact_reruns >= 3
    for it, rep in enumerate(got):
        os_ = st.iterate(10, gen_perm=True)
        assert max(s["n_unclipped_wins"] for s in os_) > 0
        for a, b in zip(rep["stats"], os_):
            assert a["n_unclipped_wins"] == b["n_unclipped_wins"]
            for key in ("loss", "v_loss", "pg_loss"):
                assert loss_close(key, a[key], b[key], IT_LOSS, floor=1.5e-6), (it, key, a[key], b[key])
        n_ep, ret_sum, len_sum = st.episode_stats
        assert (rep["episodes"]["episodes"], rep["episodes"]["return_sum"], rep["episodes"]["length_sum"]) == (n_ep, ret_sum, len_sum)
    assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < 1e-6
    agent.close(); st.close()


def test_async_guards_and_mixing_with_the_synchronous_entry_points(crl):
    L = crl._lib
    from cleanrl_jl_amd.ppo import _crl_config
    h = L.Handle(_crl_config(crl.PPOConfig(num_envs=64, num_steps=16, total_timesteps=64 * 16 * 40)), 0)
    with pytest.raises(crl.CrlError, match="parameters not set"):
        h.iterate_async()
    assert h.drain() is None
    h.init_params(1)
    assert h.iterate_async() is None                         # first call: nothing before it
    sync_stats = h.iterate(1)                                # a synchronous iteration in between is not staged …
    r = h.iterate_async()                                    # … so this call has no predecessor to report (iteration 1 was not pipelined) but still reports nothing stale
    assert r is None or r["iteration"] != 1
    r2 = h.drain()
    assert r2 is not None and r2["iteration"] == 2 and len(r2["stats"]) == 16
    assert h.iteration == 3 and len(sync_stats) == 16
    # resizing the ring with an undelivered iteration pending is refused; after a drain it is fine
    h.iterate_async()
    h.episode_ring_enable(32)
    with pytest.raises(crl.CrlError, match="drain"):
        h.iterate_async()
    assert h.drain()["iteration"] == 3
    assert h.iterate_async() is None
    assert h.drain()["iteration"] == 4
    h.close()
