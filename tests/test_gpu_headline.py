"""Parity of the kernel configuration bench.py's headline actually times (ppo.jl:123-181 at num_envs = 65536 on one GPU):
`rollout_cartpole_kernel<2, true>` with 8 waves per block (4 at 32768 envs), the start stagger of waves 4-7, the critic on fp16x2
and the compat-mode GAE fused into the kernel's tail — all inside crl_ppo_iterate, which is the only caller that fuses.

Three checks per configuration, on the buffers crl_ppo_iterate leaves behind (the rollout ran under the initial parameters; the
optimiser steps that follow do not touch the buffers):
  (a) F_ADVANTAGE / F_RETURN are bit-equal to the CPU oracle's gae (orc_gae_batch, OpenMP) evaluated on the GPU's OWN value /
      reward / terminal buffers — the fused tail is the reference's serial Float64 recurrence, so not one bit may differ — and
      agree with the standalone gae_kernel (crl_gae) up to the ≤ 1e-6 share of last-bit differences its affine-map composition has;
  (b) the whole 65536 × 128 rollout against orc_rollout: actions, observations, rewards, terminals exact, with the C2 test's
      knot-margin rule (an env may leave the oracle's trajectory only where its uniform draw sits within 1e-6 of the CDF knot);
      logprob within 1e-5 relative, value within 1e-5 relative with the parity suite's 1e-6 absolute floor;
  (c) the same with rollout_stagger = 0, which isolates the stagger.
The oracle rollout of 8.4 M env-steps takes ≈10 s on the GPU box's host cores (OpenMP); it is cached per env count."""
import numpy as np
import pytest

import oraclelib as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6
K = 128


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1, "HIP library loaded but no GPU visible"
    return crl


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / (np.abs(b) + ATOL / RTOL)) if a.size else 0.0


_oracle_cache = {}


def oracle_rollout(nt, params):
    """Oracle buffers of the first rollout after a reset under `params` (identical for every option setting)."""
    if nt not in _oracle_cache:
        _oracle_cache.clear()                      # one env count at a time: ≈0.4 GB of arrays each
        cfg = O.make_config(num_envs=nt, num_steps=K)
        st = O.State(cfg)
        st.params[:] = params
        st.env_init(); st.rollout(); st.compute_gae()
        keep = {n: np.array(getattr(st, n), copy=True) for n in ("obs", "action", "logprob", "reward", "terminal", "value", "adv", "ret",
                                                                 "next_done", "env_state")}
        keep["episode_stats"] = st.episode_stats
        keep["params"] = np.array(params, copy=True)
        st.close()
        _oracle_cache[nt] = (cfg, keep)
    cfg, keep = _oracle_cache[nt]
    assert np.array_equal(keep["params"], params), "every configuration starts from the same initial parameters"
    return cfg, keep


def knot_margin(cfg, params, obs_col, e, t, seed=0x5EED):
    u = O.lib().orc_u53(seed, int(e), int(t), 0)
    _, _, _, margin = O.get_action(cfg, params, np.asfortranarray(obs_col[:, None]), np.array([u]))
    return float(margin[0])


# the 32768-env case first: the two 65536-env cases and the C4-shard test below then share ONE cached oracle rollout
@pytest.mark.parametrize("nt,stagger,expect", [(32768, 6, "4 waves per block"), (65536, 6, "8 waves per block, staggered"),
                                                (65536, 0, "8 waves per block, no stagger")])
def test_headline_rollout_and_fused_gae_match_the_oracle(crl, nt, stagger, expect):
    F = crl._lib
    cfg = crl.PPOConfig(num_envs=nt, num_steps=K, total_timesteps=nt * K * 10)
    agent = crl.Agent(cfg, options={"rollout_stagger": stagger})
    h = agent.handle
    assert h.get_option("gemm") == 2 and h.get_option("gae_fuse") == 1 and h.get_option("rollout_split_max_tiles") < nt // 32, \
        "defaults must select the one-wave-per-tile kernel with the fused GAE tail at this size"
    params = agent.get_params()
    h.env_reset()
    h.iterate(1, want_stats=False)
    value, reward, term = h.read(F.F_VALUE), h.read(F.F_REWARD), h.read(F.F_TERMINAL)
    adv, ret = h.read(F.F_ADVANTAGE), h.read(F.F_RETURN)

    # (a) the fused tail against the oracle's gae on the SAME inputs: bit for bit; and against the standalone kernel
    zeros_f, zeros_b = np.zeros(nt, np.float32), np.zeros(nt, np.uint8)
    adv_o, ret_o = O.gae_batch(value, reward, term, zeros_f, zeros_b, 0.99, 0.95, 0)
    assert np.array_equal(adv, adv_o), f"{np.sum(adv != adv_o)} advantages of the fused GAE tail differ from orc_gae on the same inputs"
    assert np.array_equal(ret, ret_o), f"{np.sum(ret != ret_o)} returns differ"
    adv_k, ret_k = F.gae_host(value, reward, term, None, None, 0.99, 0.95, 0)
    assert np.sum(adv_k != adv) <= adv.size * 1e-6 and rel_err(adv_k, adv) < 1e-6 and rel_err(ret_k, ret) < 1e-6
    assert not adv[:, -1].any(), "compat mode: the last slot is defined as 0 (Q1)"

    # (b) / (c) the rollout itself against orc_rollout
    cfgo, o = oracle_rollout(nt, params)
    act = h.read(F.F_ACTION)
    diff = act != o["action"]
    clean = ~diff.any(axis=1)
    for e in np.flatnonzero(~clean):
        t = int(np.argmax(diff[e]))
        m = knot_margin(cfgo, params, o["obs"][:, e, t], e, t)
        assert m <= 1e-6, f"env {e} step {t}: action differs although the draw is {m:.3e} away from the CDF knot ({expect})"
    assert clean.mean() > 0.9995, f"{(~clean).sum()} envs left the oracle's trajectory; knot hits are ~1e-7 per draw"
    obs = h.read(F.F_OBS)
    assert np.array_equal(obs[:, clean], o["obs"][:, clean]), "env dynamics are bit-exact by construction"
    del obs
    assert np.array_equal(term[clean], o["terminal"][clean]) and np.array_equal(reward[clean], o["reward"][clean])
    assert np.array_equal(h.read(F.F_NEXT_DONE)[clean], o["next_done"][clean])
    assert np.array_equal(h.read(F.F_ENV_STATE)[:, clean], o["env_state"][:, clean])
    assert rel_err(h.read(F.F_LOGPROB)[clean], o["logprob"][clean]) < RTOL
    assert rel_err(value[clean], o["value"][clean]) < RTOL
    # advantages end to end (GPU values differ from the oracle's in the last bits, so this one is a tolerance, not bits)
    assert rel_err(adv[clean], o["adv"][clean]) < RTOL and rel_err(ret[clean], o["ret"][clean]) < RTOL
    if clean.all():
        es = h.episode_stats(); n_ep, ret_sum, len_sum = o["episode_stats"]
        assert (es["episodes"], es["return_sum"], es["length_sum"]) == (n_ep, ret_sum, len_sum)
    agent.close()


def _global_sample_ids(local, n, off, nt_global):
    """flat index b = e + nt*t (ppo.jl:184-189) of a shard's local samples inside the NT-env batch"""
    return (off + local % n) + nt_global * (local // n)


def test_c4_shards_at_full_size_match_the_oracle(crl):
    """BASELINE config 4 — num_envs = 65536 sharded 8192 per GPU over 8 GPUs, one gradient all-reduce per optimiser step
    (ppo.jl:197,250 cadence) — on ONE GPU, shard by shard, at full size: ranks 0, 3 and 7 of the 8-rank job each build
    their handle exactly as `bench.py --gpus 8` does (num_envs = 8192, env_id_offset = 8192·rank, world 8), run one whole iteration
    (default options ⇒ the six-wave `rollout_split6_kernel` with the fused compat-GAE tail, the blocked shuffle, 16 optimiser
    steps) and are checked against the 65536-env CPU oracle:
      (a) the shard's rollout buffers == envs [8192·r, 8192·(r+1)) of the oracle's 65536-env rollout (global env ids key the RNG):
          actions / observations / rewards / terminals exact under the knot-margin rule, logprob / value / advantages / returns 1e-5;
      (b) the gradient message of one minibatch (CRL_F_GRADS before any exchange) with the GLOBAL minibatch size 8 × 262,144 and
          GLOBAL advantage statistics — the other seven ranks' Σadv / Σadv² come from the oracle's buffers — against
          orc_loss_grad(adv_stats = global) on the shard's own buffers, scaled by 1 / world;
      (c) Σ over the three ranks of those messages == 3/8 of the oracle's gradient of the three shards' UNION minibatch
          (what ncclAllReduce(sum) contributes for these ranks), as tests/test_dist_cpu.py does at toy size on CPU."""
    F = crl._lib
    W, n, NT = 8, 8192, 65536
    Bl = n * K; Ml = Bl // 4; Mg = Ml * W
    ranks = (0, 3, 7)
    rng = np.random.default_rng(44)
    perms = [rng.permutation(Bl).astype(np.int32) for _ in range(W)]   # every rank's local b_inds for the gradient step
    mb = 2
    cfgo = cfgs = None
    msgs = {}; shard_bufs = {}
    for r in ranks:
        cfg = crl.PPOConfig(num_envs=n, num_steps=K, total_timesteps=NT * K * 10)
        agent = crl.Agent(cfg, env_id_offset=n * r)
        h = agent.handle
        h.comm_init_external(W, r)
        assert h.get_option("rollout_split") == 4 and h.get_option("rollout_split_max_tiles") >= n // 32 and h.get_option("gae_fuse") == 1, \
            "defaults must select rollout_split6_kernel (rollout_split = 4: by size) with the fused GAE tail for an 8192-env shard"
        params = agent.get_params()
        cfgo, o = oracle_rollout(NT, params)
        sl = slice(n * r, n * (r + 1))
        h.env_reset()
        h.iterate(1, want_stats=False)                  # local gradients only (host-side exchange declared): buffers are what we check
        # ---- (a) rollout + fused GAE of the shard against its slice of the 65536-env oracle
        act = h.read(F.F_ACTION)
        diff = act != o["action"][sl]
        clean = ~diff.any(axis=1)
        for e in np.flatnonzero(~clean):
            t = int(np.argmax(diff[e]))
            m = knot_margin(cfgo, params, o["obs"][:, n * r + e, t], n * r + e, t)
            assert m <= 1e-6, f"rank {r} env {e} step {t}: action differs although the draw is {m:.3e} away from the CDF knot"
        assert clean.mean() > 0.999
        obs = h.read(F.F_OBS)
        assert np.array_equal(obs[:, clean], o["obs"][:, sl][:, clean]), "env dynamics are bit-exact by construction"
        term, reward, value, logprob = h.read(F.F_TERMINAL), h.read(F.F_REWARD), h.read(F.F_VALUE), h.read(F.F_LOGPROB)
        adv, ret = h.read(F.F_ADVANTAGE), h.read(F.F_RETURN)
        assert np.array_equal(term[clean], o["terminal"][sl][clean]) and np.array_equal(reward[clean], o["reward"][sl][clean])
        assert np.array_equal(h.read(F.F_NEXT_DONE)[clean], o["next_done"][sl][clean])
        assert np.array_equal(h.read(F.F_ENV_STATE)[:, clean], o["env_state"][:, sl][:, clean])
        assert rel_err(logprob[clean], o["logprob"][sl][clean]) < RTOL and rel_err(value[clean], o["value"][sl][clean]) < RTOL
        assert rel_err(adv[clean], o["adv"][sl][clean]) < RTOL and rel_err(ret[clean], o["ret"][sl][clean]) < RTOL
        adv_o, ret_o = O.gae_batch(value, reward, term, np.zeros(n, np.float32), np.zeros(n, np.uint8), 0.99, 0.95, 0)
        assert np.array_equal(adv, adv_o) and np.array_equal(ret, ret_o), "fused tail of the split kernel: bit-equal to orc_gae on its own inputs"
        # ---- (b) one gradient message under the global minibatch: parameters back to the rollout's, this test's permutation in
        agent.set_params(params)
        h.write(F.F_PERM, perms[r])
        h.adv_stats_local()
        loc = h.read(F.F_ADV_SUMS).reshape(-1)
        tot = np.zeros((4, 2)); mine = np.zeros((4, 2))
        for q in range(W):
            a64 = (adv.ravel(order="F") if q == r else o["adv"][n * q:n * (q + 1)].ravel(order="F")).astype(np.float64)
            for j in range(4):
                x = a64[perms[q][j * Ml:(j + 1) * Ml]]
                tot[j] += (x.sum(), (x * x).sum())
                if q == r:
                    mine[j] = (x.sum(), (x * x).sum())
        # CRL_F_ADV_SUMS: [Σadv, Σadv²] pairs, minibatch by minibatch (optim.hip adv_sums_fold_kernel)
        assert np.allclose(loc, mine.reshape(-1), rtol=1e-9, atol=1e-6), "local Σadv / Σadv² of the shard's minibatches"
        h.write(F.F_ADV_SUMS, tot.reshape(-1))
        h.adv_stats_finish()
        gs = h.update_minibatch(mb, 0.0, apply_update=False)
        g = h.read(F.F_GRADS).astype(np.float64)
        mean = tot[mb, 0] / Mg
        std = np.sqrt(max((tot[mb, 1] - Mg * mean * mean) / (Mg - 1), 0.0))
        cfgs = O.make_config(num_envs=n, num_steps=K, env_id_offset=n * r)
        g_o, so = O.loss_grad(cfgs, params, obs.reshape(4, -1, order="F"), act, logprob, value, adv, ret, perms[r][mb * Ml:(mb + 1) * Ml],
                              adv_stats=[np.float32(mean), np.float32(std)])
        assert abs(gs["adv_mean"] - np.float32(mean)) < 1e-6 and abs(gs["adv_std"] - np.float32(std)) < 1e-5 * std
        off = O.param_offsets(cfgs)
        for i in range(12):
            a, b = g[off[i]:off[i + 1]], g_o[off[i]:off[i + 1]].astype(np.float64) / W
            err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
            assert err < RTOL, f"rank {r}, gradient array {i}: rel L2 error {err:.3e} against the oracle's message (global M, global statistics)"
        msgs[r] = g
        shard_bufs[r] = (obs, act, logprob, value, adv, ret)
        agent.close()
    # ---- (c) the three messages summed == 3/8 of the oracle's gradient of the union minibatch of the three shards
    nu = n * len(ranks)
    cat = lambda i, ax: np.concatenate([shard_bufs[r][i] for r in ranks], axis=ax)
    obs_u = cat(0, 1); rest = [cat(i, 0) for i in range(1, 6)]
    union = np.concatenate([_global_sample_ids(perms[r][mb * Ml:(mb + 1) * Ml], n, n * j, nu) for j, r in enumerate(ranks)]).astype(np.int32)
    cfgu = O.make_config(num_envs=nu, num_steps=K)
    g_u, _ = O.loss_grad(cfgu, params, obs_u.reshape(4, -1, order="F"), *rest, union, adv_stats=[np.float32(mean), np.float32(std)])
    total = sum(msgs[r] for r in ranks)
    want = g_u.astype(np.float64) * len(ranks) / W
    off = O.param_offsets(cfgu)
    for i in range(12):
        err = np.linalg.norm(total[off[i]:off[i + 1]] - want[off[i]:off[i + 1]]) / max(np.linalg.norm(want[off[i]:off[i + 1]]), 1e-12)
        assert err < RTOL, f"summed messages, gradient array {i}: rel L2 error {err:.3e}"


def test_unfused_and_fused_gae_agree_at_the_headline_size(crl):
    """gae_fuse = 0 sends crl_ppo_iterate through the standalone gae_kernel: same rollout bits, advantages equal up to the standalone
    kernel's last-bit differences (its segments compose affine maps instead of running the serial recurrence)."""
    F = crl._lib
    nt = 65536
    cfg = crl.PPOConfig(num_envs=nt, num_steps=K, total_timesteps=nt * K * 10)
    a1 = crl.Agent(cfg)
    a2 = crl.Agent(cfg, params=a1.get_params(), options={"gae_fuse": 0})
    for a in (a1, a2):
        a.handle.env_reset(); a.handle.iterate(1, want_stats=False)
    for f in (F.F_ACTION, F.F_VALUE, F.F_REWARD, F.F_TERMINAL, F.F_LOGPROB):
        assert np.array_equal(a1.handle.read(f), a2.handle.read(f)), f
    adv1, adv2 = a1.handle.read(F.F_ADVANTAGE), a2.handle.read(F.F_ADVANTAGE)
    assert np.sum(adv1 != adv2) <= adv1.size * 1e-6 and rel_err(adv2, adv1) < 1e-6
    a1.close(); a2.close()


@pytest.mark.parametrize("nt,gae_mode,stale", [(40013, 0, 1), (20011, 1, 0)])
def test_ragged_large_shards_match_the_oracle(crl, nt, gae_mode, stale):
    """Env counts that are no multiple of 32 (a partly filled last tile, 8 and 4 waves per block) at sizes the small ragged cases (33,
    37, 70 envs) do not reach, with the other GAE mode (fixed: bootstrap critic pass + standalone scan, never fused) and fresh
    observations after a reset: one whole iteration's buffers against the oracle."""
    F = crl._lib
    cfg = crl.PPOConfig(num_envs=nt, num_steps=K, total_timesteps=nt * K * 10, num_minibatches=1)
    agent = crl.Agent(cfg, gae_mode=gae_mode, stale_obs=stale)
    h = agent.handle
    params = agent.get_params()
    cfgo = O.make_config(num_envs=nt, num_steps=K, num_minibatches=1, gae_mode=gae_mode, stale_obs=stale)
    st = O.State(cfgo); st.params[:] = params
    st.env_init(); st.rollout(); st.compute_gae()
    h.env_reset()
    h.iterate(1, want_stats=False)
    act = h.read(F.F_ACTION)
    diff = act != st.action
    clean = ~diff.any(axis=1)
    for e in np.flatnonzero(~clean):
        t = int(np.argmax(diff[e]))
        m = knot_margin(cfgo, params, st.obs[:, e, t], e, t)
        assert m <= 1e-6, f"env {e} step {t}: action differs although the draw is {m:.3e} away from the CDF knot"
    assert clean.mean() > 0.999
    assert np.array_equal(h.read(F.F_OBS)[:, clean], st.obs[:, clean])
    assert np.array_equal(h.read(F.F_TERMINAL)[clean], st.terminal[clean]) and np.array_equal(h.read(F.F_REWARD)[clean], st.reward[clean])
    assert rel_err(h.read(F.F_VALUE)[clean], st.value[clean]) < RTOL and rel_err(h.read(F.F_LOGPROB)[clean], st.logprob[clean]) < RTOL
    assert rel_err(h.read(F.F_ADVANTAGE)[clean], st.adv[clean]) < RTOL and rel_err(h.read(F.F_RETURN)[clean], st.ret[clean]) < RTOL
    agent.close(); st.close()
