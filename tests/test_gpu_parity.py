"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bars (BASELINE.json north_star): bit-exact action indices / permutations / integer fields; float32 results within
1e-5 relative. GAE is additionally expected bit-exact (Float64 accumulation on both sides).
Whole-iteration comparisons use TIGHTER bars than the north_star's: scripts/parity_margins.py (profiles/r02_parity_margins.json)
measures 1.1e-7 relative on the losses, 3e-8 absolute on the parameters after three iterations and 1.2e-6 relative L2 on the
gradient arrays, so losses are held to 2e-6 relative and parameters to 1e-6 absolute (IT_LOSS / IT_PARAM below) — wide enough
for float32 summation-order noise on both sides, an order of magnitude inside "1e-5 relative"."""
import numpy as np
import pytest

import oraclelib as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5
IT_LOSS, IT_PARAM = 2e-6, 1e-6
# Loss scalars are compared RELATIVELY: |got − want| <= rtol·|want|. Only `pg_loss` (and `loss`, which contains it) get an absolute
# floor on top: they are means of O(1) terms −Â·ρ that cancel to ≈0 by construction (Â is normalised; in the first minibatch of an
# iteration ρ = 1 and pg_loss = −mean(Â) ≈ 1e-8), so the float32 rounding of the TERMS (1.2e-7 each, a few operations deep) is the
# unit of their error, not the size of the result. v_loss and entropy_loss are sums of same-signed terms: no floor.
LOSS_FLOOR = 5e-7


def loss_close(key, got, want, rtol, floor=LOSS_FLOOR):
    floor = floor if key in ("loss", "pg_loss") else 0.0
    return abs(got - want) <= rtol * abs(want) + floor


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1, "HIP library loaded but no GPU visible"
    return crl


def make_agent(crl, nt=8, k=128, params=None, **kw):
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, **{a: b for a, b in kw.items() if a in
                        ("num_minibatches", "update_epochs", "clip_value_loss", "anneal_lr", "lr", "clip_coef", "ent_coeff", "v_coef")})
    shape = {a: b for a, b in kw.items() if a in ("gae_mode", "shuffle_mode", "stale_obs", "env_id_offset", "seed", "options")}
    return crl.Agent(cfg, params=params, **shape)


ATOL = 1e-6


def rel_err(a, b):
    """max |a-b| / (|b| + ATOL/RTOL): `< RTOL` ⇔ |a-b| <= RTOL*|b| + ATOL. The absolute floor covers float32 results
    that cancel to ~0 (a value head summing 64 O(0.1) terms carries ~1e-7 absolute rounding on BOTH sides)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / (np.abs(b) + ATOL / RTOL)) if a.size else 0.0


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nt,k", [(1, 1), (1, 2), (3, 5), (8, 128), (33, 17), (64, 16), (100, 129), (4096, 128),
                                  (262144, 128)])   # the last one: 0.57 GB per launch, past the Infinity Cache (bench.py roofline_gae.beyond_cache)
@pytest.mark.parametrize("mode", [0, 1])
def test_gae_matches_oracle(crl, nt, k, mode):
    rng = np.random.default_rng(nt * 1000 + k)
    value = np.asfortranarray((rng.standard_normal((nt, k)) * 10).astype(np.float32))
    reward = np.asfortranarray((rng.random((nt, k)) > 0.02).astype(np.float32))
    term = np.asfortranarray((rng.random((nt, k)) < 0.02).astype(np.uint8))
    nv = (rng.standard_normal(nt) * 10).astype(np.float32); nd = (rng.random(nt) < 0.1).astype(np.uint8)
    adv_o, ret_o = O.gae_batch(value, reward, term, nv, nd, 0.99, 0.95, mode)
    adv_g, ret_g = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode)
    mism = np.sum(adv_o != adv_g)
    assert mism <= adv_o.size * 1e-6, f"{mism} of {adv_o.size} advantages differ bitwise"
    assert rel_err(adv_g, adv_o) < 1e-6 and rel_err(ret_g, ret_o) < 1e-6


def test_gae_single_env_api_and_edges(crl):
    v = [1, 2, 3, 4, 5]; r = [1, 1, 1, 1]; t = [0, 0, 1, 0, 0]
    a = crl.gae(v, r, t, 0.5, 0.5)
    assert a.tolist() == [0.75, -1.0, 0.0, 0.0]
    a = crl.gae(v, r, t, 0.5, 0.5, mode=crl._lib.GAE_FIXED)
    assert a.tolist() == [0.75, -1.0, -0.125, -0.5]
    assert crl.gae([1.0], [], [0], 0.99, 0.95).shape == (0,)       # empty rollout
    with pytest.raises(ValueError):
        crl.gae([1, 2], [1, 1], [0, 0], 0.99, 0.95)                  # ragged input


@pytest.mark.parametrize("n", [1, 31, 32, 33, 100, 4096])
def test_policy_act_matches_oracle(crl, n):
    rng = np.random.default_rng(n)
    agent = make_agent(crl)
    cfg = O.make_config()
    params = O.orthogonal_params(cfg, 3)
    params += (0.05 * rng.standard_normal(params.shape)).astype(np.float32)
    off = O.param_offsets(cfg)
    params[off[4]:off[5]] *= 30  # spread the logits so both actions occur with non-trivial probabilities
    agent.set_params(params)
    obs = np.asfortranarray((rng.standard_normal((4, n)) * np.array([[1.0], [1.0], [0.1], [1.0]])).astype(np.float32))
    u = rng.random(n)
    a_o, lp_o, v_o, margin = O.get_action(cfg, params, obs, u)
    a_g, lp_g, v_g = agent.handle.policy_act(obs, u)
    safe = margin > 1e-6
    assert np.array_equal(a_g[safe], a_o[safe]), "action indices must be bit-exact away from CDF knots"
    assert safe.mean() > 0.99
    same = a_g == a_o
    assert rel_err(lp_g[same], lp_o[same]) < RTOL
    assert rel_err(v_g, v_o) < RTOL
    # 1-based wrapper like the reference
    a1, lp1 = crl.get_action(obs, agent.actor, u=u)
    assert np.array_equal(a1, a_g + 1) and np.array_equal(lp1, lp_g)
    agent.close()


def test_logprob_actions_matches_oracle(crl):
    rng = np.random.default_rng(11)
    agent = make_agent(crl)
    cfg = O.make_config()
    params = O.orthogonal_params(cfg, 4) + (0.1 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    agent.set_params(params)
    n = 257
    obs = np.asfortranarray(rng.standard_normal((4, n)).astype(np.float32))
    acts = rng.integers(0, 2, n).astype(np.int32)
    lp_o, ent_o = O.logprob_actions(cfg, params, obs, acts)
    lp_g, ent_g = crl.logprob_actions(obs, agent.actor, acts + 1)
    assert rel_err(lp_g, lp_o) < RTOL and rel_err(ent_g, ent_o) < RTOL and ent_g.shape == (2, n)
    with pytest.raises(TypeError):
        crl.logprob_actions(obs, agent.actor, acts.astype(np.int64))
    agent.close()


# ---------------------------------------------------------------------------------------------------------
def _oracle_state(nt, k, params, **kw):
    cfg = O.make_config(num_envs=nt, num_steps=k, **kw)
    st = O.State(cfg)
    st.params[:] = params
    st.env_init()
    return cfg, st


@pytest.mark.parametrize("split", ["1", "2", "0", "3"])
@pytest.mark.parametrize("nt,stale", [(8, 1), (8, 0), (70, 1)])
def test_rollout_matches_oracle(crl, nt, stale, split):
    """Option rollout_split = 1: three waves per tile (actor rows 0-31 + sampling + env | actor rows 32-63 | critic), the small-shard
    kernel; 2: two waves per tile (actor + env | critic); 0: one wave per tile; 3: six waves per tile (the actor's rows over four waves on
    16x16x32 products, the critic's over two: rollout_split6_kernel)."""
    k = 128
    agent = make_agent(crl, nt=nt, k=k, stale_obs=stale, options={"rollout_split": int(split)})
    params = agent.get_params()
    cfg, st = _oracle_state(nt, k, params, stale_obs=stale)
    h = agent.handle
    h.env_reset()
    assert np.array_equal(h.read(crl._lib.F_CUR_OBS), st.cur_obs), "reset stream must match bit for bit"
    h.rollout_run(); st.rollout()
    act_g, act_o = h.read(crl._lib.F_ACTION), st.action
    assert np.array_equal(act_g, act_o), f"{np.sum(act_g != act_o)} actions differ"
    assert np.array_equal(h.read(crl._lib.F_TERMINAL), st.terminal)
    assert np.array_equal(h.read(crl._lib.F_REWARD), st.reward)
    assert np.array_equal(h.read(crl._lib.F_OBS), st.obs), "env dynamics are bit-exact by construction"
    assert rel_err(h.read(crl._lib.F_LOGPROB), st.logprob) < RTOL
    assert rel_err(h.read(crl._lib.F_VALUE), st.value) < RTOL
    assert np.array_equal(h.read(crl._lib.F_NEXT_DONE), st.next_done)
    assert np.array_equal(h.read(crl._lib.F_ENV_STATE), st.env_state)
    es = h.episode_stats(); n_ep, ret_sum, len_sum = st.episode_stats
    assert (es["episodes"], es["return_sum"], es["length_sum"]) == (n_ep, ret_sum, len_sum)
    h.compute_gae(); st.compute_gae()
    assert rel_err(h.read(crl._lib.F_ADVANTAGE), st.adv) < RTOL and rel_err(h.read(crl._lib.F_RETURN), st.ret) < RTOL
    agent.close(); st.close()


def _inject_batch(crl, agent, st, rng, ret_scale=10.0):
    """Same synthetic rollout buffer into the GPU handle and the oracle state."""
    nt, k = st.cfg.num_envs, st.cfg.num_steps
    st.obs[:] = rng.standard_normal((4, nt, k)).astype(np.float32)
    st.action[:] = rng.integers(0, 2, (nt, k))
    st.logprob[:] = (np.log(0.5) + 0.3 * rng.standard_normal((nt, k))).astype(np.float32)
    st.value[:] = rng.standard_normal((nt, k)).astype(np.float32) * (1.0 if ret_scale > 1 else 0.05)
    st.adv[:] = (2 * rng.standard_normal((nt, k))).astype(np.float32)
    st.ret[:] = (ret_scale * rng.standard_normal((nt, k))).astype(np.float32)
    st.perm[:] = rng.permutation(nt * k).astype(np.int32)
    h = agent.handle; F = crl._lib
    for f, a in ((F.F_OBS, st.obs), (F.F_ACTION, st.action), (F.F_LOGPROB, st.logprob), (F.F_VALUE, st.value),
                 (F.F_ADVANTAGE, st.adv), (F.F_RETURN, st.ret), (F.F_PERM, st.perm)):
        h.write(f, a)


def _grad_close(g_gpu, g_orc, off, tol=RTOL):
    for i in range(12):
        a, b = g_gpu[off[i]:off[i + 1]].astype(np.float64), g_orc[off[i]:off[i + 1]].astype(np.float64)
        err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
        assert err < tol, f"gradient array {i}: rel L2 error {err:.3e}"


@pytest.mark.parametrize("nt,k,ret_scale,clipv", [(8, 128, 10.0, True), (8, 128, 0.05, True), (8, 128, 3.0, False),
                                                   (64, 128, 10.0, True), (37, 64, 10.0, True)])
def test_update_gradient_matches_oracle(crl, nt, k, ret_scale, clipv):
    """ret_scale=0.05 drives u = mean(v - R²) > 0: the speculative value-loss pass must be redone exactly (Q4)."""
    if (nt * k) % 4:
        pytest.skip("batch not divisible")
    rng = np.random.default_rng(nt + k)
    cfgo = O.make_config(num_envs=nt, num_steps=k, clip_value_loss=clipv)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    off = O.param_offsets(cfgo)
    if ret_scale < 1:
        params[off[11]] = 0.3
    agent = make_agent(crl, nt=nt, k=k, params=params, clip_value_loss=clipv)
    st = O.State(cfgo); st.params[:] = params
    _inject_batch(crl, agent, st, rng, ret_scale)
    h = agent.handle
    h.adv_stats()
    M = nt * k // 4
    for mb in (0, 3):
        gs = h.update_minibatch(mb, 2.5e-4, apply_update=False)
        g_gpu = h.read(crl._lib.F_GRADS)
        g_orc, so = O.loss_grad(cfgo, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret,
                                st.perm[mb * M:(mb + 1) * M])
        if ret_scale < 1 and clipv:
            assert so["n_unclipped_wins"] > 0 and gs["n_unclipped_wins"] == so["n_unclipped_wins"]
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
        assert abs(gs["adv_mean"] - so["adv_mean"]) < 1e-6 and abs(gs["adv_std"] - so["adv_std"]) < 1e-5 * so["adv_std"]
        _grad_close(g_gpu, g_orc, off)
    agent.close(); st.close()


def test_optimizer_step_matches_oracle(crl):
    rng = np.random.default_rng(21)
    nt, k = 8, 128
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    params = O.orthogonal_params(cfgo, 6)
    agent = make_agent(crl, nt=nt, k=k, params=params)
    st = O.State(cfgo); st.params[:] = params
    _inject_batch(crl, agent, st, rng)
    h = agent.handle
    h.adv_stats()
    m = np.zeros_like(params); v = np.zeros_like(params); betap = np.array([0.9, 0.999] * 12)
    p_ref = params.copy()
    for step, mb in enumerate((0, 1, 2)):
        h.update_minibatch(mb, 2.5e-4 * (1 - 0.1 * step), apply_update=True)
        g = h.read(crl._lib.F_GRADS).copy()
        O.clipnorm_adam(cfgo, p_ref, g, m, v, betap, 2.5e-4 * (1 - 0.1 * step))
        assert np.array_equal(h.read(crl._lib.F_PARAMS), p_ref), "ClipNorm+Adam is Float64 scalar math: expect bit parity"
        assert np.array_equal(h.read(crl._lib.F_ADAM_M), m) and np.array_equal(h.read(crl._lib.F_ADAM_V), v)
        assert np.allclose(h.read(crl._lib.F_BETAP), betap, rtol=1e-15)
    agent.close(); st.close()


def test_shuffle_fisher_yates_matches_oracle(crl):
    agent = make_agent(crl, nt=8, k=128, shuffle_mode=0, seed=77)
    h = agent.handle
    ref = np.arange(1024, dtype=np.int32)
    for ep in range(3):  # cumulative like b_inds = shuffle(b_inds) (ppo.jl:194)
        h.shuffle(ep); ref = O.shuffle_fy(ref, 77, ep)
        assert np.array_equal(h.read(crl._lib.F_PERM), ref)
    agent.close()


@pytest.mark.parametrize("nt,k", [(8, 128), (37, 64), (4096, 128)])
def test_shuffle_bijection_is_a_permutation(crl, nt, k):
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=1)
    h = agent.handle
    h.shuffle(5); p1 = h.read(crl._lib.F_PERM)
    h.shuffle(5); p2 = h.read(crl._lib.F_PERM)
    h.shuffle(6); p3 = h.read(crl._lib.F_PERM)
    B = nt * k
    assert np.array_equal(np.sort(p1), np.arange(B)) and np.array_equal(p1, p2) and not np.array_equal(p1, p3)
    # crude mixing check: each minibatch slice draws from the whole index range
    q = p1[:B // 4].astype(np.float64)
    assert abs(q.mean() / B - 0.5) < 0.05 and np.mean(p1 == np.arange(B)) < 0.01
    agent.close()


@pytest.mark.parametrize("n_iters,fuse_optim,gemm", [(1, 1, 2), (3, 1, 2), (3, 0, 2), (3, 1, 1)])
def test_full_iteration_matches_oracle(crl, n_iters, fuse_optim, gemm):
    """C1 (BASELINE configs[0]): num_envs=8, num_steps=128 — whole ppo.jl:117-253 loop body, exact Fisher–Yates; with the gradient
    reduction + ClipNorm + Adam as one launch (the default inside crl_ppo_iterate on one GPU) and as two; gemm = 1 is the bf16x3
    flavour (24-bit operands) behind bench.py's `strict_f32` figure — rollout_split_kernel + update_x3_kernel."""
    nt, k = 8, 128
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=0, options={"fuse_optim": fuse_optim, "gemm": gemm})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle
    h.env_reset()
    for it in range(n_iters):
        gs = h.iterate(1)
        os_ = st.iterate(10, gen_perm=True)
        assert np.array_equal(h.read(crl._lib.F_PERM), st.perm)
        acts = h.read(crl._lib.F_ACTION)
        assert np.array_equal(acts, st.action), f"iteration {it}: {np.sum(acts != st.action)} actions differ"
        assert rel_err(h.read(crl._lib.F_ADVANTAGE), st.adv) < RTOL
        for a, b in zip(gs, os_):
            for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                assert loss_close(key, a[key], b[key], IT_LOSS), (it, key, a[key], b[key])
        pg, po = h.read(crl._lib.F_PARAMS), st.params
        assert np.max(np.abs(pg - po)) < IT_PARAM, np.max(np.abs(pg - po))
    assert h.iteration == n_iters
    agent.close(); st.close()


def test_fused_and_two_launch_optimiser_steps_agree_at_c2_size(crl):
    """reduce_optim_kernel (one launch, grid-wide meeting point) against reduce_kernel + clipnorm_adam_kernel on 4096 envs x 128 steps,
    two whole iterations = 32 optimiser steps: the gradients are the same bits, the per-array norms differ at most in the order their
    Float64 partial sums are added, so parameters and Adam state agree to a few ulps."""
    F = crl._lib
    a1 = make_agent(crl, nt=4096, k=128)
    a0 = make_agent(crl, nt=4096, k=128, params=a1.get_params(), options={"fuse_optim": 0})
    for a in (a1, a0):
        a.handle.env_reset(); a.handle.iterate(2, want_stats=False)
    for f in (F.F_PARAMS, F.F_ADAM_M, F.F_ADAM_V):
        x, y = a1.handle.read(f), a0.handle.read(f)
        assert np.max(np.abs(x - y)) <= 1e-7 * max(1.0, float(np.max(np.abs(y)))), (f, np.max(np.abs(x - y)))
    assert np.array_equal(a1.handle.read(F.F_BETAP), a0.handle.read(F.F_BETAP))
    assert np.array_equal(a1.handle.read(F.F_ACTION), a0.handle.read(F.F_ACTION))
    a1.close(); a0.close()


def test_errors_are_reported_not_thrown(crl):
    with pytest.raises(crl.CrlError, match="normalize_advantages"):
        crl.Agent(crl.PPOConfig(normalize_advantages=False))
    with pytest.raises(crl.CrlError, match="divisible"):
        crl.Agent(crl.PPOConfig(num_envs=3, num_steps=5, num_minibatches=4))
    with pytest.raises(crl.CrlError, match="hidden must be"):
        crl.Agent(crl.PPOConfig(), hidden=96)
    agent = make_agent(crl)
    with pytest.raises(crl.CrlError, match="out of range"):
        agent.handle.update_minibatch(9, 1e-3)
    bad = np.arange(agent.config.num_envs * agent.config.num_steps, dtype=np.int32)
    bad[5] = bad.size                      # the kernels index the batch through b_inds: out-of-range entries are refused
    with pytest.raises(crl.CrlError, match="outside"):
        agent.handle.write(crl._lib.F_PERM, bad)
    agent.close()


# ---- BASELINE full size: properties that do not need the oracle to be fast --------------------------------
def test_full_size_gae_and_scan_linearity(crl):
    nt, k = 65536, 128
    rng = np.random.default_rng(0)
    value = np.asfortranarray((rng.standard_normal((nt, k)) * 10).astype(np.float32))
    reward = np.asfortranarray((rng.random((nt, k)) > 0.02).astype(np.float32))
    term = np.asfortranarray((rng.random((nt, k)) < 0.02).astype(np.uint8))
    adv, ret = crl._lib.gae_host(value, reward, term, None, None, 0.99, 0.95, 0)
    adv_o, ret_o = O.gae_batch(value, reward, term, np.zeros(nt, np.float32), np.zeros(nt, np.uint8), 0.99, 0.95, 0)
    assert np.sum(adv != adv_o) <= 10 and rel_err(adv, adv_o) < 1e-6
    assert np.array_equal(ret, adv + value) and not adv[:, -1].any()
    # a terminal at t+1 cuts the scan: advantage at t is exactly δ_t = r_t - v_t
    e, t = np.argwhere(term[:, 1:] == 1)[0]
    assert adv[e, t] == np.float32(np.float64(reward[e, t]) - np.float64(value[e, t]))


def test_rccl_path_world1(crl):
    """The gradient / advantage-statistics all-reduces go through RCCL (dlopen'ed librccl) even on one GPU with option
    comm_force = 1: a sum over one rank is the identity, so the iteration must still match the oracle."""
    nt, k = 8, 128
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=0, options={"comm_force": 1})
    h = agent.handle
    h.comm_init(crl.comm_unique_id(), 1, 0)
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h.env_reset()
    h.prof_enable(True)
    gs = h.iterate(1); os_ = st.iterate(10, gen_perm=True)
    assert h.prof_read()["allreduce"][1] == 16, "one all-reduce per optimiser step (ppo.jl:250 cadence)"
    for a, b in zip(gs, os_):
        assert loss_close("loss", a["loss"], b["loss"], IT_LOSS), (a["loss"], b["loss"])
    assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < IT_PARAM
    agent.close(); st.close()


@pytest.mark.parametrize("nt,k", [(8, 128), (37, 64), (4096, 128)])
def test_adv_stats_through_inverse_bijection(crl, nt, k):
    """With the keyed-bijection shuffle the advantage statistics are summed in sample order and each sample's minibatch
    comes from the INVERSE bijection; they must equal mean/std over the perm slices (ppo.jl:221)."""
    rng = np.random.default_rng(nt)
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=1)
    h = agent.handle; F = crl._lib
    adv = (3 * rng.standard_normal((nt, k)) + 0.7).astype(np.float32)
    for f, a in ((F.F_ADVANTAGE, adv), (F.F_OBS, rng.standard_normal((4, nt, k)).astype(np.float32)),
                 (F.F_RETURN, 10 * rng.standard_normal((nt, k)).astype(np.float32))):
        h.write(f, a)
    h.shuffle(11)
    perm = h.read(F.F_PERM)
    h.adv_stats()
    M = nt * k // 4
    flat = adv.ravel(order="F").astype(np.float64)
    for mb in range(4):
        st = h.update_minibatch(mb, 0.0, apply_update=False)
        sl = flat[perm[mb * M:(mb + 1) * M]]
        assert abs(st["adv_mean"] - np.float32(sl.mean())) < 1e-6
        assert abs(st["adv_std"] - np.float32(sl.std(ddof=1))) < 1e-5
    agent.close()


def test_external_env_path_store_then_update(crl):
    """The reference's own loop shape (ppo.jl:123-166): the HOST steps the envs, asks the device for actions
    (crl_policy_act), stores each transition (crl_rollout_store = Buffer.add!), then GAE + one optimiser step on device."""
    nt, k = 8, 32
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=0)
    params = agent.get_params()
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h = agent.handle; F = crl._lib
    # host-side env loop driven by the oracle's CartPole; actions come from the GPU with the oracle's uniform draws
    import ctypes as C
    s = st.env_state; co = st.cur_obs
    next_done = np.zeros(nt, np.uint8)
    t_env = np.zeros(nt, np.int32)
    for step in range(k):
        u = np.array([O.lib().orc_u53(cfgo.seed, e, step, 0) for e in range(nt)])
        a, lp, v = h.policy_act(co, u)
        a_o, lp_o, v_o, margin = O.get_action(cfgo, params, np.asfortranarray(co), u)
        assert np.array_equal(a[margin > 1e-6], a_o[margin > 1e-6])
        rew = np.zeros(nt, np.float32); done = np.zeros(nt, np.int32)
        obs_in = co.copy()
        for e in range(nt):
            se = np.ascontiguousarray(s[:, e]); te = t_env[e:e + 1].copy(); de = np.zeros(1, np.int32)
            O.lib().orc_cartpole_step(O.fptr(se), te.ctypes.data_as(C.POINTER(C.c_int32)), int(a[e]), 500, de.ctypes.data_as(C.POINTER(C.c_int32)))
            s[:, e] = se; t_env[e] = te[0]; done[e] = de[0]; rew[e] = 0.0 if de[0] else 1.0
        h.rollout_store(step, obs_in, a, lp, rew, next_done, v)
        st.obs[:, :, step] = obs_in; st.action[:, step] = a; st.logprob[:, step] = lp; st.reward[:, step] = rew
        st.terminal[:, step] = next_done; st.value[:, step] = v
        co[:] = s; next_done = done.astype(np.uint8)
        for e in range(nt):
            if done[e]:
                se = np.zeros(4, np.float32)
                O.lib().orc_env_reset(C.byref(cfgo), O.fptr(se), e, step, 1); s[:, e] = se; t_env[e] = 0
    st.next_done[:] = next_done
    h.write(F.F_NEXT_DONE, next_done)
    assert np.array_equal(h.read(F.F_OBS), st.obs) and np.array_equal(h.read(F.F_TERMINAL), st.terminal)
    h.compute_gae(); st.compute_gae()
    assert rel_err(h.read(F.F_ADVANTAGE), st.adv) < RTOL
    st.perm[:] = np.random.default_rng(1).permutation(nt * k).astype(np.int32)
    h.write(F.F_PERM, st.perm)
    h.adv_stats()
    gs = h.update_minibatch(1, 2.5e-4, apply_update=True)
    so = st.update_minibatch(1, 2.5e-4)
    assert loss_close("loss", gs["loss"], so["loss"], IT_LOSS), (gs["loss"], so["loss"])
    assert np.max(np.abs(h.read(F.F_PARAMS) - st.params)) < 1e-6
    # critic(obs) through the Policy object, like `value = critic(next_obs)` (ppo.jl:128)
    vv = agent.critic(co)
    assert vv.shape == (1, nt)
    agent.close(); st.close()


@pytest.mark.parametrize("nt,k", [(1, 4), (8, 128), (37, 64), (4096, 128), (65536, 128), (131200, 128)])
def test_shuffle_blocked_fisher_yates_matches_oracle(crl, nt, k):
    """Exact parallel shuffle (Rao–Sandelius split + Fisher–Yates leaves): bit-identical to its CPU restatement, a
    permutation, deterministic, and different per epoch. The last size (16,793,600 samples) is past 2^24: a sample's second digit
    no longer rides in the top byte of its scattered entry, the leaf pass recomputes it."""
    if (nt * k) % 4:
        pytest.skip("batch not divisible")
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=2, seed=99)
    h = agent.handle
    B = nt * k
    for ep in (0, 5):
        h.shuffle(ep)
        p = h.read(crl._lib.F_PERM)
        assert np.array_equal(p, O.shuffle_blocked_fy(B, 99, ep))
        assert np.array_equal(np.sort(p), np.arange(B))
    agent.close()


def test_iteration_with_blocked_fisher_yates(crl):
    """Whole iteration with the exact parallel shuffle: same losses/parameters as the oracle fed the same permutations."""
    nt, k = 8, 128
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=2)
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle
    h.env_reset(); h.rollout_run(); h.compute_gae()
    st.rollout(); st.compute_gae()
    for ep in range(2):
        h.shuffle(ep); h.adv_stats()
        st.perm[:] = O.shuffle_blocked_fy(nt * k, cfgo.seed, ep)
        assert np.array_equal(h.read(crl._lib.F_PERM), st.perm)
        for mb in range(4):
            a = h.update_minibatch(mb, 2.5e-4)
            b = st.update_minibatch(mb, 2.5e-4)
            assert loss_close("loss", a["loss"], b["loss"], IT_LOSS), (a["loss"], b["loss"])
    assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < IT_PARAM
    agent.close(); st.close()


def _knot_margin(cfgo, params, st, e, t, seed=0x5EED):
    """Distance of the draw of env e at step t (first rollout after a reset) from the nearest CDF knot, by the oracle, on the
    oracle's own observation (identical on both sides up to the first differing action)."""
    u = O.lib().orc_u53(seed, int(e), int(t), 0)
    _, _, _, margin = O.get_action(cfgo, params, np.asfortranarray(st.obs[:, e, t][:, None]), np.array([u]))
    return float(margin[0])


@pytest.mark.parametrize("gemm", [2, 1])
def test_c2_size_rollout_and_update_match_oracle(crl, gemm):
    """gemm = 2: the fp16x2 default; gemm = 1: bf16x3 everywhere (bench.py `strict_f32`).
    BASELINE configs[1] (C2): num_envs=4096, num_steps=128, 2x64 — the oracle still finishes in seconds here, so the
    whole rollout, GAE and one minibatch gradient (M = 131,072 samples) are compared directly, not through properties.
    An action may differ from the oracle's ONLY where the uniform draw sits within 1e-6 of a CDF knot: every first
    difference of an env is checked against that margin and fails the test otherwise; envs without one must match exactly."""
    nt, k = 4096, 128
    agent = make_agent(crl, nt=nt, k=k, options={"gemm": gemm})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); st.rollout()
    act = h.read(F.F_ACTION)
    diff = act != st.action
    clean = ~diff.any(axis=1)                       # envs whose whole trajectory has identical actions
    for e in np.flatnonzero(~clean):
        t = int(np.argmax(diff[e]))
        m = _knot_margin(cfgo, params, st, e, t)
        assert m <= 1e-6, f"env {e} step {t}: action differs although the draw is {m:.3e} away from the CDF knot"
    assert clean.mean() > 0.999, "knot hits are ~1e-7 per draw; more than a handful means a real bug"
    assert np.array_equal(h.read(F.F_OBS)[:, clean], st.obs[:, clean]) and np.array_equal(h.read(F.F_TERMINAL)[clean], st.terminal[clean])
    assert np.array_equal(h.read(F.F_REWARD)[clean], st.reward[clean])
    assert rel_err(h.read(F.F_LOGPROB)[clean], st.logprob[clean]) < RTOL and rel_err(h.read(F.F_VALUE)[clean], st.value[clean]) < RTOL
    h.compute_gae(); st.compute_gae()
    assert rel_err(h.read(F.F_ADVANTAGE)[clean], st.adv[clean]) < RTOL
    # gradient of minibatch 2 at the initial parameters: the oracle differentiates the GPU's own buffer, so a knot hit in the
    # rollout cannot leak into this comparison
    obs, action, logprob, value = h.read(F.F_OBS), h.read(F.F_ACTION), h.read(F.F_LOGPROB), h.read(F.F_VALUE)
    adv, ret = h.read(F.F_ADVANTAGE), h.read(F.F_RETURN)
    perm = np.random.default_rng(5).permutation(nt * k).astype(np.int32)
    h.write(F.F_PERM, perm)
    h.adv_stats()
    gs = h.update_minibatch(2, 0.0, apply_update=False)
    M = nt * k // 4
    g_o, so = O.loss_grad(cfgo, params, obs.reshape(4, -1, order="F"), action, logprob, value, adv, ret, perm[2 * M:3 * M])
    for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
        assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
    _grad_close(h.read(F.F_GRADS), g_o, O.param_offsets(cfgo))
    agent.close(); st.close()


@pytest.mark.parametrize("gemm", [2, 1])
def test_c4_size_minibatch_gradient_matches_oracle(crl, gemm):
    """gemm = 1: update_x3_kernel at M = 2,097,152 — the launch bench.py's `strict_f32` record times. BASELINE full size, directly: ONE minibatch of the headline configuration (num_envs=65536, num_steps=128: M = 2,097,152
    samples) through the HIP update path and through orc_loss_grad (OpenMP) on the same buffer — the four loss scalars and all
    twelve gradient arrays. (The rollout that fills the buffer is the GPU's; its parity is the C2 test's business.)"""
    nt, k = 65536, 128
    agent = make_agent(crl, nt=nt, k=k, options={"gemm": gemm})
    params = agent.get_params()
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(3); h.adv_stats()
    gs = h.update_minibatch(1, 0.0, apply_update=False)
    g = h.read(F.F_GRADS)
    M = nt * k // 4
    perm = h.read(F.F_PERM)
    g_o, so = O.loss_grad(cfgo, params, h.read(F.F_OBS).reshape(4, -1, order="F"), h.read(F.F_ACTION), h.read(F.F_LOGPROB),
                          h.read(F.F_VALUE), h.read(F.F_ADVANTAGE), h.read(F.F_RETURN), perm[M:2 * M])
    for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
        assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
    assert abs(gs["adv_mean"] - so["adv_mean"]) < 1e-6 and abs(gs["adv_std"] - so["adv_std"]) < 1e-5 * so["adv_std"]
    _grad_close(g, g_o, O.param_offsets(cfgo))
    agent.close()


@pytest.mark.parametrize("nmb", [64, 128, 1024])
def test_many_minibatches_at_tiny_size(crl, nmb):
    """num_minibatches far above the default: every minibatch gets its own advantage statistics (mean/std of ITS slice,
    ppo.jl:221) and its own gradient; nothing borrows scratch sized for 4 minibatches. M = 8192/nmb samples (8 at nmb=1024)."""
    nt, k = 64, 128
    rng = np.random.default_rng(nmb)
    cfgo = O.make_config(num_envs=nt, num_steps=k, num_minibatches=nmb)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    agent = make_agent(crl, nt=nt, k=k, params=params, num_minibatches=nmb)
    st = O.State(cfgo); st.params[:] = params
    _inject_batch(crl, agent, st, rng)
    h = agent.handle
    h.adv_stats()
    M = nt * k // nmb
    for mb in (0, nmb // 2 + 1, nmb - 1):
        gs = h.update_minibatch(mb, 0.0, apply_update=False)
        g_orc, so = O.loss_grad(cfgo, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret,
                                st.perm[mb * M:(mb + 1) * M])
        assert abs(gs["adv_mean"] - so["adv_mean"]) < 1e-6 and abs(gs["adv_std"] - so["adv_std"]) < 1e-5 * so["adv_std"], mb
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (mb, key, gs[key], so[key])
        _grad_close(h.read(crl._lib.F_GRADS), g_orc, O.param_offsets(cfgo))
    agent.close(); st.close()
    with pytest.raises(crl._lib.CrlError):
        make_agent(crl, nt=nt, k=k, num_minibatches=2048)


def test_full_size_update_is_deterministic_and_additive(crl):
    """BASELINE full size (num_envs=65536): the oracle is too slow, so check size-independent properties —
    (1) two runs of the same optimiser step give bit-identical gradients (fixed-order reductions, no float atomics);
    (2) lr = 0 leaves the parameters untouched; (3) the gradient is finite and non-zero."""
    nt, k = 65536, 128
    agent = make_agent(crl, nt=nt, k=k)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(1); h.adv_stats()
    p0 = h.read(F.F_PARAMS)
    # one launch first: the fp16x2 weight-gradient scale (mlp_x2.hpp) is predicted from the previous launch and settles here;
    # with the scale settled, the same inputs must give the same bits
    h.update_minibatch(1, 0.0, apply_update=False, want_stats=False)
    s1 = h.update_minibatch(1, 0.0, apply_update=True); g1 = h.read(F.F_GRADS)
    assert np.array_equal(h.read(F.F_PARAMS), p0)
    s2 = h.update_minibatch(1, 0.0, apply_update=False); g2 = h.read(F.F_GRADS)
    assert np.array_equal(g1, g2) and s1["loss"] == s2["loss"]
    assert np.isfinite(g1).all() and np.linalg.norm(g1) > 0
    # the plain tile striding (update_xcd_align = 0: a tile's actor and critic blocks on different XCDs) sums the same samples in
    # another grouping: same gradient up to float32 summation order
    h.set_option("update_xcd_align", 0)
    h.update_minibatch(1, 0.0, apply_update=False, want_stats=False)
    s3 = h.update_minibatch(1, 0.0, apply_update=False); g3 = h.read(F.F_GRADS)
    assert np.linalg.norm(g3.astype(np.float64) - g1) <= 2e-6 * np.linalg.norm(g1.astype(np.float64)) and loss_close("loss", s3["loss"], s1["loss"], 2e-6)
    h.set_option("update_xcd_align", 1)
    # advantages: a terminal cuts the scan (δ only) — same property as the standalone GAE test, on the resident buffer
    adv = h.read(F.F_ADVANTAGE); term = h.read(F.F_TERMINAL); rew = h.read(F.F_REWARD); val = h.read(F.F_VALUE)
    e, t = np.argwhere(term[:, 1:] == 1)[0]
    assert adv[e, t] == np.float32(np.float64(rew[e, t]) - np.float64(val[e, t]))
    agent.close()


@pytest.mark.parametrize("forced_comm", [False, True])
def test_iteration_with_live_unclipped_value_branch(crl, forced_comm):
    """γ = 0 makes every return 0 or 1 while a critic head bias of 5 puts u = mean(v − R²) ≈ 4 above every clipped term
    (Q4): the whole iteration must still match the oracle. Inside crl_ppo_iterate the speculative pass raises the sticky flag
    and the guard window repeats the iteration from its snapshot with the exact step — on one GPU and, with an RCCL
    communicator (forced 1-rank), with the count and critic-slice all-reduces. (Host-driven single steps keep the in-line
    fix-up: test_update_gradient_matches_oracle[...0.05-True].)"""
    nt, k = 8, 128
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, gamma=0.0)
    agent = crl.Agent(cfg, shuffle_mode=crl._lib.SHUFFLE_FISHER_YATES, options={"comm_force": int(forced_comm)})
    cfgo = O.make_config(num_envs=nt, num_steps=k, gamma=0.0)
    params = agent.get_params()
    params[O.param_offsets(cfgo)[11]] = 5.0
    agent.set_params(params)
    h = agent.handle
    if forced_comm:
        h.comm_init(crl.comm_unique_id(), 1, 0)
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h.env_reset()
    for it in range(2):
        gs = h.iterate(1); os_ = st.iterate(10, gen_perm=True)
        assert max(s["n_unclipped_wins"] for s in os_) > 0, "the test must exercise the u > q branch"
        for a, b in zip(gs, os_):
            assert a["n_unclipped_wins"] == b["n_unclipped_wins"]
            # pg_loss floor of THIS configuration: γ = 0 and a critic bias of 5 make every advantage ≈ −4 with a spread of a few hundredths, so the
            # normalisation (adv − mean) / std amplifies the float32 rounding of the VALUES (1e-7 relative of 5 = 5e-7 absolute, on both sides) by
            # 1 / std: measured |Δ pg_loss| over three iterations 0.8 / 4.6 / 5.9 e-7 with the three-wave rollout kernel and 0.5 / 5.9 / 5.0 e-7 with the
            # six-wave one (scripts/diag_split.py; logprob deviations identical: 1.2e-7 max) — the default floor of 5e-7 sat inside that band
            for key in ("loss", "v_loss", "pg_loss"):
                assert loss_close(key, a[key], b[key], IT_LOSS, floor=1.5e-6), (it, key, a[key], b[key])
        assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < IT_PARAM
    assert h.exact_reruns == 2, "both iterations had to be repeated with the exact value-loss pass (guard window)"
    agent.close(); st.close()


@pytest.mark.parametrize("forced_comm", [False, True])
def test_guard_window_reruns_all_its_iterations(crl, forced_comm):
    """Guard window (single GPU, and with a forced 1-rank RCCL communicator): three iterations are enqueued WITHOUT any read-back; the
    speculation fails in each (γ = 0, critic bias 5). The first host-visible read settles the window: the library restores the
    snapshot (parameters, Adam state, env state, episode accumulators) and repeats all three iterations exactly — parameters
    and episode statistics must equal the oracle's three iterations, and anneal_lr must have followed the rewound counter."""
    nt, k = 8, 128
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, gamma=0.0)
    agent = crl.Agent(cfg, shuffle_mode=crl._lib.SHUFFLE_FISHER_YATES, options={"comm_force": int(forced_comm), "guard_window": 8})
    cfgo = O.make_config(num_envs=nt, num_steps=k, gamma=0.0)
    params = agent.get_params()
    params[O.param_offsets(cfgo)[11]] = 5.0
    agent.set_params(params)
    h = agent.handle
    if forced_comm:
        h.comm_init(crl.comm_unique_id(), 1, 0)
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h.env_reset()
    h.iterate(3, want_stats=False)
    assert h.exact_reruns == 0, "nothing has been read back yet: the window is still open"
    for _ in range(3):
        os_ = st.iterate(10, gen_perm=True)
    assert max(s["n_unclipped_wins"] for s in os_) > 0
    got = h.read(crl._lib.F_PARAMS)            # first host-visible read: settles the window
    assert h.exact_reruns == 3
    assert np.max(np.abs(got - st.params)) < IT_PARAM
    assert np.array_equal(h.read(crl._lib.F_ACTION), st.action) and np.array_equal(h.read(crl._lib.F_ENV_STATE), st.env_state)
    es = h.episode_stats(); n_ep, ret_sum, len_sum = st.episode_stats
    assert (es["episodes"], es["return_sum"], es["length_sum"]) == (n_ep, ret_sum, len_sum)
    assert h.iteration == 3
    # a window that closes by itself (option guard_window iterations) is settled inside crl_ppo_iterate
    agent.close(); st.close()


@pytest.mark.parametrize("then", ["update_minibatch", "env_reset", "set_option"])
def test_host_calls_inside_an_open_guard_window_settle_it_first(crl, then):
    """Two iterations are enqueued without a read-back and their speculation fails (γ = 0, critic bias 5): the guard window is open
    with the sticky flag up. A host-driven step, an env reset or an option change issued NOW must first settle the window (restore +
    exact replay) — otherwise a later settle would restore the snapshot and silently undo the host's call."""
    nt, k = 8, 128
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, gamma=0.0)
    agent = crl.Agent(cfg, shuffle_mode=crl._lib.SHUFFLE_FISHER_YATES)
    cfgo = O.make_config(num_envs=nt, num_steps=k, gamma=0.0)
    params = agent.get_params()
    params[O.param_offsets(cfgo)[11]] = 5.0
    agent.set_params(params)
    h = agent.handle; F = crl._lib
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h.env_reset()
    h.iterate(2, want_stats=False)
    assert h.exact_reruns == 0
    for _ in range(2):
        st.iterate(10, gen_perm=True)
    if then == "update_minibatch":
        gs = h.update_minibatch(0, 1e-4)
        assert h.exact_reruns == 2, "the call had to settle the window before taking its own optimiser step"
        so = st.update_minibatch(0, 1e-4)
        assert loss_close("loss", gs["loss"], so["loss"], IT_LOSS)
        h.sync()
        assert np.max(np.abs(h.read(F.F_PARAMS) - st.params)) < IT_PARAM, "the host-driven step must survive the settle"
    elif then == "env_reset":
        h.env_reset()
        assert h.exact_reruns == 2
        h.sync()
        st2 = O.State(cfgo); st2.env_init()
        assert np.array_equal(h.read(F.F_CUR_OBS), st2.cur_obs), "the reset must not be rolled back by a later settle"
        assert np.max(np.abs(h.read(F.F_PARAMS) - st.params)) < IT_PARAM
        st2.close()
    else:
        h.set_option("rollout_split", 0)
        assert h.exact_reruns == 2 and h.get_option("rollout_split") == 0
        assert np.max(np.abs(h.read(F.F_PARAMS) - st.params)) < IT_PARAM
    assert h.iteration == 2
    agent.close(); st.close()


def test_options_are_validated_and_fallback_is_automatic(crl):
    """crl_ppo_set_option rejects unknown names and out-of-range values; a hidden-layer weight outside the fp16x2 window (|w| >= 255)
    no longer raises an error: the launch runs that role as bf16x3 and the result still matches the oracle."""
    nt, k = 8, 128
    rng = np.random.default_rng(3)
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    off = O.param_offsets(cfgo)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    agent = make_agent(crl, nt=nt, k=k, params=params)
    h = agent.handle
    with pytest.raises(crl.CrlError, match="unknown option"):
        h.set_option("no_such_option", 1)
    with pytest.raises(crl.CrlError, match="outside"):
        h.set_option("gemm", 7)
    assert h.options()["gemm"] == 2 and h.get_option("gemm_fallback_seen") == 0
    st = O.State(cfgo)
    for which, base in (("critic", off[8]), ("actor", off[2])):      # W2 of the critic, then of the actor
        p = params.copy()
        p[base + 5] = 300.0
        agent.set_params(p); st.params[:] = p
        _inject_batch(crl, agent, st, rng)
        h.adv_stats()
        gs = h.update_minibatch(1, 0.0, apply_update=False)
        M = nt * k // 4
        g_o, so = O.loss_grad(cfgo, p, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[M:2 * M])
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (which, key, gs[key], so[key])
        _grad_close(h.read(crl._lib.F_GRADS), g_o, off)
        assert h.get_option("gemm_fallback_seen") == 1
    # the rollout's critic takes the same fallback (value compared at 1e-5), in the one-wave and the three-wave kernel
    p = params.copy()
    p[off[8] + 5] = 300.0
    agent.set_params(p)
    for split in (0, 1, 3):
        h.set_option("rollout_split", split)
        st2 = O.State(cfgo); st2.params[:] = p; st2.env_init()
        h.env_reset(); h.rollout_run(); st2.rollout()
        assert np.array_equal(h.read(crl._lib.F_ACTION), st2.action)
        assert rel_err(h.read(crl._lib.F_VALUE), st2.value) < RTOL
        st2.close()
    agent.close(); st.close()


def test_crl_options_environment_hook_is_read_at_create_and_validated(crl, monkeypatch):
    """CRL_OPTIONS="k=v,…" is the one environment hook for options (bench A/B runs without touching the host code): applied by
    crl_ppo_create before anything is sized, overridden by later crl_ppo_set_option calls, and rejected loudly when malformed."""
    monkeypatch.setenv("CRL_OPTIONS", "gemm=1,guard_window=3,,rollout_stagger=0")
    agent = make_agent(crl, nt=8, k=128)
    o = agent.handle.options()
    assert (o["gemm"], o["guard_window"], o["rollout_stagger"]) == (1, 3, 0) and o["gae_fuse"] == 1
    agent.handle.set_option("gemm", 2)
    assert agent.handle.get_option("gemm") == 2
    agent.close()
    for bad, msg in (("gemm", "key=value"), ("gemm=two", "not an integer"), ("gemm=9", "outside"), ("nope=1", "unknown option")):
        monkeypatch.setenv("CRL_OPTIONS", bad)
        with pytest.raises(crl.CrlError, match=msg):
            make_agent(crl, nt=8, k=128)
    monkeypatch.delenv("CRL_OPTIONS")
    agent = make_agent(crl, nt=8, k=128, options={"guard_window": 2})
    assert agent.handle.get_option("guard_window") == 2 and agent.handle.get_option("gemm") == 2
    agent.close()


def _episodes_from_buffers(reward, terminal, next_done, env_id_offset=0):
    """Episode records of the FIRST rollout after a reset, rebuilt from the buffer: done at step t is terminal[e, t+1]
    (next_done for the last step); return / length accumulate like ppo.jl:125,145."""
    nt, k = reward.shape
    out = []
    for e in range(nt):
        ret = 0.0; length = 0
        for t in range(k):
            ret = np.float32(ret + reward[e, t]); length += 1
            done = terminal[e, t + 1] if t + 1 < k else next_done[e]
            if done:
                out.append((t, env_id_offset + e, float(ret), length)); ret = 0.0; length = 0
    out.sort()
    return out


@pytest.mark.parametrize("kind", ["fused-split", "fused-split6", "fused-single", "wide"])
def test_episode_record_ring(crl, kind, monkeypatch):
    """ppo.jl:147-165 per-episode records (opt-in ring): every episode end of a rollout with its return, length, env and
    step — equal to what the stored rewards / terminals imply; overflow is counted, not stored."""
    L = crl._lib
    nt, k = 70, 128
    if kind == "wide":
        monkeypatch.setenv("CRL_FORCE_WIDE", "1")      # read once, by crl_ppo_create
    agent = make_agent(crl, nt=nt, k=k, env_id_offset=1000, options={"rollout_split": {"fused-split": 1, "fused-split6": 3}.get(kind, 0)})
    h = agent.handle
    with pytest.raises(crl.CrlError, match="not enabled"):
        h._ring_cap = 4; h.episode_records()
    h.episode_ring_enable(4096)
    h.env_reset(); h.rollout_run()
    recs, total = h.episode_records()
    want = _episodes_from_buffers(h.read(L.F_REWARD), h.read(L.F_TERMINAL), h.read(L.F_NEXT_DONE), 1000)
    assert total == len(want) > 100 and recs == want
    es = h.episode_stats()
    assert es["episodes"] == total and es["length_sum"] == sum(r[3] for r in recs)
    h.episode_ring_enable(16)                      # a ring smaller than the episode count keeps counting
    h.env_reset(); h.rollout_run()
    recs2, total2 = h.episode_records()
    assert total2 == total and len(recs2) == 16 and set(recs2) <= set(want)
    h.episode_ring_enable(0)
    agent.close()


@pytest.mark.parametrize("nt,nmb", [(8, 4), (256, 4), (256, 8), (1024, 2), (96, 1), (200, 4)])
def test_iterate_advantage_sums_match_the_permutation(crl, nt, nmb):
    """crl_ppo_iterate gets Σadv, Σadv² of every minibatch (ppo.jl:221) from ONE sequential pass over the advantages: a sample's
    minibatch follows from its blocked-Fisher–Yates bucket, per-sample only inside the few buckets that straddle a boundary
    (shuffle.hip). Cross-check against the permutation itself: sums over adv[b_inds[minibatch]] of the last epoch, in Float64."""
    k = 128
    agent = make_agent(crl, nt=nt, k=k, num_minibatches=nmb)
    h = agent.handle
    h.env_reset()
    h.iterate(1, want_stats=False)
    L = crl._lib
    perm = h.read(L.F_PERM); adv = h.read(L.F_ADVANTAGE).reshape(-1, order="F").astype(np.float64)
    sums = h.read(L.F_ADV_SUMS).reshape(nmb, 2)
    assert sorted(perm.tolist()) == list(range(nt * k))
    M = nt * k // nmb
    for mb in range(nmb):
        a = adv[perm[mb * M:(mb + 1) * M]]
        assert abs(sums[mb, 0] - a.sum()) <= 1e-9 * max(1.0, np.abs(a).sum()), (mb, sums[mb, 0], a.sum())
        assert abs(sums[mb, 1] - (a * a).sum()) <= 1e-9 * max(1.0, (a * a).sum()), mb
    agent.close()


@pytest.mark.parametrize("epochs,nmb", [(1, 2), (3, 8), (6, 1)])
def test_iterate_equals_host_driven_steps_for_other_epoch_and_minibatch_counts(crl, epochs, nmb):
    """crl_ppo_iterate draws all update_epochs permutations up front (one slot, one bucket table per epoch) and takes the advantage
    sums from one sequential pass; the same iteration driven call by call (crl_shuffle / crl_adv_stats / crl_ppo_update_minibatch
    per epoch) must land on the same parameters — for epoch / minibatch counts other than the default 4 x 4."""
    nt, k = 64, 128
    a1 = make_agent(crl, nt=nt, k=k, shuffle_mode=2, update_epochs=epochs, num_minibatches=nmb, anneal_lr=False)
    a2 = make_agent(crl, nt=nt, k=k, shuffle_mode=2, update_epochs=epochs, num_minibatches=nmb, anneal_lr=False, params=a1.get_params())
    h1, h2 = a1.handle, a2.handle
    h1.env_reset(); h2.env_reset()
    h1.iterate(1, want_stats=False)
    h2.rollout_run(); h2.compute_gae()
    assert np.array_equal(h1.read(crl._lib.F_ACTION), h2.read(crl._lib.F_ACTION))
    assert np.array_equal(h1.read(crl._lib.F_ADVANTAGE), h2.read(crl._lib.F_ADVANTAGE))
    for ep in range(epochs):
        h2.shuffle(ep); h2.adv_stats()
        for mb in range(nmb):
            h2.update_minibatch(mb, 2.5e-4, apply_update=True, want_stats=False)
    assert np.array_equal(h1.read(crl._lib.F_PERM), h2.read(crl._lib.F_PERM))          # the last epoch's b_inds
    p1, p2 = h1.read(crl._lib.F_PARAMS), h2.read(crl._lib.F_PARAMS)
    assert np.max(np.abs(p1 - p2)) < 1e-7, np.max(np.abs(p1 - p2))
    a1.close(); a2.close()


def test_ppo_entry_point_emits_the_reference_log_stream(crl, tmp_path):
    """`ppo(config)` (ppo.jl:75) end to end through the host shell: three updates of 8 envs x 32 steps. The records that reach the
    "CleanRL" logger (captured through the JSON-lines sink of Logger.make_logger, logger.jl:7-29) must be what ppo.jl produces for
    the episodes the ORACLE yields from the same parameters and seed: one "Episode Statistics" record per finished episode in
    (step, env) order with global_step as of that step (ppo.jl:124,147-165), then the 16 "Training Statistics" records of the update
    (ppo.jl:246-248); log_step_increment = 0 until the first record, then global_step − last_log_step (ppo.jl:155,246)."""
    import json
    nt, k, updates = 8, 32, 3
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * updates)
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    params = O.orthogonal_params(cfgo, 9)
    crl.ppo(cfg, params=params, shuffle_mode=crl._lib.SHUFFLE_FISHER_YATES, run_name="ppo-entry",
            logger_kw=dict(to_tensorboard=False, to_json=True, log_dir=str(tmp_path)))
    got = [json.loads(line) for line in open(tmp_path / "ppo-entry.json")]

    # what ppo.jl:117-253 logs, from the oracle's buffers
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    want = []
    global_step = 0; last = 0
    ep_ret = np.zeros(nt, np.float32); ep_len = np.zeros(nt, np.int64)
    for _ in range(updates):
        stats = st.iterate(updates, gen_perm=True)
        for t in range(k):
            global_step += nt                                            # ppo.jl:124
            ep_len += 1; ep_ret += st.reward[:, t]                       # ppo.jl:125,145
            done = st.terminal[:, t + 1] if t + 1 < k else st.next_done
            for e in np.flatnonzero(done):                               # ppo.jl:149-163, ascending env
                want.append(("Episode Statistics", dict(episode_return=float(ep_ret[e]), episode_length=int(ep_len[e]), global_step=global_step,
                                                        log_step_increment=0 if last == 0 else global_step - last)))
                ep_ret[e] = 0; ep_len[e] = 0; last = global_step
        for s in stats:                                                  # ppo.jl:246-248
            want.append(("Training Statistics", dict(loss=s["loss"], pg_loss=s["pg_loss"], v_loss=s["v_loss"], entropy_loss=s["entropy_loss"],
                                                     log_step_increment=0 if last == 0 else global_step - last)))
            last = global_step
    st.close()
    assert [g["msg"] for g in got] == [w[0] for w in want], "record names / order"
    n_ep = sum(1 for w in want if w[0] == "Episode Statistics")
    assert n_ep >= 5 and len(got) == n_ep + updates * 16
    for g, (name, w) in zip(got, want):
        if name == "Episode Statistics":
            assert set(g) == {"msg", "episode_return", "episode_length", "global_step", "steps_per_sec", "log_step_increment"}   # ppo.jl:157
            assert (g["episode_return"], g["episode_length"], g["global_step"], g["log_step_increment"]) == \
                   (w["episode_return"], w["episode_length"], w["global_step"], w["log_step_increment"]), (g, w)
            assert g["steps_per_sec"] > 0
        else:
            assert set(g) == {"msg", "loss", "pg_loss", "v_loss", "entropy_loss", "log_step_increment"}                          # ppo.jl:247
            assert g["log_step_increment"] == w["log_step_increment"], (g, w)
            for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                assert loss_close(key, g[key], w[key], IT_LOSS), (key, g[key], w[key])


def test_rollout_steps_track_the_libm_cartpole(crl):
    """The env kernel evaluates sin / cos by a small-angle polynomial (csrc/env.hpp) and the parity oracle shares it bit for bit — so
    "observations bit-exact" above is HIP ≡ the kernel's own polynomial. Against the reference-side statement (libm sinf / cosf,
    orc_cartpole_step_libm — what Julia's sin / cos amount to) every single transition of a 4096-env rollout must stay within 1e-5
    relative (1e-6 absolute floor): s' = step_libm(s, a) from the GPU's OWN state s and action a, for all steps that do not follow a reset."""
    F = crl._lib
    nt, k = 4096, 128
    agent = make_agent(crl, nt=nt, k=k)
    h = agent.handle
    h.env_reset(); h.rollout_run()
    obs, act, term = h.read(F.F_OBS), h.read(F.F_ACTION), h.read(F.F_TERMINAL)
    # transition t -> t+1 of env e is visible in the buffer when no reset happened in between: terminal[e, t] == 0 means obs[:, e, t] is the
    # live state the action was applied to (after a terminal the stored obs is the stale terminal one, Q7, and the env was reset underneath)
    live = term[:, :k - 1] == 0
    e_idx, t_idx = np.nonzero(live)
    s = np.asfortranarray(obs[:, e_idx, t_idx]); a = act[e_idx, t_idx]
    nxt = obs[:, e_idx, t_idx + 1]
    lib_next, _ = O.cartpole_step_batch(s, a, libm=True)
    poly_next, _ = O.cartpole_step_batch(s, a, libm=False)
    assert np.array_equal(poly_next, nxt), "the shared polynomial: bit for bit"
    err = np.abs(nxt.astype(np.float64) - lib_next) / (np.abs(lib_next) + 0.1)
    assert len(e_idx) > 400_000 and err.max() < 1e-5, (len(e_idx), err.max())
    agent.close()


@pytest.mark.parametrize("nt,k", [(64, 128), (256, 100), (8, 9), (4, 1), (132, 37)])
def test_streaming_gae_kernel_is_bit_equal_to_the_oracle(crl, nt, k):
    """gae_stream_kernel (option gae_tile = 4; what batches of 67 M samples or more take by themselves — bench.py roofline_gae.beyond_cache):
    the reference's serial Float64 recurrence (ppo.jl:63-69) in its own order, four envs per thread, chunks of
    eight steps from the end — so every advantage and return must equal orc_gae's bit for bit, for step counts that are no multiple of
    the chunk as well. Driven through a handle's resident buffers (crl_compute_gae, compat mode)."""
    F = crl._lib
    rng = np.random.default_rng(nt * 131 + k)
    agent = make_agent(crl, nt=nt, k=k, num_minibatches=1, options={"gae_tile": 4})
    h = agent.handle
    value = np.asfortranarray((rng.standard_normal((nt, k)) * 10).astype(np.float32))
    reward = np.asfortranarray((rng.random((nt, k)) > 0.02).astype(np.float32))
    term = np.asfortranarray((rng.random((nt, k)) < 0.05).astype(np.uint8))
    nd = (rng.random(nt) < 0.3).astype(np.uint8)
    h.write(F.F_VALUE, value); h.write(F.F_REWARD, reward); h.write(F.F_TERMINAL, term); h.write(F.F_NEXT_DONE, nd)
    h.compute_gae()
    adv_o, ret_o = O.gae_batch(value, reward, term, np.zeros(nt, np.float32), nd, 0.99, 0.95, 0)
    assert np.array_equal(h.read(F.F_ADVANTAGE), adv_o) and np.array_equal(h.read(F.F_RETURN), ret_o)
    h.set_option("gae_tile", 0)
    h.compute_gae()                                   # the segmented kernel on the same inputs: last-bit differences only
    assert rel_err(h.read(F.F_ADVANTAGE), adv_o) < 1e-6
    agent.close()
    if nt == 4:
        with pytest.raises(crl.CrlError, match="num_envs % gae_tile"):
            a2 = make_agent(crl, nt=6, k=8, num_minibatches=1, options={"gae_tile": 4})
            a2.handle.compute_gae()


def _gae_inputs(nt, k, seed):
    rng = np.random.default_rng(seed)
    value = np.asfortranarray((rng.standard_normal((nt, k), dtype=np.float32) * 10))
    reward = np.asfortranarray((rng.random((nt, k), dtype=np.float32) > 0.02).astype(np.float32))
    term = np.asfortranarray((rng.random((nt, k), dtype=np.float32) < 0.02).astype(np.uint8))
    nv = (rng.standard_normal(nt) * 10).astype(np.float32); nd = (rng.random(nt) < 0.3).astype(np.uint8)
    return value, reward, term, nv, nd


@pytest.mark.parametrize("nt_loads", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("nt,k,tile,seg", [(8192, 128, 4, 0), (65536, 128, 4, 0), (65536, 128, 2, 0), (65536, 128, 2, 16), (65536, 128, 1, 4), (1028, 37, 4, 16),
                                           (1028, 37, 2, 8), (65540, 9, 4, 8), (65542, 9, 2, 16), (65541, 3, 1, 8)])
def test_streaming_gae_kernel_many_blocks_both_modes_both_load_flavours(crl, nt, k, tile, seg, mode, nt_loads):
    """gae_stream_kernel with MANY blocks (`blockIdx.x * 256` indexing, ragged last blocks at 1028 / 65540 / 65541 / 65542 envs), every
    envs-per-thread width (gae_tile = 4 / 2 / 1: 16- / 8- / 4-byte accesses) and window depth (gae_seg = 4 / 8 / 16; k = 37 / 9 / 3 leave the
    rolling window partly empty or never full), the cached and the nontemporal flavour, compat mode AND fixed mode with a random
    bootstrap (next_value / next_done through the vector loads that seed vnext / dnext): crl_gae_opt on host arrays must equal orc_gae bit
    for bit — the kernel runs the reference's serial Float64 recurrence (ppo.jl:63-69) in the reference's order."""
    value, reward, term, nv, nd = _gae_inputs(nt, k, nt * 7 + k + mode)
    adv_o, ret_o = O.gae_batch(value, reward, term, nv, nd, 0.99, 0.95, mode)
    adv_g, ret_g = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode, tile=tile, seg=seg, nt_loads=nt_loads)
    assert np.array_equal(adv_g, adv_o) and np.array_equal(ret_g, ret_o)
    if mode == 1:     # the bootstrap is live: a different next_value must move the last column
        adv_2, _ = crl._lib.gae_host(value, reward, term, nv + 1.0, np.zeros_like(nd), 0.99, 0.95, mode, tile=tile, seg=seg, nt_loads=nt_loads)
        assert (adv_2[:, -1] != adv_g[:, -1]).mean() > 0.9


@pytest.mark.parametrize("mode", [0, 1])
def test_automatic_gae_kernel_choice_past_the_infinity_cache_is_bit_equal(crl, mode):
    """(524288, 128) = 2^26 samples, 1.14 GB per launch — the shape bench.py's roofline_gae.beyond_cache times: crl_gae takes the streaming
    kernel BY ITSELF here (gae.hip launch_gae: tile = 0, nt·k >= 2^25, nt >= 262144: two envs per thread, window of 8) with nontemporal
    accesses (crl_gae's own rule from 4 M samples): 1024 blocks, every CU four of them. Bit-equal to orc_gae; the explicit cached flavour as
    well. (262144 x 128, the other beyond_cache row, goes the same way in test_gae_matches_oracle's last case.)"""
    nt, k = 524288, 128
    value, reward, term, nv, nd = _gae_inputs(nt, k, 99 + mode)
    adv_o, ret_o = O.gae_batch(value, reward, term, nv, nd, 0.99, 0.95, mode)
    adv_g, ret_g = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode)
    assert np.array_equal(adv_g, adv_o) and np.array_equal(ret_g, ret_o)
    del adv_g, ret_g
    adv_g, ret_g = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode, tile=0, nt_loads=0)
    assert np.array_equal(adv_g, adv_o) and np.array_equal(ret_g, ret_o)
    # the segmented kernel forced on the same inputs composes affine maps: last-bit differences only (<= 1e-6 of the outputs)
    adv_s, _ = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode, tile=64, nt_loads=1)
    assert np.sum(adv_s != adv_o) <= adv_o.size * 1e-6 and rel_err(adv_s, adv_o) < 1e-6


@pytest.mark.parametrize("nt_loads", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("nt,k,tile,seg", [(65536, 128, 128, 0), (65536, 128, 256, 8), (65536, 128, 128, 16), (1030, 37, 128, 0), (65542, 9, 256, 0), (2, 1, 128, 0),
                                           (4096, 300, 128, 0)])
def test_two_envs_per_thread_segmented_gae_kernel_equals_the_one_env_kernel_bit_for_bit(crl, nt, k, tile, seg, mode, nt_loads):
    """gae_seg2_kernel (gae_tile = 128 / 256: 32 / 64 env PAIRS per block, 8-byte accesses) composes the same affine maps in the same order as gae_kernel: its
    outputs must be bit-identical to the one-env-per-thread segmented kernel's on the same inputs (same segment length), and like those differ from the oracle's
    serial recurrence in the last bit of at most 1e-6 of the outputs. Many blocks, ragged last blocks, both modes with a live bootstrap, both load flavours,
    k not a multiple of the segment length, k > 256 (16-step segments)."""
    value, reward, term, nv, nd = _gae_inputs(nt, k, nt * 3 + k + mode)
    adv_o, ret_o = O.gae_batch(value, reward, term, nv, nd, 0.99, 0.95, mode)
    adv_p, ret_p = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode, tile=tile, seg=seg, nt_loads=nt_loads)
    L = seg if seg in (8, 16) else (8 if k <= 128 else 16)
    adv_1, ret_1 = crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, mode, tile=64, seg=L, nt_loads=nt_loads)
    assert np.array_equal(adv_p, adv_1) and np.array_equal(ret_p, ret_1)
    assert np.sum(adv_p != adv_o) <= max(1, adv_o.size * 1e-6) and rel_err(adv_p, adv_o) < 1e-6 and rel_err(ret_p, ret_o) < 1e-6


def test_gae_opt_rejects_bad_flavours(crl):
    value, reward, term, nv, nd = _gae_inputs(8, 4, 1)
    for kw in (dict(seg=5), dict(tile=3), dict(nt_loads=3)):
        with pytest.raises(crl.CrlError, match="gae_seg is 0"):
            crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, 0, **{**dict(seg=0, tile=8, nt_loads=0), **kw})
    value, reward, term, nv, nd = _gae_inputs(6, 4, 1)
    with pytest.raises(crl.CrlError, match="num_envs % gae_tile"):
        crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, 0, tile=4)
    value, reward, term, nv, nd = _gae_inputs(7, 4, 1)
    with pytest.raises(crl.CrlError, match="even num_envs"):
        crl._lib.gae_host(value, reward, term, nv, nd, 0.99, 0.95, 0, tile=128)


def test_gae_bench_entry_point_times_the_scan_and_its_copy(crl):
    """crl_gae_bench (bench.py roofline_gae.beyond_cache): the standalone scan on synthetic device-resident inputs next to a float4 copy of the
    same byte count; every flavour returns `reps` positive launch times, and a shape the streaming kernel cannot take is an error, not a
    silent fall-back."""
    for tile, seg, ntl in ((4, 0, 0), (4, 16, 1), (2, 4, 1), (1, 8, 0), (64, 16, 0), (0, 0, 1)):
        g, c = crl._lib.gae_bench(4096, 128, seg=seg, tile=tile, nt_loads=ntl, flush_mb=0, reps=3)
        assert len(g) == 3 and len(c) == 3 and min(g) > 0 and min(c) > 0 and max(g) < 50.0
    with pytest.raises(crl.CrlError):
        crl._lib.gae_bench(4098, 128, seg=0, tile=4, nt_loads=0, flush_mb=0, reps=1)


@pytest.mark.parametrize("nt,nmb", [(4096, 4), (4096, 8), (1024, 2)])
def test_advantage_sums_of_every_flavour_equal_the_sums_through_the_permutation(crl, nt, nmb):
    """crl_ppo_iterate's per-minibatch advantage sums (ppo.jl:219-221 needs mean / std of mb_advantages) come from one sequential pass over the advantages that
    looks a sample's minibatch up in the blocked shuffle's bucket table: option adv_seq = 1 (default) with the bucket digit the shuffle's first pass stored per
    sample and epoch, 2 with the digit recomputed (one Philox call), 0 by gathers through the finished permutations. All three must give, to float64 rounding,
    the sums a plain numpy walk through the LAST epoch's permutation gives (the current slot after iterate), and the same run."""
    F = crl._lib
    k = 128
    runs = {}
    for seq in (1, 2, 0):
        agent = make_agent(crl, nt=nt, k=k, num_minibatches=nmb, options={"adv_seq": seq})
        h = agent.handle
        assert h.get_option("adv_seq") == seq
        h.env_reset()
        st = h.iterate(1, want_stats=True)
        adv = h.read(F.F_ADVANTAGE).astype(np.float64).reshape(-1, order="F")
        perm = h.read(F.F_PERM).reshape(-1)
        sums = h.read(F.F_ADV_SUMS).reshape(nmb, 2)
        M = nt * k // nmb
        ref = np.array([[adv[perm[m * M:(m + 1) * M]].sum(), (adv[perm[m * M:(m + 1) * M]] ** 2).sum()] for m in range(nmb)])
        assert np.allclose(sums, ref, rtol=1e-11, atol=1e-9), (seq, sums, ref)
        runs[seq] = (sums, np.asarray([r["loss"] for r in st]), agent.get_params())
        agent.close()
    for seq in (2, 0):
        assert np.allclose(runs[1][0], runs[seq][0], rtol=1e-11, atol=1e-9)
        assert np.allclose(runs[1][1], runs[seq][1], rtol=1e-5, atol=1e-6)       # the loss records of the iteration's 4 x nmb optimiser steps
        assert np.allclose(runs[1][2], runs[seq][2], rtol=1e-5, atol=1e-7)       # parameters after them


def test_fresh_handle_refuses_to_compute_and_library_init_trains(crl):
    """ppo.jl:87 at the C boundary (verdict r5, row b): a handle whose parameters were never written holds zeros — h1 = h2 = 0, every gradient
    except the head biases' is 0 for ever — so every entry point that computes with the networks fails loudly until CRL_F_PARAMS is
    written or crl_ppo_init_params has run; after crl_ppo_init_params the handle holds the reference-shaped start and an iteration
    moves every parameter array."""
    from cleanrl_jl_amd.ppo import _crl_config
    L = crl._lib
    cfg = _crl_config(crl.PPOConfig(num_envs=64, num_steps=16, total_timesteps=64 * 16 * 4))
    h = L.Handle(cfg, 0)
    obs = np.zeros((4, 3), np.float32, order="F")
    for call in (lambda: h.iterate(1), lambda: h.rollout_run(), lambda: h.policy_act(obs, np.zeros(3)),
                 lambda: h.logprob_actions(obs, np.zeros(3, np.int32)), lambda: h.update_minibatch(0, 1e-3)):
        with pytest.raises(crl.CrlError, match="parameters not set"):
            call()
    assert not h.read(L.F_PARAMS).any()                    # nothing ran on them
    h.init_params(5)
    p0 = h.read(L.F_PARAMS)
    assert np.array_equal(p0, L.make_actor_critic_host(4, 2, 64, seed=5))
    h.iterate(1)
    p1 = h.read(L.F_PARAMS)
    from cleanrl_jl_amd import networks
    off = networks.param_offsets(2, 4, [64, 64])
    for i in range(12):
        assert not np.array_equal(p0[off[i]:off[i + 1]], p1[off[i]:off[i + 1]]), "array %d did not move" % i
    h.close()
    # a written vector counts as set too (the Julia shell's set_params! path), on the layer-wise path as well
    cfgw = _crl_config(crl.PPOConfig(num_envs=64, num_steps=16, total_timesteps=64 * 16 * 4), obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
    hw = L.Handle(cfgw, 0)
    with pytest.raises(crl.CrlError, match="parameters not set"):
        hw.iterate(1)
    hw.init_params(1)
    hw.iterate(1)
    assert np.isfinite(hw.read(L.F_PARAMS)).all()
    hw.close()


def test_product_probe_flavours_agree_with_a_float64_product(crl):
    """crl_product_probe (the measurement behind profiles/*_product_error.json) multiplies through the production split functions: every flavour within
    2e-6 relative L2 of a Float64 product on well-scaled operands, per-column scales and K-chunking included."""
    L = crl._lib
    rng = np.random.default_rng(3)
    A = rng.standard_normal((64, 128)).astype(np.float32)
    B = (rng.standard_normal((96, 128)) * 0.5).astype(np.float32)
    ref = B.astype(np.float64) @ A.astype(np.float64).T
    for fl in range(4):
        for chunks in (1, 4):
            got = L.product_probe(fl, A, B, chunks=chunks, scale_a=256.0, scale_b=1024.0).astype(np.float64).sum(0)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6, (fl, chunks)
    cs = np.ldexp(1.0, rng.integers(0, 12, 96)).astype(np.float32)
    got = L.product_probe(0, A, B, col_scale=cs, scale_a=256.0)[0].astype(np.float64)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6
    with pytest.raises(crl.CrlError, match="multiples of 32"):
        L.product_probe(0, A[:33], B)
    med, lo, hi = L.clock_probe(0, 2.0)
    assert 500.0 < lo <= med <= hi < 3000.0, (lo, med, hi)      # MHz: a shader clock, not a tick counter
