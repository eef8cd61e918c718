"""Parity of the generic-shape ("wide", layer-wise MFMA GEMM) HIP path against the CPU oracle: BASELINE configs[2]
(C3: LunarLander-shaped obs 8 / act 4, 2x256 MLP) and other shapes the fused 4/2/64 kernels do not cover.
Same bars as test_gpu_parity.py: bit-exact integer fields, float32 within 1e-5 relative."""
import numpy as np
import pytest

import oraclelib as O
from test_gpu_parity import ATOL, RTOL, _grad_close, crl, loss_close, rel_err  # noqa: F401  (crl is the module fixture)

pytestmark = pytest.mark.gpu

SHAPES = [(8, 4, 256), (6, 3, 128), (4, 2, 64), (17, 16, 64), (3, 5, 256)]


def rel_err_s(a, b):
    """rel_err with the absolute floor scaled to the array's magnitude: |a-b| <= RTOL*|b| + ATOL*max(1, max|b|).
    Log-probabilities here come from logits spread to ±5 (the actor head is scaled x30 so that all 4 actions occur): they
    are differences of numbers that large, and float32 summation-order noise on a 256-term dot product (both the
    v_mfma_f32 and the bf16x3 flavour measure ~4e-7 mean, 9e-6 max absolute) is relative to THAT scale."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    floor = ATOL / RTOL * max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    return np.max(np.abs(a - b) / (np.abs(b) + floor)) if a.size else 0.0


def make_wide(crl, nt, k, D, A, Hd, params=None, **kw):
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, **{a: b for a, b in kw.items() if a in
                        ("num_minibatches", "update_epochs", "clip_value_loss", "anneal_lr", "lr")})
    shape = {a: b for a, b in kw.items() if a in ("gae_mode", "shuffle_mode", "stale_obs", "env_id_offset", "seed", "options")}
    return crl.Agent(cfg, params=params, obs_dim=D, n_act=A, hidden=Hd, env_kind=crl._lib.ENV_SYNTHETIC, **shape)


def ocfg(nt, k, D, A, Hd, **kw):
    return O.make_config(num_envs=nt, num_steps=k, obs_dim=D, n_act=A, hidden=Hd, env_kind=1, **kw)


def spread_params(cfg, seed):
    rng = np.random.default_rng(seed)
    params = O.orthogonal_params(cfg, seed) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    off = O.param_offsets(cfg)
    params[off[4]:off[5]] *= 30   # spread the logits: every action occurs
    return params


@pytest.fixture(autouse=True)
def force_wide(monkeypatch):
    monkeypatch.setenv("CRL_FORCE_WIDE", "1")   # sends the 4/2/64 shape down the generic path too


@pytest.mark.parametrize("D,A,Hd", SHAPES)
@pytest.mark.parametrize("n", [1, 33, 300])
def test_wide_policy_act_and_logprob_match_oracle(crl, D, A, Hd, n):
    rng = np.random.default_rng(n + D)
    cfg = ocfg(8, 16, D, A, Hd)
    params = spread_params(cfg, 3)
    agent = make_wide(crl, 8, 16, D, A, Hd, params=params)
    obs = np.asfortranarray(rng.standard_normal((D, n)).astype(np.float32))
    u = rng.random(n)
    a_o, lp_o, v_o, margin = O.get_action(cfg, params, obs, u)
    a_g, lp_g, v_g = agent.handle.policy_act(obs, u)
    safe = margin > 1e-6
    assert np.array_equal(a_g[safe], a_o[safe]), "action indices must be bit-exact away from CDF knots"
    assert safe.mean() > 0.98 and len(set(a_o.tolist())) >= min(A, 2 if n > 1 else 1)
    same = a_g == a_o
    assert rel_err_s(lp_g[same], lp_o[same]) < RTOL and rel_err(v_g, v_o) < RTOL
    acts = rng.integers(0, A, n).astype(np.int32)
    lp_o2, ent_o = O.logprob_actions(cfg, params, obs, acts)
    lp_g2, ent_g = crl.logprob_actions(obs, agent.actor, acts + 1)
    assert ent_g.shape == (A, n) and rel_err_s(lp_g2, lp_o2) < RTOL and rel_err(ent_g, ent_o) < RTOL
    agent.close()


@pytest.mark.parametrize("D,A,Hd,nt,k", [(8, 4, 256, 70, 16), (6, 3, 128, 8, 32), (4, 2, 64, 33, 8), (8, 4, 256, 192, 24), (16, 8, 256, 64, 16), (3, 2, 256, 128, 8)])
def test_wide_rollout_matches_oracle(crl, D, A, Hd, nt, k):
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 5)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params)
    st = O.State(cfg); st.params[:] = params; st.env_init()
    h = agent.handle; F = crl._lib
    h.env_reset()
    assert np.array_equal(h.read(F.F_CUR_OBS), st.cur_obs)
    h.rollout_run(); st.rollout()
    assert np.array_equal(h.read(F.F_ACTION), st.action), f"{np.sum(h.read(F.F_ACTION) != st.action)} actions differ"
    assert np.array_equal(h.read(F.F_OBS), st.obs) and np.array_equal(h.read(F.F_REWARD), st.reward)
    assert np.array_equal(h.read(F.F_TERMINAL), st.terminal) and np.array_equal(h.read(F.F_NEXT_DONE), st.next_done)
    assert rel_err_s(h.read(F.F_LOGPROB), st.logprob) < RTOL and rel_err(h.read(F.F_VALUE), st.value) < RTOL
    es = h.episode_stats(); n_ep, ret_sum, len_sum = st.episode_stats
    assert es["episodes"] == n_ep and es["length_sum"] == len_sum and abs(es["return_sum"] - ret_sum) < 1e-4 * max(1, abs(ret_sum))
    h.compute_gae(); st.compute_gae()
    assert rel_err(h.read(F.F_ADVANTAGE), st.adv) < RTOL and rel_err(h.read(F.F_RETURN), st.ret) < RTOL
    agent.close(); st.close()


def inject(crl, agent, st, rng, D, A, ret_scale=10.0):
    nt, k = st.cfg.num_envs, st.cfg.num_steps
    st.obs[:] = rng.standard_normal((D, nt, k)).astype(np.float32)
    st.action[:] = rng.integers(0, A, (nt, k))
    st.logprob[:] = (np.log(1.0 / A) + 0.3 * rng.standard_normal((nt, k))).astype(np.float32)
    st.value[:] = rng.standard_normal((nt, k)).astype(np.float32) * (1.0 if ret_scale > 1 else 0.05)
    st.adv[:] = (2 * rng.standard_normal((nt, k))).astype(np.float32)
    st.ret[:] = (ret_scale * rng.standard_normal((nt, k))).astype(np.float32)
    st.perm[:] = rng.permutation(nt * k).astype(np.int32)
    h = agent.handle; F = crl._lib
    for f, a in ((F.F_OBS, st.obs), (F.F_ACTION, st.action), (F.F_LOGPROB, st.logprob), (F.F_VALUE, st.value),
                 (F.F_ADVANTAGE, st.adv), (F.F_RETURN, st.ret), (F.F_PERM, st.perm)):
        h.write(f, a)


@pytest.mark.parametrize("D,A,Hd,nt,k,ret_scale,clipv", [
    (8, 4, 256, 8, 128, 10.0, True), (8, 4, 256, 8, 128, 0.05, True), (8, 4, 256, 37, 64, 3.0, False),
    (6, 3, 128, 16, 64, 10.0, True), (4, 2, 64, 8, 128, 0.05, True), (17, 16, 64, 64, 36, 10.0, True),
    (3, 5, 256, 5, 520, 10.0, True)])
def test_wide_update_gradient_matches_oracle(crl, D, A, Hd, nt, k, ret_scale, clipv):
    """ret_scale = 0.05 drives u = mean(v - R²) > 0 (Q4): here the exact count is known before the loss kernel runs."""
    rng = np.random.default_rng(nt + k + D)
    cfg = ocfg(nt, k, D, A, Hd, clip_value_loss=clipv)
    params = O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    off = O.param_offsets(cfg)
    if ret_scale < 1:
        params[off[11]] = 0.3
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, clip_value_loss=clipv)
    st = O.State(cfg); st.params[:] = params
    inject(crl, agent, st, rng, D, A, ret_scale)
    h = agent.handle
    h.adv_stats()
    M = nt * k // 4
    for mb in (0, 3):
        gs = h.update_minibatch(mb, 2.5e-4, apply_update=False)
        g_gpu = h.read(crl._lib.F_GRADS)
        g_orc, so = O.loss_grad(cfg, params, st.obs.reshape(D, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret,
                                st.perm[mb * M:(mb + 1) * M])
        if ret_scale < 1 and clipv:
            assert so["n_unclipped_wins"] > 0 and gs["n_unclipped_wins"] == so["n_unclipped_wins"]
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
        _grad_close(g_gpu, g_orc, off)
    agent.close(); st.close()


@pytest.mark.parametrize("opts", [
    {"wide_fuse": 0}, {"wide_fuse": 1}, {"wide_fuse": 2}, {"wide_fuse": 3},
    {"wide_fuse": 3, "wide_fuse_pc": 0}, {"wide_fuse": 2, "wide_fuse_pc": 0}, {"wide_fuse": 3, "wide_wgrad_full": 0},
    {"wide_fuse": 3, "shuffle_overlap": 0}, {"wide_fuse": 3, "wide_fwd_wbufs": 3, "wide_rs": 0}, {"wide_fuse": 3, "wide_fwd_wbufs": 0, "wide_rs": 0},
    {"wide_fuse": 3, "wide_d2_split": 0}, {"wide_fuse": 3, "wide_rs": 0}, {"wide_fuse": 3, "wide_rs": 1}, {"wide_fuse": 3, "wide_rs": 8}, {"wide_fuse": 3, "wide_rs": 9}, {"wide_fuse": 3, "wide_rs": 11}, {"wide_fuse": 3, "wide_rs": 27}], ids=lambda o: ",".join(f"{a}={b}" for a, b in o.items()))
@pytest.mark.parametrize("D,A,nt", [(8, 4, 24), (16, 8, 16), (3, 2, 12)])
def test_every_2x256_kernel_flavour_matches_the_oracle(crl, opts, D, A, nt):
    """The 2x256 shape has four selectable pipelines (option wide_fuse: 0 layer-wise GEMMs, 1 tile-resident forward, 2 + tile-
    resident backward, 3 + h1 never stored) and the flavours under them (producer/consumer forward, the two weight-gradient
    kernels, dW3 on the side stream, δ2 handed to the weight gradient as f32 instead of as the backward's fp16x2 planes; since round 6 the
    forward and backward of wide_fuse = 3 are the register-stationary wide_rs_fwd_kernel / wide_rs_bwd_kernel — option wide_rs bits 0 and 3,
    obs_dim a multiple of 4 — and wide_rs = 0 keeps round 5's kernels and their flavours alive).  Every one of them must give the oracle's loss scalars and gradient on the same
    buffers — the default is only the fastest of equals.  Shapes: C3's, the largest the fused kernels take (obs 16, 8
    actions) and an odd small one (obs 3, 2 actions)."""
    k, Hd = 128, 256
    rng = np.random.default_rng(nt + D)
    cfg = ocfg(nt, k, D, A, Hd)
    params = O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params)
    h = agent.handle
    for name, v in opts.items():
        h.set_option(name, v)
    st = O.State(cfg); st.params[:] = params
    inject(crl, agent, st, rng, D, A, 3.0)
    h.adv_stats()
    M = nt * k // 4
    assert M % 128 == 0
    off = O.param_offsets(cfg)
    for mb in (0, 3):
        gs = h.update_minibatch(mb, 2.5e-4, apply_update=False)
        g_gpu = h.read(crl._lib.F_GRADS)
        g_orc, so = O.loss_grad(cfg, params, st.obs.reshape(D, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret,
                                st.perm[mb * M:(mb + 1) * M])
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
        _grad_close(g_gpu, g_orc, off)
    agent.close(); st.close()


@pytest.mark.parametrize("split", [1, 0])
def test_weight_gradient_from_split_planes_with_cotangents_of_very_different_size(crl, split):
    """Option wide_d2_split = 1 (default): the backward kernel hands δ2 to the weight-gradient kernel as fp16x2 pieces scaled by a power of two PER SAMPLE, and the
    weight-gradient kernel (wide_wgrad_split_kernel) puts 1/scale onto its other operand, relative to the chunk's largest. Here the samples' cotangents span nine
    orders of magnitude (advantages and return errors from 1e-6 to 1e3, a block of exact zeros among them): the gradient still matches the oracle at the usual bar,
    with both flavours — the large samples dominate the sum, and the small ones must neither overflow, underflow to NaN nor disturb it."""
    D, A, Hd, nt, k = 8, 4, 256, 32, 128
    rng = np.random.default_rng(77)
    cfg = ocfg(nt, k, D, A, Hd, clip_value_loss=False)
    params = O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, clip_value_loss=False)
    h = agent.handle
    h.set_option("wide_d2_split", split)
    st = O.State(cfg); st.params[:] = params
    inject(crl, agent, st, rng, D, A, 3.0)
    mag = (10.0 ** rng.uniform(-6, 3, (nt, k))).astype(np.float32)
    mag[:2, :] = 0.0                                              # two envs' worth of samples with advantage 0 and return == value prediction scale 0
    st.adv[:] = (st.adv * mag).astype(np.float32)
    st.ret[:] = (st.value + (st.ret - st.value) * mag).astype(np.float32)
    F = crl._lib
    h.write(F.F_ADVANTAGE, st.adv); h.write(F.F_RETURN, st.ret)
    # the advantages are normalised per minibatch (ppo.jl:219): the spread survives it (mean and std are set by the largest samples)
    h.adv_stats()
    M = nt * k // 4
    off = O.param_offsets(cfg)
    for mb in (0, 2):
        gs = h.update_minibatch(mb, 2.5e-4, apply_update=False)
        g_gpu = h.read(F.F_GRADS)
        assert np.isfinite(g_gpu).all()
        g_orc, so = O.loss_grad(cfg, params, st.obs.reshape(D, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[mb * M:(mb + 1) * M])
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
        _grad_close(g_gpu, g_orc, off)
    agent.close(); st.close()


def test_wide_fp16x2_weight_scale_follows_the_weights(crl):
    """The 256-wide fp16x2 products stage W2 with a power of two taken from the largest |w| of the network at every optimiser step
    (wide_w2scale_kernel), so a weight of 300 — outside the fixed 2^8 window of rounds 1-2, where it raised an error — and a network
    whose weights are all tiny both run on the fp16x2 path at full accuracy: gradient, rollout actions and values match the oracle,
    nothing raises. (Bars: 1e-5 as everywhere; 3e-5 on the gradient arrays of the 300-weight case, whose pre-activations are 300 times
    larger and carry float32 summation-order noise in proportion on both sides.)"""
    D, A, Hd, nt, k = 8, 4, 256, 8, 128
    rng = np.random.default_rng(5)
    cfg = ocfg(nt, k, D, A, Hd)
    base = O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    off = O.param_offsets(cfg)
    agent = make_wide(crl, nt, k, D, A, Hd, params=base)
    h = agent.handle; F = crl._lib
    st = O.State(cfg)
    for case in ("big", "tiny", "plain"):
        params = base.copy()
        if case == "big":
            params[off[8] + 7] = 300.0               # critic W2
            params[off[2] + 11] = -300.0             # actor W2
        elif case == "tiny":
            params[off[2]:off[3]] *= 1e-4; params[off[8]:off[9]] *= 1e-4
        agent.set_params(params); st.params[:] = params
        inject(crl, agent, st, rng, D, A, 10.0)
        h.adv_stats()
        M = nt * k // 4
        gs = h.update_minibatch(2, 0.0, apply_update=False)
        g_o, so = O.loss_grad(cfg, params, st.obs.reshape(D, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[2 * M:3 * M])
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (case, key, gs[key], so[key])
        _grad_close(h.read(F.F_GRADS), g_o, off, tol=3e-5 if case == "big" else RTOL)
        h.sync()                                     # no error is pending
    params = base.copy(); params[off[8] + 7] = 300.0
    agent.set_params(params)
    st2 = O.State(cfg); st2.params[:] = params; st2.env_init()
    h.env_reset(); h.rollout_run(); st2.rollout()
    assert np.array_equal(h.read(F.F_ACTION), st2.action) and rel_err(h.read(F.F_VALUE), st2.value) < RTOL
    agent.close(); st.close(); st2.close()


def test_wide_and_fused_paths_agree_on_the_cartpole_shape(crl, monkeypatch):
    """Same 4/2/64 minibatch through update.hip (fused, bf16x3) and wide.hip (layer-wise f32 MFMA)."""
    from test_gpu_parity import _inject_batch, make_agent
    rng = np.random.default_rng(9)
    nt, k = 64, 128
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    grads = {}
    for force in ("0", "1"):
        monkeypatch.setenv("CRL_FORCE_WIDE", force)
        agent = make_agent(crl, nt=nt, k=k, params=params)
        st = O.State(cfgo); st.params[:] = params
        _inject_batch(crl, agent, st, np.random.default_rng(10))
        agent.handle.adv_stats()
        gs = agent.handle.update_minibatch(1, 2.5e-4, apply_update=True)
        grads[force] = (agent.handle.read(crl._lib.F_GRADS).copy(), gs, agent.handle.read(crl._lib.F_PARAMS).copy())
        agent.close(); st.close()
    _grad_close(grads["1"][0], grads["0"][0], O.param_offsets(cfgo), tol=2e-5)
    assert abs(grads["1"][1]["loss"] - grads["0"][1]["loss"]) < 1e-5
    assert np.max(np.abs(grads["1"][2] - grads["0"][2])) < 1e-6


@pytest.mark.parametrize("D,A,Hd,nt,k", [(8, 4, 256, 16, 32), (4, 2, 64, 8, 128)])
def test_wide_full_iteration_matches_oracle(crl, D, A, Hd, nt, k):
    """Whole ppo.jl:117-253 loop body on the synthetic env, exact Fisher–Yates, two iterations."""
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 7)
    off = O.param_offsets(cfg)
    params[off[4]:off[5]] /= 10
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=0)
    st = O.State(cfg); st.params[:] = params; st.env_init()
    h = agent.handle; F = crl._lib
    h.env_reset()
    for it in range(2):
        gs = h.iterate(1)
        os_ = st.iterate(10, gen_perm=True)
        assert np.array_equal(h.read(F.F_PERM), st.perm)
        acts = h.read(F.F_ACTION)
        assert np.array_equal(acts, st.action), f"iteration {it}: {np.sum(acts != st.action)} actions differ"
        assert rel_err(h.read(F.F_ADVANTAGE), st.adv) < RTOL
        for a, b in zip(gs, os_):
            for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                assert loss_close(key, a[key], b[key], RTOL), (it, key, a[key], b[key])
        assert np.max(np.abs(h.read(F.F_PARAMS) - st.params)) < 1e-5
    agent.close(); st.close()


@pytest.mark.parametrize("rs", [0, 2, 6])
@pytest.mark.parametrize("env,D,A,nt,k", [("synthetic", 8, 4, 128, 40), ("synthetic", 16, 8, 64, 24), ("synthetic", 5, 3, 64, 16), ("cartpole", 4, 2, 128, 160)])
def test_2x256_rollout_flavours_match_the_oracle(crl, rs, env, D, A, nt, k):
    """Option wide_rs bit 1 (default): the rollout of a 2x256 shape keeps the ACTOR's weights in the eight waves' registers for all num_steps steps
    (wide_rs_rollout_kernel, env state and observations in LDS) and evaluates the critic over the stored observations as one batched forward behind it
    (bit 2: on the register-stationary forward instead of the producer / consumer kernel); wide_rs = 0 is round 5's wide_rollout_pc_kernel. All three
    against the oracle: actions, observations, rewards, terminals, episode statistics exact, log-probabilities / values / advantages at 1e-5 —
    on the synthetic env (obs 8 / 16 / an odd 5) and on CartPole itself (160 steps: episodes end and reset inside the launch; the state that
    lives in LDS during the launch must come back to global memory for the next one, so TWO rollouts are compared)."""
    Hd = 256
    cart = env == "cartpole"
    cfg = O.make_config(num_envs=nt, num_steps=k, obs_dim=D, n_act=A, hidden=Hd, env_kind=0 if cart else 1)
    params = spread_params(cfg, 5)
    if cart:
        off = O.param_offsets(cfg); params[off[4]:off[5]] /= 10     # logits ±0.5: both actions occur and episodes last
    pc = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10)
    agent = crl.Agent(pc, params=params, obs_dim=D, n_act=A, hidden=Hd, env_kind=crl._lib.ENV_CARTPOLE if cart else crl._lib.ENV_SYNTHETIC,
                      options={"wide_rs": rs})
    st = O.State(cfg); st.params[:] = params; st.env_init()
    h = agent.handle; F = crl._lib
    h.env_reset()
    for it in range(2):
        h.rollout_run(); st.rollout()
        assert np.array_equal(h.read(F.F_ACTION), st.action), f"rollout {it}: {np.sum(h.read(F.F_ACTION) != st.action)} actions differ"
        assert np.array_equal(h.read(F.F_OBS), st.obs) and np.array_equal(h.read(F.F_REWARD), st.reward)
        assert np.array_equal(h.read(F.F_TERMINAL), st.terminal) and np.array_equal(h.read(F.F_NEXT_DONE), st.next_done)
        assert np.array_equal(h.read(F.F_CUR_OBS), st.cur_obs)
        assert rel_err_s(h.read(F.F_LOGPROB), st.logprob) < RTOL and rel_err(h.read(F.F_VALUE), st.value) < RTOL
        es = h.episode_stats(); n_ep, ret_sum, len_sum = st.episode_stats
        assert es["episodes"] == n_ep and es["length_sum"] == len_sum and abs(es["return_sum"] - ret_sum) < 1e-4 * max(1, abs(ret_sum))
        if cart:
            assert n_ep > 0, "the test must see episodes end inside the launch"
        h.compute_gae(); st.compute_gae()
        assert rel_err(h.read(F.F_ADVANTAGE), st.adv) < RTOL and rel_err(h.read(F.F_RETURN), st.ret) < RTOL
    agent.close(); st.close()


@pytest.mark.parametrize("persist", [2, 1, 0, {"wide_rs_actor_pct": 60}, {"wide_rs": 11}, {"wide_rs": 0}], ids=str)
def test_c3_shaped_iteration_at_1024_envs_matches_oracle(crl, persist):
    """(wide_rollout_persist = 2: the rollout as one launch — since round 6 the register-stationary wide_rs_rollout_kernel, with wide_rs = 0 round 5's
    producer / consumer wide_rollout_pc_kernel; 1: one launch, wide_rollout_persist_kernel; 0: three launches per step. The dict cases: the
    register-stationary backward with the CUs split 60 : 40 between the actor's and the critic's blocks — 128 tiles per network here, the smallest
    launch that takes the uneven split —, with the dW3 sweeps still launched, and round 5's kernels throughout.)
    BASELINE configs[2]'s shape (obs 8 / act 4 / 2x256, synthetic env) at num_envs = 1024 — 32 tiles per launch, so the multi-tile
    paths of the layer-wise kernels run (chunked weight gradients, several blocks per GEMM) — for one whole iteration against the
    oracle at the north_star bar: actions and permutation bit-equal, advantages / losses within 1e-5 relative, parameters within 1e-5
    relative L2 per array and 1e-5 absolute."""
    D, A, Hd, nt, k = 8, 4, 256, 1024, 16
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 9)
    off = O.param_offsets(cfg)
    params[off[4]:off[5]] /= 10
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=0, options=persist if isinstance(persist, dict) else {"wide_rollout_persist": persist})
    st = O.State(cfg); st.params[:] = params; st.env_init()
    h = agent.handle; F = crl._lib
    h.env_reset()
    gs = h.iterate(1)
    os_ = st.iterate(10, gen_perm=True)
    assert np.array_equal(h.read(F.F_PERM), st.perm)
    assert np.array_equal(h.read(F.F_ACTION), st.action)
    assert np.array_equal(h.read(F.F_OBS), st.obs) and np.array_equal(h.read(F.F_REWARD), st.reward) and np.array_equal(h.read(F.F_TERMINAL), st.terminal)
    assert rel_err(h.read(F.F_LOGPROB), st.logprob) < RTOL and rel_err(h.read(F.F_VALUE), st.value) < RTOL
    assert rel_err(h.read(F.F_ADVANTAGE), st.adv) < RTOL
    for a, b in zip(gs, os_):
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, a[key], b[key], RTOL), (key, a[key], b[key])
    pg, po = h.read(F.F_PARAMS).astype(np.float64), st.params.astype(np.float64)
    for i in range(12):
        err = np.linalg.norm(pg[off[i]:off[i + 1]] - po[off[i]:off[i + 1]]) / max(np.linalg.norm(po[off[i]:off[i + 1]]), 1e-12)
        assert err < RTOL, (i, err)
    assert np.max(np.abs(pg - po)) < 1e-5
    agent.close(); st.close()


def test_wide_rejects_unsupported_shapes(crl):
    for kw, msg in ((dict(hidden=96), "hidden"), (dict(n_act=17, hidden=256), "n_act"), (dict(obs_dim=65, hidden=256), "obs_dim")):
        with pytest.raises(crl.CrlError, match=msg):
            crl.Agent(crl.PPOConfig(), env_kind=crl._lib.ENV_SYNTHETIC, **kw)
    with pytest.raises(crl.CrlError, match="CARTPOLE"):
        crl.Agent(crl.PPOConfig(), obs_dim=8, n_act=4, hidden=256)       # CartPole dynamics need obs 4 / act 2


def test_c3_full_size_properties(crl):
    """BASELINE configs[2] at full size (num_envs=16384, num_steps=128, obs 8 / act 4 / 2x256): the oracle needs minutes
    here, so: run-to-run bit-identical gradients (fixed-order reductions), lr = 0 leaves the parameters untouched,
    finite non-zero gradient, and the rollout's integer fields equal the oracle's on the first 64 envs' first steps."""
    nt, k, D, A, Hd = 16384, 128, 8, 4, 256
    cfg = ocfg(64, 4, D, A, Hd)
    params = spread_params(cfg, 11)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=1)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(1); h.adv_stats()
    st = O.State(cfg); st.params[:] = params; st.env_init(); st.rollout()
    acts = h.read(F.F_ACTION)
    assert np.array_equal(acts[:64, :4], st.action) and np.array_equal(h.read(F.F_OBS)[:, :64, :4], st.obs)
    p0 = h.read(F.F_PARAMS)
    s1 = h.update_minibatch(1, 0.0, apply_update=True); g1 = h.read(F.F_GRADS)
    assert np.array_equal(h.read(F.F_PARAMS), p0)
    s2 = h.update_minibatch(1, 0.0, apply_update=False); g2 = h.read(F.F_GRADS)
    assert np.array_equal(g1, g2) and s1["loss"] == s2["loss"]
    assert np.isfinite(g1).all() and np.linalg.norm(g1) > 0
    agent.close(); st.close()


def test_c3_full_size_rollout_slices_match_the_oracle(crl):
    """BASELINE configs[2] at full size: the whole 16384-env x 128-step rollout of the full-grid `wide_rs_rollout_kernel` (256 blocks of 64
    envs, one per CU, and the batched critic pass behind it — the default for this shape since round 6) against the oracle on three 64-env slices — the first block, one in the middle of the
    grid and the last — for ALL 128 steps: the oracle rolls envs [o, o + 64) out under `env_id_offset = o` (global env ids key every
    random stream, so a slice is independent of the other envs). Actions exact under the knot-margin rule (a draw within 1e-6 of a CDF
    knot may flip; the synthetic env's observations do not depend on the action, so nothing else moves), observations / rewards /
    terminals bit-equal, logprob / value / advantages / returns 1e-5."""
    nt, k, D, A, Hd = 16384, 128, 8, 4, 256
    cfg_full = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg_full, 17)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params)
    h = agent.handle; F = crl._lib
    assert h.get_option("wide_rollout_persist") == 2 and h.get_option("wide_fuse") == 3 and h.get_option("wide_rs") == 27, "defaults must select wide_rs_rollout_kernel"
    h.env_reset(); h.rollout_run(); h.compute_gae()
    act, obs, rew, term = h.read(F.F_ACTION), h.read(F.F_OBS), h.read(F.F_REWARD), h.read(F.F_TERMINAL)
    lp, val, adv, ret = h.read(F.F_LOGPROB), h.read(F.F_VALUE), h.read(F.F_ADVANTAGE), h.read(F.F_RETURN)
    nd = h.read(F.F_NEXT_DONE)
    for o in (0, 8160, 16320):
        sl = slice(o, o + 64)
        cfg = ocfg(64, k, D, A, Hd, env_id_offset=o)
        st = O.State(cfg); st.params[:] = params; st.env_init(); st.rollout(); st.compute_gae()
        diff = act[sl] != st.action
        for e, t in zip(*np.nonzero(diff)):
            u = O.lib().orc_u53(cfg.seed, int(o + e), int(t), 0)
            _, _, _, m = O.get_action(cfg, params, st.obs[:, e, t].reshape(-1, 1), np.array([u]))
            assert m[0] <= 1e-6, f"slice {o} env {e} step {t}: action differs although the draw is {m[0]:.3e} away from the CDF knot"
        assert diff.mean() < 1e-4
        assert np.array_equal(obs[:, sl], st.obs) and np.array_equal(rew[sl], st.reward) and np.array_equal(term[sl], st.terminal)
        assert np.array_equal(nd[sl], st.next_done)
        same = ~diff
        assert rel_err_s(lp[sl][same], st.logprob[same]) < RTOL and rel_err(val[sl], st.value) < RTOL
        assert rel_err(adv[sl], st.adv) < RTOL and rel_err(ret[sl], st.ret) < RTOL
        st.close()
    agent.close()


def test_c3_size_minibatch_gradient_matches_oracle(crl):
    """BASELINE configs[2] at full size, directly: ONE minibatch (num_envs=16384, num_steps=128, obs 8 / act 4 / 2x256: M = 524,288
    samples, 4.3e11 flop) through the layer-wise HIP path and through orc_loss_grad (OpenMP) on the same buffer — four loss
    scalars and all twelve gradient arrays."""
    nt, k, D, A, Hd = 16384, 128, 8, 4, 256
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 13)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=1)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(2); h.adv_stats()
    gs = h.update_minibatch(2, 0.0, apply_update=False)
    g = h.read(F.F_GRADS)
    M = nt * k // 4
    perm = h.read(F.F_PERM)
    g_o, so = O.loss_grad(cfg, params, h.read(F.F_OBS).reshape(D, -1, order="F"), h.read(F.F_ACTION), h.read(F.F_LOGPROB),
                          h.read(F.F_VALUE), h.read(F.F_ADVANTAGE), h.read(F.F_RETURN), perm[2 * M:3 * M])
    for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
        assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
    _grad_close(g, g_o, O.param_offsets(cfg), tol=RTOL)
    agent.close()


@pytest.mark.parametrize("D,A,nt,k,rs", [(16, 8, 4096, 128, 27), (16, 8, 4096, 128, 11), (12, 6, 2048, 128, 27), (8, 4, 6400, 128, 27)])
def test_2x256_many_tiles_per_block_other_shapes(crl, D, A, nt, k, rs):
    """The register-stationary kernels away from C3's own shape, with enough tiles per block for their buffers to rotate: obs 16 / 8 actions at
    M = 131,072 (32 tiles per block: the forward's 8-tile observation chunks go round their three LDS buffers, the 16-float rows take the second
    DMA piece pair), obs 12 / 6 actions (16 tiles per block, a row width between the two piece pairs), and C3's shape at 6400 envs —
    M = 204,800: 50 tiles per block, not a multiple of the 16-tile chunk, the CUs split 133 : 123 between the networks — against orc_loss_grad
    on the rollout's own buffers. wide_rs = 11 keeps the dW3 sweeps (the backward without its dW3 part)."""
    Hd = 256
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 21)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=1, options={"wide_rs": rs})
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(1); h.adv_stats()
    gs = h.update_minibatch(1, 0.0, apply_update=False)
    g = h.read(F.F_GRADS)
    M = nt * k // 4
    perm = h.read(F.F_PERM)
    g_o, so = O.loss_grad(cfg, params, h.read(F.F_OBS).reshape(D, -1, order="F"), h.read(F.F_ACTION), h.read(F.F_LOGPROB),
                          h.read(F.F_VALUE), h.read(F.F_ADVANTAGE), h.read(F.F_RETURN), perm[1 * M:2 * M])
    for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
        assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
    _grad_close(g, g_o, O.param_offsets(cfg), tol=RTOL)
    agent.close()


def test_wide_rccl_path_world1(crl):
    """C3 shape with a forced 1-rank RCCL communicator: the gradient message, the advantage statistics AND the two extra
    value-loss scalars (Σ(v − R²), #{u > q}) travel through ncclAllReduce; a sum over one rank is the identity."""
    D, A, Hd, nt, k = 8, 4, 256, 16, 32
    cfg = ocfg(nt, k, D, A, Hd)
    params = spread_params(cfg, 7)
    agent = make_wide(crl, nt, k, D, A, Hd, params=params, shuffle_mode=0)
    h = agent.handle
    h.set_option("comm_force", 1)
    h.comm_init(crl.comm_unique_id(), 1, 0)
    st = O.State(cfg); st.params[:] = params; st.env_init()
    h.env_reset()
    h.prof_enable(True)
    gs = h.iterate(1); os_ = st.iterate(10, gen_perm=True)
    assert h.prof_read()["allreduce"][1] == 16
    for a, b in zip(gs, os_):
        assert loss_close("loss", a["loss"], b["loss"], RTOL), (a["loss"], b["loss"])
    assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < 1e-5
    agent.close(); st.close()


def test_wide_host_calls_chunk_over_the_workspace(crl):
    """crl_policy_act / crl_logprob_actions on more observations than the activation workspace holds (it is sized for one
    minibatch) are processed in chunks; empty inputs are no-ops."""
    D, A, Hd = 8, 4, 256
    cfg = ocfg(4, 8, D, A, Hd)                    # workspace: max(M, nt) = 8 samples
    params = spread_params(cfg, 2)
    agent = make_wide(crl, 4, 8, D, A, Hd, params=params)
    rng = np.random.default_rng(1)
    n = 1000
    obs = np.asfortranarray(rng.standard_normal((D, n)).astype(np.float32)); u = rng.random(n)
    a_o, lp_o, v_o, margin = O.get_action(cfg, params, obs, u)
    a_g, lp_g, v_g = agent.handle.policy_act(obs, u)
    safe = margin > 1e-6
    assert np.array_equal(a_g[safe], a_o[safe]) and rel_err(v_g, v_o) < RTOL
    lp2, ent = crl.logprob_actions(obs, agent.actor, (a_o + 1).astype(np.int32))
    assert rel_err_s(lp2, lp_o) < RTOL and ent.shape == (A, n)
    a0, lp0, v0 = agent.handle.policy_act(np.zeros((D, 0), np.float32, order="F"), np.zeros(0))
    assert a0.shape == (0,) and lp0.shape == (0,)
    agent.close()
