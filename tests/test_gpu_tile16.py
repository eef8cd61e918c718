"""The 16-sample-tile update kernel (csrc/update16.hpp, option update_tile = 16: three waves per SIMD, one early-exit repair launch) against the CPU
oracle, wherever the 32-sample kernel is tested: small and ragged minibatches, the live u > q value-loss branch, whole iterations on C1, the C2-size
minibatch (the size the option exists for), a weight outside the fp16 window, a scale miss on every tile (update_tile = 17: the repair launch produces
the result), determinism. Same bars as tests/test_gpu_parity.py (ppo.jl:202-250)."""
import numpy as np
import pytest

import oraclelib as O
from test_gpu_parity import IT_LOSS, IT_PARAM, RTOL, _grad_close, _inject_batch, _oracle_state, loss_close, make_agent, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1
    return crl


@pytest.mark.parametrize("tile", [16, 17])
@pytest.mark.parametrize("nt,k,ret_scale,clipv", [(8, 128, 10.0, True), (8, 128, 0.05, True), (8, 128, 3.0, False), (64, 128, 10.0, True), (37, 64, 10.0, True),
                                                   (3, 20, 10.0, True)])
def test_update_gradient_matches_oracle_on_16_sample_tiles(crl, nt, k, ret_scale, clipv, tile):
    """(3, 20): M = 15 — a single, ragged tile; ret_scale = 0.05 drives u > 0: the exact value-loss pass follows the 16-sample launch like the 32-sample one."""
    rng = np.random.default_rng(nt + k)
    cfgo = O.make_config(num_envs=nt, num_steps=k, clip_value_loss=clipv)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    off = O.param_offsets(cfgo)
    if ret_scale < 1:
        params[off[11]] = 0.3
    agent = make_agent(crl, nt=nt, k=k, params=params, clip_value_loss=clipv, options={"update_tile": tile})
    st = O.State(cfgo); st.params[:] = params
    _inject_batch(crl, agent, st, rng, ret_scale)
    h = agent.handle
    h.adv_stats()
    M = nt * k // 4
    for mb in (0, 3):
        gs = h.update_minibatch(mb, 2.5e-4, apply_update=False)
        g_orc, so = O.loss_grad(cfgo, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[mb * M:(mb + 1) * M])
        if ret_scale < 1 and clipv:
            assert so["n_unclipped_wins"] > 0 and gs["n_unclipped_wins"] == so["n_unclipped_wins"]
        for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
            assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
        _grad_close(h.read(crl._lib.F_GRADS), g_orc, off)
    agent.close(); st.close()


@pytest.mark.parametrize("fuse_optim", [1, 0])
def test_full_iterations_on_16_sample_tiles_match_oracle(crl, fuse_optim):
    nt, k = 8, 128
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=0, options={"fuse_optim": fuse_optim, "update_tile": 16})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle
    h.env_reset()
    for it in range(3):
        gs = h.iterate(1)
        os_ = st.iterate(10, gen_perm=True)
        assert np.array_equal(h.read(crl._lib.F_ACTION), st.action)
        for a, b in zip(gs, os_):
            for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                assert loss_close(key, a[key], b[key], IT_LOSS), (it, key, a[key], b[key])
        assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < IT_PARAM
    agent.close(); st.close()


def test_c2_size_minibatch_on_16_sample_tiles_matches_oracle_and_is_deterministic(crl):
    """BASELINE configs[1] (num_envs=4096, M = 131,072): the size the 16-sample kernel is for. Gradient and losses against orc_loss_grad on the GPU's own
    buffer; two launches with a settled scale give the same bits; the 32-sample kernel on the same minibatch agrees to float32 summation order."""
    nt, k = 4096, 128
    agent = make_agent(crl, nt=nt, k=k, options={"update_tile": 16})
    params = agent.get_params()
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); h.compute_gae(); h.shuffle(3); h.adv_stats()
    h.update_minibatch(2, 0.0, apply_update=False, want_stats=False)        # settles the carried weight-gradient scale
    gs = h.update_minibatch(2, 0.0, apply_update=False); g = h.read(F.F_GRADS)
    gs2 = h.update_minibatch(2, 0.0, apply_update=False); g2 = h.read(F.F_GRADS)
    assert np.array_equal(g, g2) and gs["loss"] == gs2["loss"]
    M = nt * k // 4
    perm = h.read(F.F_PERM)
    g_o, so = O.loss_grad(cfgo, params, h.read(F.F_OBS).reshape(4, -1, order="F"), h.read(F.F_ACTION), h.read(F.F_LOGPROB), h.read(F.F_VALUE),
                          h.read(F.F_ADVANTAGE), h.read(F.F_RETURN), perm[2 * M:3 * M])
    for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
        assert loss_close(key, gs[key], so[key], RTOL), (key, gs[key], so[key])
    _grad_close(g, g_o, O.param_offsets(cfgo))
    h.set_option("update_tile", 32)
    gs32 = h.update_minibatch(2, 0.0, apply_update=False); g32 = h.read(F.F_GRADS)
    assert np.linalg.norm(g32.astype(np.float64) - g) <= 2e-6 * np.linalg.norm(g.astype(np.float64)) and loss_close("loss", gs32["loss"], gs["loss"], 2e-6)
    agent.close()


def test_weight_outside_the_fp16_window_is_repaired_on_16_sample_tiles(crl):
    nt, k = 8, 128
    rng = np.random.default_rng(3)
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    off = O.param_offsets(cfgo)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    agent = make_agent(crl, nt=nt, k=k, params=params, options={"update_tile": 16})
    h = agent.handle
    assert h.get_option("gemm_fallback_seen") == 0
    st = O.State(cfgo)
    for which, base in (("critic", off[8]), ("actor", off[2])):
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
    # and the launch after the weights are back in range runs clean again (the miss flag is lowered by the reduce)
    agent.set_params(params); st.params[:] = params
    _inject_batch(crl, agent, st, rng)
    h.adv_stats()
    gs = h.update_minibatch(0, 0.0, apply_update=False)
    M = nt * k // 4
    g_o, so = O.loss_grad(cfgo, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[:M])
    _grad_close(h.read(crl._lib.F_GRADS), g_o, off)
    with pytest.raises(crl.CrlError, match="update_tile"):
        h.set_option("update_tile", 8)
    agent.close(); st.close()
