"""The six-wave rollout kernel (csrc/policy.hip: rollout_split6_kernel, option rollout_split = 3) at the sizes it exists for — BASELINE configs[1]
(4096 envs) and one 8192-env shard of configs[3] — against the CPU oracle: the whole 128-step rollout (ppo.jl:123-166) under the knot-margin rule every
rollout kernel is held to, the fused compat-GAE tail bit-equal to orc_gae on the kernel's own buffers (ppo.jl:48-73), and whole crl_ppo_iterate
iterations on C1 against the oracle's."""
import numpy as np
import pytest

import oraclelib as O
from test_gpu_parity import IT_LOSS, IT_PARAM, RTOL, _knot_margin, _oracle_state, loss_close, make_agent, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1
    return crl


@pytest.mark.parametrize("nt", [4096, 8192, 100])
def test_six_wave_rollout_matches_oracle_at_shard_sizes(crl, nt):
    """nt = 100: a ragged last tile (four envs in it)."""
    k = 128
    agent = make_agent(crl, nt=nt, k=k, options={"rollout_split": 3})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle; F = crl._lib
    h.env_reset(); h.rollout_run(); st.rollout()
    act = h.read(F.F_ACTION)
    diff = act != st.action
    clean = ~diff.any(axis=1)
    for e in np.flatnonzero(~clean):
        t = int(np.argmax(diff[e]))
        m = _knot_margin(cfgo, params, st, e, t)
        assert m <= 1e-6, f"env {e} step {t}: action differs although the draw is {m:.3e} away from the CDF knot"
    assert clean.mean() > 0.999
    assert np.array_equal(h.read(F.F_OBS)[:, clean], st.obs[:, clean]) and np.array_equal(h.read(F.F_TERMINAL)[clean], st.terminal[clean])
    assert np.array_equal(h.read(F.F_REWARD)[clean], st.reward[clean])
    assert rel_err(h.read(F.F_LOGPROB)[clean], st.logprob[clean]) < RTOL and rel_err(h.read(F.F_VALUE)[clean], st.value[clean]) < RTOL
    assert np.array_equal(h.read(F.F_NEXT_DONE)[clean], st.next_done[clean]) and np.array_equal(h.read(F.F_ENV_STATE)[:, clean], st.env_state[:, clean])
    if clean.all():
        es = h.episode_stats(); n_ep, ret_sum, len_sum = st.episode_stats
        assert (es["episodes"], es["return_sum"], es["length_sum"]) == (n_ep, ret_sum, len_sum)
    agent.close(); st.close()


def test_six_wave_rollout_fused_gae_tail_is_bit_equal_to_the_oracle_scan(crl):
    """Inside crl_ppo_iterate the compat-mode GAE is the tail of the rollout kernel: advantages / returns bit-equal to orc_gae on the buffers the kernel
    itself wrote (8192 envs: the 8-GPU job's shard)."""
    nt, k = 8192, 128
    agent = make_agent(crl, nt=nt, k=k, options={"rollout_split": 3})
    h = agent.handle; F = crl._lib
    assert h.get_option("gae_fuse") == 1
    h.env_reset(); h.iterate(1, want_stats=False); h.sync()
    value, reward, term = h.read(F.F_VALUE), h.read(F.F_REWARD), h.read(F.F_TERMINAL)
    adv_o, ret_o = O.gae_batch(value, reward, term, np.zeros(nt, np.float32), np.zeros(nt, np.uint8), 0.99, 0.95, 0)
    assert np.array_equal(h.read(F.F_ADVANTAGE), adv_o) and np.array_equal(h.read(F.F_RETURN), ret_o)
    agent.close()


def test_full_iterations_with_the_six_wave_rollout_match_oracle(crl):
    nt, k = 8, 128
    agent = make_agent(crl, nt=nt, k=k, shuffle_mode=0, options={"rollout_split": 3})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle
    h.env_reset()
    for it in range(3):
        gs = h.iterate(1)
        os_ = st.iterate(10, gen_perm=True)
        assert np.array_equal(h.read(crl._lib.F_ACTION), st.action)
        assert rel_err(h.read(crl._lib.F_ADVANTAGE), st.adv) < RTOL
        for a, b in zip(gs, os_):
            for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                assert loss_close(key, a[key], b[key], IT_LOSS), (it, key, a[key], b[key])
        assert np.max(np.abs(h.read(crl._lib.F_PARAMS) - st.params)) < IT_PARAM
    agent.close(); st.close()
