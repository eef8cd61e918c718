"""Generates tests/golden/widen_fixture.npz with the CPU oracles (oracle/ppo_oracle.c on the C3 shape, a2c_oracle.c,
dqn_oracle.c): expected outputs of fixed-seed runs of the rows built after the PPO/CartPole path — BASELINE configs[2]
(obs 8 / act 4 / 2x256), A2C and DQN. Initial weights come from fixed numpy seeds (oraclelib.orthogonal_params / dqn_params);
large parameter vectors are stored as 64 probed entries + their L2 norm.

    python tests/golden/make_golden_widen.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oraclelib as O  # noqa: E402

PROBE = np.random.default_rng(99)


def probe(v, name, out):
    v = np.asarray(v, np.float64)
    ix = np.random.default_rng(hash(name) % 2**32).integers(0, v.size, 64)
    out[name + "_ix"] = ix; out[name + "_val"] = v[ix].astype(np.float32); out[name + "_norm"] = np.float64(np.linalg.norm(v))


def c3_inputs():
    cfg = O.make_config(num_envs=16, num_steps=32, obs_dim=8, n_act=4, hidden=256, env_kind=1)
    rng = np.random.default_rng(7)
    params = O.orthogonal_params(cfg, 7) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    off = O.param_offsets(cfg)
    params[off[4]:off[5]] *= 3
    return cfg, params


def main():
    out = {}
    # ---- C3 shape: two whole iterations on the synthetic env, exact Fisher–Yates ----
    cfg, params = c3_inputs()
    st = O.State(cfg); st.params[:] = params; st.env_init()
    for it in range(2):
        stats = st.iterate(10, gen_perm=True)
    out["c3_action"] = st.action.copy(); out["c3_perm"] = st.perm.copy(); out["c3_reward"] = st.reward.copy()
    out["c3_terminal"] = st.terminal.copy(); out["c3_adv"] = st.adv.copy()
    out["c3_stats"] = np.array([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats])
    probe(st.params, "c3_params", out)
    st.close()
    # ---- A2C: 4000 steps from seed 17 ----
    a = O.A2CState(O.a2c_config(total_timesteps=4000, lr=1e-3, seed=17), O.orthogonal_params(O.make_config(), 4))
    eps, losses = [], []
    while True:
        taken, ts, e = a.run_until_update()
        eps += e
        if ts["trained"]:
            losses.append((ts["n"], ts["critic_loss"], ts["actor_loss"]))
        if taken == 0 or a.env()[1] >= 4000:
            break
    out["a2c_episodes"] = np.array(eps, np.float64); out["a2c_losses"] = np.array(losses, np.float64)
    out["a2c_env"] = a.env()[0]; out["a2c_params"] = a.get_params()
    a.close()
    # ---- DQN: 1500 steps from seed 23 ----
    d = O.DQNState(O.dqn_config(total_timesteps=1500, lr=1e-3, log_frequency=100, seed=23), O.dqn_params(1))
    taken, eps, losses = d.run(10_000)
    out["dqn_episodes"] = np.array(eps, np.float64); out["dqn_losses"] = np.array(losses, np.float64)
    out["dqn_env"] = d.env()["state"]; out["dqn_q"] = d.params()[0]; out["dqn_target"] = d.params()[1]
    d.close()
    np.savez_compressed(os.path.join(HERE, "widen_fixture.npz"), **out)
    print("wrote widen_fixture.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
