"""One randomised whole-iteration parity case: crl_ppo_iterate (serial Fisher–Yates, the shuffle the oracle's orc_iterate draws
itself) or the call-by-call path with the blocked shuffle, against the CPU oracle, on a random small configuration — shapes the
fixed tests do not enumerate (env counts that are not multiples of 32, other num_steps / minibatch / epoch counts, both GAE modes,
stale and fresh observations, clipped and unclipped value loss, annealing on and off). Shared by tests/test_gpu_fuzz.py (40 cases,
seed 1, in the GPU suite) and scripts/fuzz_parity.py (any number of cases / seeds from a shell).

Bars: permutations and actions bit-equal; every loss record within LOSS relative (floored as in tests/test_gpu_parity.py);
parameters within 1e-5 relative L2 and lr/2 per entry. (Not "1e-6 per entry" as in the fixed tests: Adam's first steps divide by
|g| + 1e-8, so an entry whose gradient is ~1e-8 — five orders below the array's typical entry — turns a 1e-9 absolute difference
in that gradient, i.e. float32 summation order, into a visible fraction of one learning-rate step; over 240 random configurations
49 had such entries, at most 65 of 9,155, the largest 0.13·lr, while every loss stayed within 6.2e-7.)"""
import numpy as np

LOSS, PARAM = 2e-6, 1e-6
LOSS_FLOOR = 2e-7     # pg_loss / loss are means of O(1) terms that cancel to ~0: float32 rounding of the terms, not of the result


def run_case(crl, O, rng, case=0):
    L = crl._lib
    k = int(rng.choice([8, 16, 32, 64, 128]))
    nt = int(rng.integers(1, 97))
    B = nt * k
    nmb = int(rng.choice([d for d in (1, 2, 4, 8, 16) if B % d == 0]))
    epochs = int(rng.integers(1, 5))
    kw = dict(num_minibatches=nmb, update_epochs=epochs, clip_value_loss=bool(rng.integers(0, 2)), anneal_lr=bool(rng.integers(0, 2)),
              lr=float(rng.choice([2.5e-4, 1e-3])), clip_coef=float(rng.choice([0.1, 0.2])), ent_coeff=float(rng.choice([0.0, 0.01])))
    shape = dict(gae_mode=int(rng.integers(0, 2)), stale_obs=int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    blocked = bool(rng.integers(0, 2))
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=B * 7, **kw)
    agent = crl.Agent(cfg, shuffle_mode=L.SHUFFLE_BLOCKED_FY if blocked else L.SHUFFLE_FISHER_YATES, init_seed=int(rng.integers(0, 100)), **shape)
    params = agent.get_params()
    cfgo = O.make_config(num_envs=nt, num_steps=k, **kw, **shape)
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h = agent.handle
    h.env_reset()
    worst = {"loss": 0.0, "param": 0.0, "n_param": 0, "l2": 0.0}
    ok = True
    why = ""
    try:
        for it in range(2):
            if not blocked:
                gs = h.iterate(1)
                os_ = st.iterate(7, gen_perm=True)
                if not np.array_equal(h.read(L.F_PERM), st.perm):
                    ok = False; why = "perm"
            else:
                # the oracle's own loop draws serial shuffles: drive both sides epoch by epoch with the blocked permutations
                eta = cfg.lr * (1.0 - it / 7.0) if cfg.anneal_lr else cfg.lr
                h.rollout_run(); h.compute_gae(); st.rollout(); st.compute_gae()
                gs, os_ = [], []
                for ep in range(epochs):
                    eid = it * epochs + ep
                    h.shuffle(eid); h.adv_stats()
                    st.perm[:] = O.shuffle_blocked_fy(B, cfgo.seed, eid)
                    if not np.array_equal(h.read(L.F_PERM), st.perm):
                        ok = False; why = "perm"
                    for mb in range(nmb):
                        gs.append(h.update_minibatch(mb, np.float32(eta))); os_.append(st.update_minibatch(mb, np.float32(eta)))
            if not np.array_equal(h.read(L.F_ACTION), st.action):
                ok = False; why = why or "actions"
            for a, b in zip(gs, os_):
                for key in ("loss", "pg_loss", "v_loss", "entropy_loss"):
                    floor = LOSS_FLOOR if key in ("loss", "pg_loss") else 0.0
                    e = max(0.0, abs(a[key] - b[key]) - floor) / max(abs(b[key]), 1e-30)
                    worst["loss"] = max(worst["loss"], e)
            dp = np.abs(h.read(L.F_PARAMS) - st.params)
            worst["param"] = max(worst["param"], float(dp.max())); worst["n_param"] = max(worst["n_param"], int((dp > PARAM).sum()))
            worst["l2"] = max(worst["l2"], float(np.linalg.norm(dp.astype(np.float64)) / np.linalg.norm(st.params.astype(np.float64))))
        if worst["loss"] > LOSS or worst["l2"] > 1e-5 or worst["param"] > 0.5 * cfg.lr:
            ok = False; why = why or "tolerance"
    except Exception as e:   # noqa: BLE001 — a library error is a finding too
        ok = False; why = f"exception: {e}"
    line = dict(case=case, nt=nt, k=k, nmb=nmb, epochs=epochs, blocked=blocked, **{a: kw[a] for a in ("clip_value_loss", "anneal_lr")}, **shape,
                lr=kw["lr"], loss_rel=worst["loss"], param_abs=worst["param"], params_over_1e6=worst["n_param"], param_rel_l2=worst["l2"], ok=ok, why=why)
    agent.close(); st.close()
    return line
