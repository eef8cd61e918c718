import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import cleanrl_jl_amd as crl, oraclelib as O
L = crl._lib
# replay the fuzz stream up to a chosen case
target = int(sys.argv[1]); rng = np.random.default_rng(1)
for case in range(target + 1):
    k = int(rng.choice([8, 16, 32, 64, 128])); nt = int(rng.integers(1, 97)); B = nt * k
    nmb = int(rng.choice([d for d in (1, 2, 4, 8, 16) if B % d == 0])); epochs = int(rng.integers(1, 5))
    kw = dict(num_minibatches=nmb, update_epochs=epochs, clip_value_loss=bool(rng.integers(0, 2)), anneal_lr=bool(rng.integers(0, 2)),
              lr=float(rng.choice([2.5e-4, 1e-3])), clip_coef=float(rng.choice([0.1, 0.2])), ent_coeff=float(rng.choice([0.0, 0.01])))
    shape = dict(gae_mode=int(rng.integers(0, 2)), stale_obs=int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    blocked = bool(rng.integers(0, 2)); init_seed = int(rng.integers(0, 100))
print(dict(nt=nt, k=k, nmb=nmb, epochs=epochs, blocked=blocked, **kw, **shape))
cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=B * 7, **kw)
agent = crl.Agent(cfg, shuffle_mode=L.SHUFFLE_BLOCKED_FY if blocked else L.SHUFFLE_FISHER_YATES, init_seed=init_seed, **shape)
params = agent.get_params()
cfgo = O.make_config(num_envs=nt, num_steps=k, **kw, **shape)
st = O.State(cfgo); st.params[:] = params; st.env_init()
h = agent.handle; h.env_reset()
off = O.param_offsets(cfgo)
names = ["aW1","ab1","aW2","ab2","aW3","ab3","cW1","cb1","cW2","cb2","cW3","cb3"]
for it in range(2):
    if not blocked:
        h.iterate(1); st.iterate(7, gen_perm=True)
    else:
        eta = cfg.lr * (1.0 - it / 7.0) if cfg.anneal_lr else cfg.lr
        h.rollout_run(); h.compute_gae(); st.rollout(); st.compute_gae()
        for ep in range(epochs):
            eid = it * epochs + ep
            h.shuffle(eid); h.adv_stats(); st.perm[:] = O.shuffle_blocked_fy(B, cfgo.seed, eid)
            for mb in range(nmb):
                h.update_minibatch(mb, np.float32(eta)); st.update_minibatch(mb, np.float32(eta))
    pg, po = h.read(L.F_PARAMS), st.params
    d = np.abs(pg - po); i = int(np.argmax(d)); arr = int(np.searchsorted(off, i, side="right") - 1)
    vg, vo = h.read(L.F_ADAM_V), st.adam_v; mg, mo = h.read(L.F_ADAM_M), st.adam_m
    gg = h.read(L.F_GRADS); go = st.grads
    print(f"iter {it}: max |dp| {d.max():.3e} at {i} ({names[arr]}), adam_v gpu/orc {vg[i]:.3e}/{vo[i]:.3e}, m {mg[i]:.3e}/{mo[i]:.3e}, last grad {gg[i]:.3e}/{go[i]:.3e}; "
          f"entries with |dp|>1e-6: {(d>1e-6).sum()}, their median sqrt(v): {np.median(np.sqrt(vo[d>1e-6])) if (d>1e-6).any() else 0:.3e}; median sqrt(v) overall {np.median(np.sqrt(vo)):.3e}")
