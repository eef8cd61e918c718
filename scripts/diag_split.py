import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np
import cleanrl_jl_amd as crl
import oraclelib as O
from test_gpu_parity import make_agent, _oracle_state
F = crl._lib
nt, k = 4096, 128
for split in (1, 3, 0):
    agent = make_agent(crl, nt=nt, k=k, options={"rollout_split": split})
    params = agent.get_params()
    cfgo, st = _oracle_state(nt, k, params)
    h = agent.handle
    h.env_reset(); h.rollout_run(); st.rollout()
    act = h.read(F.F_ACTION); clean = ~(act != st.action).any(axis=1)
    lp_g = h.read(F.F_LOGPROB)[clean].astype(np.float64); lp_o = st.logprob[clean].astype(np.float64)
    v_g = h.read(F.F_VALUE)[clean].astype(np.float64); v_o = st.value[clean].astype(np.float64)
    a = st.action[clean]
    d = lp_g - lp_o
    print(f"split {split}: logprob max|d| {np.abs(d).max():.3e} rms {np.sqrt((d**2).mean()):.3e} mean {d.mean():+.3e} mean|a=0 {d[a==0].mean():+.3e} mean|a=1 {d[a==1].mean():+.3e}; value max|d| {np.abs(v_g-v_o).max():.3e} rms {np.sqrt(((v_g-v_o)**2).mean()):.3e}")
    agent.close(); st.close()
# the failing test's configuration, iteration by iteration, both flavours: pg_loss deviation
for split in (1, 3):
    nt, k = 8, 128
    cfg = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10, gamma=0.0)
    agent = crl.Agent(cfg, shuffle_mode=F.SHUFFLE_FISHER_YATES, options={"rollout_split": split})
    cfgo = O.make_config(num_envs=nt, num_steps=k, gamma=0.0)
    params = agent.get_params(); params[O.param_offsets(cfgo)[11]] = 5.0; agent.set_params(params)
    h = agent.handle
    st = O.State(cfgo); st.params[:] = params; st.env_init()
    h.env_reset()
    for it in range(3):
        gs = h.iterate(1); os_ = st.iterate(10, gen_perm=True)
        dev = [abs(a["pg_loss"] - b["pg_loss"]) for a, b in zip(gs, os_)]
        lpd = np.abs(h.read(F.F_LOGPROB).astype(np.float64) - st.logprob).max()
        print(f"split {split} it {it}: max |pg_loss dev| {max(dev):.3e} (floor 5e-7), max |logprob dev| {lpd:.3e}, params dev {np.abs(h.read(F.F_PARAMS)-st.params).max():.3e}")
    agent.close(); st.close()
