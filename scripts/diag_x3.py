import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oraclelib as O
os.environ["CRL_WIDE_GEMM"] = sys.argv[1]
import cleanrl_jl_amd as crl
from test_gpu_wide import make_wide, ocfg, spread_params
D, A, Hd, nt, k = 8, 4, 256, 70, 16
cfg = ocfg(nt, k, D, A, Hd)
params = spread_params(cfg, 5)
agent = make_wide(crl, nt, k, D, A, Hd, params=params)
st = O.State(cfg); st.params[:] = params; st.env_init()
h = agent.handle; F = crl._lib
h.env_reset(); h.rollout_run(); st.rollout()
for name, f, ref in (("logprob", F.F_LOGPROB, st.logprob), ("value", F.F_VALUE, st.value)):
    g = h.read(f).astype(np.float64); r = ref.astype(np.float64)
    err = np.abs(g - r); rel = err / (np.abs(r) + 0.1)
    i = np.argmax(rel)
    print(sys.argv[1], name, "max rel", rel.max(), "abs err there", err.flat[i], "ref there", r.flat[i], "mean abs err", err.mean(), "max abs", err.max(), "ref absmax", np.abs(r).max())
