#!/usr/bin/env python3
"""Is the fp16x2 product "narrower than Float32" in any way that shows? (verdict r5, weak 3 / next 2)

The reference multiplies Float32 matrices (`Flux.f32`, ppo.jl:87; Dense = W*x, networks.jl:6-13). This library's default multiplies the hidden-layer
products as fp16x2 split operands (hi + lo in f16, three f16 MFMAs, f32 accumulate; option gemm = 2), option gemm = 1 as bf16x3 (24-bit operands, six
MFMAs). This script takes the three 64x64 products of the headline's update pass ON THE HEADLINE'S OWN OPERANDS — a 65536-env CartPole rollout after a
few PPO iterations, real observations / actions / advantages, the handle's own parameters — and measures, for each of

    fp16x2 (production split + scales)   bf16x3 (production split)   v_mfma_f32_32x32x2_f32   a sequential v_fma_f32 chain per element

the error against a Float64 product of the SAME float32 operands (crl_product_probe runs the production split functions on the GPU):

    forward      z2  = W2 · h1          (K = 64)        weights x 2^8, activations x 2^14
    backward     dh1 = W2' · δ2         (K = 64)        weights x 2^8, δ2 scaled per SAMPLE by a power of two (mlp_x2.hpp: sample_scale)
    weight grad  dW2' = Σ_b h1_b δ2_b'  (K = samples)   activations x 2^14, δ2 x one power of two G per launch; 32-sample partial sums folded in f32

for both networks. Output: profiles/<tag>_product_error.json. What it shows is in DESIGN.md §3.0.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
import cleanrl_jl_amd as crl  # noqa: E402

L = crl._lib
TAG = sys.argv[1] if len(sys.argv) > 1 else bench.PROFILE_TAG
NT, K_STEPS, N = 65536, 128, 65536          # the headline's rollout; N samples of one of its minibatches
FLAVOURS = {0: "fp16x2", 1: "bf16x3", 2: "mfma_f32", 3: "f32_fma_chain"}


def tanh_fast_f32(x):
    """NNlib.tanh_fast(::Float32) as the oracle restates it (VERIFY_WITH_JULIA.md item 1), evaluated in float32."""
    x = x.astype(np.float32)
    x2 = x * x
    n = x * (np.float32(1) + x2 * (np.float32(0.1346604) + x2 * (np.float32(0.0035974074) + x2 * (np.float32(2.2332108e-5) + x2 * np.float32(1.587199e-8)))))
    d = np.float32(1) + x2 * (np.float32(0.4679937) + x2 * (np.float32(0.026262015) + x2 * (np.float32(0.0003453992) + x2 * np.float32(8.7767893e-7))))
    return np.where(x2 < 66, n / d, np.sign(x)).astype(np.float32)


def main():
    cfg = crl.PPOConfig(num_envs=NT, num_steps=K_STEPS, total_timesteps=NT * K_STEPS * 100)
    agent = crl.Agent(cfg, device=0)
    h = agent.handle
    h.env_reset()
    h.iterate(6, want_stats=False)          # a policy and a critic that have moved off their orthogonal start
    h.sync()
    p = h.read(L.F_PARAMS).astype(np.float32)
    from cleanrl_jl_amd import networks
    off = networks.param_offsets(2, 4, [64, 64])
    seg = [p[off[i]:off[i + 1]] for i in range(12)]
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(NT * K_STEPS, N, replace=False))
    obs = h.read(L.F_OBS).reshape(4, -1, order="F")[:, idx].astype(np.float32)       # (4, N)
    act = h.read(L.F_ACTION).reshape(-1, order="F")[idx]
    old_lp = h.read(L.F_LOGPROB).reshape(-1, order="F")[idx].astype(np.float64)
    adv = h.read(L.F_ADVANTAGE).reshape(-1, order="F")[idx].astype(np.float64)
    ret = h.read(L.F_RETURN).reshape(-1, order="F")[idx].astype(np.float64)
    old_v = h.read(L.F_VALUE).reshape(-1, order="F")[idx].astype(np.float64)
    agent.close()

    out = {"source_hash": bench.source_hash(), "samples": N,
           "operands": f"num_envs={NT}, num_steps={K_STEPS} CartPole rollout after 6 PPO iterations: {N} samples drawn from the batch, the handle's own parameters; "
                       "activations / cotangents computed in float32 (forward) and float64-then-rounded (cotangents) from them — every flavour multiplies the SAME float32 operands",
           "metric": "err = C_flavour - C_float64 per output element; rel_l2 = ||err|| / ||C_float64||, max_abs_over_rms = max|err| / rms(C_float64); "
                     "unit_roundoff_f32 = 2^-24 = 5.96e-8 for scale",
           "products": {}}
    M = float(N)
    for net, (w1, b1, w2, b2, w3, b3) in (("actor", seg[0:6]), ("critic", seg[6:12])):
        n_out = 2 if net == "actor" else 1
        W1 = w1.reshape(64, 4, order="F"); W2 = w2.reshape(64, 64, order="F"); W3 = w3.reshape(n_out, 64, order="F")
        h1 = tanh_fast_f32(W1 @ obs + b1[:, None])                                   # (64, N) float32
        z2 = (W2.astype(np.float64) @ h1.astype(np.float64)).astype(np.float32) + b2[:, None]
        h2 = tanh_fast_f32(z2)
        z3 = W3.astype(np.float64) @ h2.astype(np.float64) + b3[:, None].astype(np.float64)
        if net == "actor":                                                           # §8-GRAD of SURVEY.md (ppo.jl:203-243)
            zmax = z3.max(0); e = np.exp(z3 - zmax); pr = e / e.sum(0); lp = (z3 - zmax) - np.log(e.sum(0))
            nlp = lp[act, np.arange(N)]
            A_hat = (adv - adv.mean()) / (adv.std(ddof=1) + 1e-8)
            rho = np.exp(nlp - old_lp)
            unclipped_wins = (-A_hat * rho) > (-A_hat * np.clip(rho, 0.8, 1.2))
            dnlp = np.where(unclipped_wins | ((rho >= 0.8) & (rho <= 1.2)), -A_hat * rho, 0.0) / M
            onehot = np.zeros_like(pr); onehot[act, np.arange(N)] = 1.0
            Hb = -(pr * lp).sum(0)
            d3 = dnlp * (onehot - pr) - 0.01 * (-(1.0 / (2 * M)) * pr * (lp + Hb))
        else:
            v = z3[0]
            vc = old_v + np.clip(v - old_v, -0.2, 0.2)
            d3 = (0.5 * 0.5 / M * 2 * (vc - ret) * (np.abs(v - old_v) <= 0.2))[None, :]
        d2 = ((W3.astype(np.float64).T @ d3) * (1.0 - h2.astype(np.float64) ** 2)).astype(np.float32)   # (64, N) float32: δ2

        # per-sample power-of-two scale of the backward product (mlp_x2.hpp: sample_scale): largest |δ2| of the sample into [2^14, 2^15)
        m = np.abs(d2).max(0)
        ex = np.where(m > 0, np.floor(np.log2(np.maximum(m, 1e-300))), -127).astype(np.int64)
        col_scale = np.ldexp(1.0, (14 - ex).clip(-100, 100)).astype(np.float32)
        # one power of two per launch for the weight gradient (dw_next_scale: largest·G ≈ 2^8, exponent a multiple of 8)
        e_max = int(np.floor(np.log2(float(m.max()))))
        kq = 8 - e_max; kq = int(np.sign(kq) * ((abs(kq) + 4) // 8) * 8)
        G = float(np.ldexp(1.0, kq))

        cases = {
            "forward  W2*h1 (K=64)": dict(A=W2, B=h1.T, chunks=1, sa=256.0, sb=16384.0, cs=None),
            "backward W2'*d2 (K=64)": dict(A=W2.T, B=d2.T, chunks=1, sa=256.0, sb=1.0, cs=col_scale),
            f"weight gradient h1*d2' (K={N} samples, 32-sample partials folded in f32)": dict(A=h1, B=d2, chunks=N // 32, sa=16384.0, sb=G, cs=None),
        }
        for name, c in cases.items():
            A = np.ascontiguousarray(c["A"], np.float32); B = np.ascontiguousarray(c["B"], np.float32)
            ref = (B.astype(np.float64) @ A.astype(np.float64).T)                  # [cols, rows]
            rms = float(np.sqrt(np.mean(ref ** 2)))
            row = {"rows": int(A.shape[0]), "cols": int(B.shape[0]), "K": int(A.shape[1]), "rms_of_exact_result": rms}
            if "weight" in name:
                row["scale_G"] = G
            for fl, fname in FLAVOURS.items():
                if fl == 3 and c["chunks"] > 1:
                    # the scalar chain over ALL samples in one go is what a scalar f32 matmul does for K = samples
                    part = L.product_probe(fl, A, B, chunks=1)
                    got = part[0].astype(np.float64)
                else:
                    part = L.product_probe(fl, A, B, chunks=c["chunks"], scale_a=c["sa"], scale_b=c["sb"], col_scale=c["cs"])
                    acc = np.zeros(part.shape[1:], np.float32)
                    for ch in range(part.shape[0]):                                 # fixed-order f32 fold, like the production reduce
                        acc += part[ch]
                    got = acc.astype(np.float64)
                err = got - ref
                row[fname] = {"rel_l2": float(np.linalg.norm(err) / np.linalg.norm(ref)), "max_abs_over_rms": float(np.abs(err).max() / rms)}
            out["products"][f"{net}: {name}"] = row
            print(net, name, json.dumps({k: v for k, v in row.items() if isinstance(v, dict)}), flush=True)
    worst = {f: max(r[f]["rel_l2"] for r in out["products"].values()) for f in FLAVOURS.values()}
    out["worst_rel_l2"] = worst
    out["reading"] = ("fp16x2 against the float32 routes the reference could take: if worst_rel_l2.fp16x2 <= worst_rel_l2.f32_fma_chain the split operands lose nothing a Float32 "
                      "matmul keeps — the f32 ACCUMULATION of a K-term product costs more than the 2^-22 the operands give up")
    path = os.path.join(ROOT, "profiles", f"{TAG}_product_error.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, json.dumps(worst))


if __name__ == "__main__":
    main()
