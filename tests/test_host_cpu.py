"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/cleanrl_hip.h declares,
mirrors the reference's configuration surface, and FAILS LOUDLY without a GPU (no compute fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def crl():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "cleanrl.jl_amd", "libcleanrl_hip.so")):
        g.build()
    import cleanrl_jl_amd as crl
    return crl


def test_library_exports_every_declared_symbol(crl):
    hdr = open(os.path.join(ROOT, "include", "cleanrl_hip.h")).read()
    declared = set(re.findall(r"\b(crl_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"crl_ppo_config", "crl_ppo_stats", "crl_episode_stats"}
    lib = crl._lib.load()
    assert declared == set(crl._lib.EXPORTS), declared ^ set(crl._lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.crl_version() == 100


def test_config_struct_matches_header_and_reference_defaults(crl):
    # ppo.jl:1-19 defaults
    c = crl.PPOConfig()
    assert (c.total_timesteps, c.num_steps, c.num_envs, c.num_minibatches, c.update_epochs) == (500_000, 32, 4, 4, 4)
    assert (c.lr, c.gamma, c.gae_lambda, c.clip_coef, c.ent_coeff, c.v_coef) == (2.5e-4, 0.99, 0.95, 0.2, 0.01, 0.5)
    assert c.normalize_advantages and c.clip_value_loss and c.anneal_lr
    # field order of crl_ppo_config in the header == ctypes mirror
    hdr = open(os.path.join(ROOT, "include", "cleanrl_hip.h")).read()
    body = hdr[hdr.index("typedef struct crl_ppo_config {"):hdr.index("} crl_ppo_config;")]
    names = re.findall(r"^\s*(?:int64_t|int32_t|float|uint64_t)\s+([a-z_]+);", body, re.M)
    assert names == [n for n, _ in crl._lib.CrlConfig._fields_]
    assert C.sizeof(crl._lib.CrlConfig) == 104


def test_no_gpu_means_loud_failure_not_cpu_fallback(crl):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    assert crl.device_count() == 0
    with pytest.raises(crl.CrlError):
        crl.Agent(crl.PPOConfig())
    with pytest.raises(crl.CrlError):
        crl.gae([1, 2, 3], [1, 1], [0, 0, 0], 0.99, 0.95)


def test_networks_layout_matches_flux_param_order(crl):
    p = crl.make_actor_critic(2, 4, (64, 64), seed=0)
    off = crl.networks.param_offsets(2, 4) if hasattr(crl, "networks") else None
    from importlib import import_module
    nets = import_module("cleanrl_jl_amd.networks")
    off = nets.param_offsets(2, 4)
    assert off[-1] == 9155 == p.size  # 9,155 params for 2x64, obs 4, act 2
    W1 = p[off[0]:off[1]].reshape(64, 4, order="F")
    assert np.allclose(W1.T @ W1, 2 * np.eye(4), atol=1e-5)          # orthogonal columns, gain sqrt(2)
    W3a = p[off[4]:off[5]].reshape(2, 64, order="F")
    assert np.allclose(W3a @ W3a.T, 1e-4 * np.eye(2), atol=1e-8)      # gain 0.01
    W3c = p[off[10]:off[11]].reshape(1, 64, order="F")
    assert np.allclose(W3c @ W3c.T, 1.0, atol=1e-5)
    for b in (1, 3, 5, 7, 9, 11):
        assert not p[off[b]:off[b + 1]].any()                          # zero biases


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cleanrl.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("//", "#", "*", "/*")))
                assert "oraclelib" not in code and "libppo_oracle" not in code and "ppo_oracle.h" not in code, f


def test_argparse_struct_mirrors_config_parser():
    """config_parser.jl:18-40: one --field option per struct field, defaults from the struct, same struct type back."""
    import cleanrl_jl_amd as crl
    from cleanrl_jl_amd.config_parser import argparse_struct
    cfg = argparse_struct(crl.PPOConfig(), ["--num_envs", "64", "--lr", "0.001", "--anneal_lr", "false"])
    assert isinstance(cfg, crl.PPOConfig) and cfg.num_envs == 64 and cfg.lr == 0.001 and cfg.anneal_lr is False
    assert cfg.num_steps == crl.PPOConfig().num_steps and cfg.clip_value_loss is True
    d = argparse_struct(crl.DQNConfig(), ["--log_frequencey", "50", "--epsilon_duration", "300"])
    assert d.log_frequencey == 50 and d.epsilon_duration == 300.0 and isinstance(d, crl.DQNConfig)
    a = argparse_struct(crl.A2CConfig(run_name="x"), [])
    assert a == crl.A2CConfig(run_name="x")
    with pytest.raises(SystemExit):
        argparse_struct(crl.PPOConfig(), ["--no_such_field", "1"])
    with pytest.raises(TypeError):
        argparse_struct({"a": 1}, [])
