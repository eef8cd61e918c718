"""Data-parallel arithmetic on ONE GPU: two shard handles (world_size 2, exchange done by the host) against one handle
that owns all envs. Checks what the 8-GPU run relies on and the 1-GPU box cannot run through RCCL:
  * a shard's rollout / GAE equals its slice of the single-handle run (global env ids key the RNG);
  * with GLOBAL advantage statistics and the global 1/M, the SUM of the shard gradient messages equals the gradient of
    the union minibatch (what ncclAllReduce(sum) produces), and the summed loss terms give the same statistics."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    return crl


@pytest.mark.parametrize("path", ["fused", "wide"])
def test_two_shards_equal_one_handle(crl, path):
    """fused = the 4/2/64 kernels of update.hip; wide = the layer-wise path on the C3 shape (obs 8 / act 4 / 2x256) with
    the plain value loss (the clipped one needs two extra scalar all-reduces that only the in-library RCCL path does)."""
    L = crl._lib
    NT, k, W = 16, 128, 2
    n = NT // W
    extra = {} if path == "fused" else dict(clip_value_loss=False)
    shape = {} if path == "fused" else dict(obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
    cfg_full = crl.PPOConfig(num_envs=NT, num_steps=k, total_timesteps=NT * k * 10, **extra)
    cfg_sh = crl.PPOConfig(num_envs=n, num_steps=k, total_timesteps=NT * k * 10, **extra)
    full = crl.Agent(cfg_full, **shape)
    params = full.get_params()
    shards = [crl.Agent(cfg_sh, params=params, env_id_offset=r * n, **shape) for r in range(W)]
    for r, a in enumerate(shards):
        a.handle.comm_init_external(W, r)
    for a in [full] + shards:
        a.handle.env_reset(); a.handle.rollout_run(); a.handle.compute_gae()
    hf = full.handle
    for f in (L.F_ACTION, L.F_REWARD, L.F_TERMINAL, L.F_VALUE, L.F_LOGPROB, L.F_ADVANTAGE, L.F_RETURN):
        whole = hf.read(f)
        for r, a in enumerate(shards):
            assert np.array_equal(a.handle.read(f), whole[r * n:(r + 1) * n]), f"field {f}, shard {r}"
    assert np.array_equal(np.concatenate([a.handle.read(L.F_OBS) for a in shards], axis=1), hf.read(L.F_OBS))

    # local permutations on the shards; the single handle gets the union, minibatch by minibatch
    rng = np.random.default_rng(0)
    Bl = n * k; Ml = Bl // 4
    perms = [rng.permutation(Bl).astype(np.int32) for _ in range(W)]
    union = []
    for mb in range(4):
        for r in range(W):
            loc = perms[r][mb * Ml:(mb + 1) * Ml]
            e, t = loc % n, loc // n
            union.append((r * n + e) + NT * t)
    hf.write(L.F_PERM, np.concatenate(union).astype(np.int32))
    for r, a in enumerate(shards):
        a.handle.write(L.F_PERM, perms[r])
    # advantage statistics: local sums, host "all-reduce", finish on every shard
    for a in shards:
        a.handle.adv_stats_local()
    tot = sum(a.handle.read(L.F_ADV_SUMS) for a in shards)
    for a in shards:
        a.handle.write(L.F_ADV_SUMS, tot); a.handle.adv_stats_finish()
    hf.adv_stats()
    for mb in (0, 2):
        sf = hf.update_minibatch(mb, 0.0, apply_update=False)
        gf = hf.read(L.F_GRADS).astype(np.float64)
        gs = np.zeros_like(gf)
        for a in shards:
            a.handle.update_minibatch(mb, 0.0, apply_update=False, want_stats=False)
            gs += a.handle.read(L.F_GRADS)
        err = np.linalg.norm(gs - gf) / np.linalg.norm(gf)
        assert err < 1e-5, err
        assert abs(sf["adv_std"]) > 0
    for a in [full] + shards:
        a.close()


def test_single_minibatch_mode_one_message_per_epoch(crl):
    """num_minibatches = 1 (north_star's "single all-reduce of gradients per update epoch"): the whole local batch is one
    minibatch, so an epoch is ONE optimiser step and one gradient message. Two shards with host-side exchange == one handle."""
    L = crl._lib
    NT, k, W = 16, 128, 2
    n = NT // W
    full = crl.Agent(crl.PPOConfig(num_envs=NT, num_steps=k, num_minibatches=1, total_timesteps=NT * k * 10))
    params = full.get_params()
    shards = [crl.Agent(crl.PPOConfig(num_envs=n, num_steps=k, num_minibatches=1, total_timesteps=NT * k * 10), params=params,
                        env_id_offset=r * n) for r in range(W)]
    for r, a in enumerate(shards):
        a.handle.comm_init_external(W, r)
    for a in [full] + shards:
        a.handle.env_reset(); a.handle.rollout_run(); a.handle.compute_gae()
    rng = np.random.default_rng(1)
    perms = [rng.permutation(n * k).astype(np.int32) for _ in range(W)]
    union = np.concatenate([(r * n + perms[r] % n) + NT * (perms[r] // n) for r in range(W)]).astype(np.int32)
    full.handle.write(L.F_PERM, union)
    for r, a in enumerate(shards):
        a.handle.write(L.F_PERM, perms[r]); a.handle.adv_stats_local()
    tot = sum(a.handle.read(L.F_ADV_SUMS) for a in shards)
    assert tot.shape == (2,)
    for a in shards:
        a.handle.write(L.F_ADV_SUMS, tot); a.handle.adv_stats_finish()
    full.handle.adv_stats()
    sf = full.handle.update_minibatch(0, 0.0, apply_update=False)
    gf = full.handle.read(L.F_GRADS).astype(np.float64)
    gs = np.zeros_like(gf)
    for a in shards:
        a.handle.update_minibatch(0, 0.0, apply_update=False, want_stats=False)
        gs += a.handle.read(L.F_GRADS)
    assert np.linalg.norm(gs - gf) / np.linalg.norm(gf) < 1e-5 and np.isfinite(sf["loss"])
    for a in [full] + shards:
        a.close()
