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


_C_CALLER = r'''
/* A plain C99 translation unit: includes the public header, checks the layouts the ctypes / ccall mirrors assume and calls
 * the two entry points that need no GPU. */
#include <stdio.h>
#include <string.h>
#include "cleanrl_hip.h"
int main(void) {
  if (crl_version() != CRL_VERSION) { printf("version %d\n", (int)crl_version()); return 2; }
  const char* e = crl_last_error();
  if (!e) return 3;
  crl_ppo* h = (crl_ppo*)0;
  crl_ppo_config cfg; memset(&cfg, 0, sizeof cfg);
  if (crl_ppo_create(&cfg, 0, &h) == 0) return 4;         /* an all-zero config must be rejected, with a message */
  if (strlen(crl_last_error()) == 0) return 5;
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(crl_ppo_config), sizeof(crl_ppo_stats), sizeof(crl_episode_stats),
         sizeof(crl_episode_record), sizeof(crl_a2c_config), sizeof(crl_dqn_config), sizeof(crl_dqn_status), sizeof(crl_ppo_iteration_report));
  return 0;
}
'''


def test_header_is_c99_and_a_c_program_links_the_library(crl, tmp_path):
    """include/cleanrl_hip.h must be consumable by a C compiler (that is what a Julia `ccall` / cgo / ctypes binding
    assumes), and a C program linked against libcleanrl_hip.so must be able to call it."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    hdr = os.path.join(inc, "cleanrl_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "caller.c"
    src.write_text(_C_CALLER)
    exe = tmp_path / "caller"
    libdir = os.path.join(ROOT, "cleanrl.jl_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe),
                           os.path.join(libdir, "libcleanrl_hip.so"), f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    sizes = [int(x) for x in out.stdout.split()]
    L = crl._lib
    assert sizes == [C.sizeof(L.CrlConfig), C.sizeof(L.CrlStats), C.sizeof(L.CrlEpisodeStats), C.sizeof(L.CrlEpisodeRecord),
                     C.sizeof(L.CrlA2CConfig), C.sizeof(L.CrlDQNConfig), C.sizeof(L.CrlDQNStatus), C.sizeof(L.CrlIterationReport)], sizes
    assert sizes[:1] == [104] and sizes[4] == 40 and sizes[5] == 112 and sizes[7] == 56


def test_a2c_and_dqn_config_mirrors_follow_the_header(crl):
    """Field order of crl_a2c_config / crl_dqn_config in the header == the ctypes mirrors (sizes are pinned by the C program
    of the previous test)."""
    hdr = open(os.path.join(ROOT, "include", "cleanrl_hip.h")).read()
    for cname, mirror in (("crl_a2c_config", crl._lib.CrlA2CConfig), ("crl_dqn_config", crl._lib.CrlDQNConfig)):
        body = hdr[hdr.index("typedef struct %s {" % cname):hdr.index("} %s;" % cname)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in re.findall(r"(?:int64_t|int32_t|double|uint64_t)\s+([a-z_, ]+);", body):
            names += [n.strip() for n in decl.split(",")]
        assert names == [n for n, _ in mirror._fields_], (cname, names)


def test_julia_shell_structs_and_symbols_follow_the_header(crl):
    """julia/CleanRLHip.jl cannot be executed here (no Julia in the image): at least its isbits mirrors must list the header's
    fields in the header's order, and every symbol it `ccall`s must be one the library exports."""
    jl = open(os.path.join(ROOT, "julia", "CleanRLHip.jl")).read()
    L = crl._lib
    for jname, mirror in (("CrlConfig", L.CrlConfig), ("CrlStats", L.CrlStats), ("CrlEpisodeStats", L.CrlEpisodeStats),
                          ("CrlEpisodeRecord", L.CrlEpisodeRecord), ("CrlA2CConfig", L.CrlA2CConfig), ("CrlDQNConfig", L.CrlDQNConfig),
                          ("CrlIterationReport", L.CrlIterationReport)):
        m = re.search(r"struct %s\b(.*?)\bend" % jname, jl, re.S)
        assert m, jname
        body = re.sub(r"#.*", "", m.group(1))
        names = re.findall(r"([A-Za-z_][A-Za-z_0-9]*)::", body)
        assert names == [n for n, _ in mirror._fields_], (jname, names)
    called = set(re.findall(r"ccall\(\(:(crl_[a-z_0-9]+),", jl))
    assert called and called <= set(L.EXPORTS), called - set(L.EXPORTS)
    for must in ("crl_comm_init", "crl_comm_unique_id", "crl_episode_ring_enable", "crl_ppo_iterate_async", "crl_ppo_drain", "crl_dqn_run", "crl_dqn_q_values",
                 "crl_ppo_set_option", "crl_ppo_get_option", "crl_comm_destroy"):
        assert must in called, must
    assert "shuffle_mode::Integer=2" in jl     # exact blocked Fisher-Yates by default, like the ctypes mirror
    # shapes are arguments, not literals (ppo.jl:85-87, networks.jl:36-38): the constructor forwards them into the C struct in header order,
    # the entropy matrix is (n_act, batch), and ppo() passes shape keywords through — BASELINE config 3 (obs 8 / act 4 / 2x256) is reachable
    ctor = re.search(r"function Agent\(config::PPOConfig;(.*?)\)\n(.*?)\n  end", jl, re.S)
    assert ctor, "Agent constructor not found"
    kw = ctor.group(1)
    for name, dflt in (("obs_dim", "4"), ("n_act", "2"), ("hidden", "64"), ("gae_mode", "0"), ("stale_obs", "true"), ("device", "0"), ("env_id_offset", "0")):
        assert re.search(r"\b%s(::[A-Za-z]+)?=%s\b" % (name, dflt), kw), (name, kw)
    assert "env_kind" in kw
    call = re.search(r"c = CrlConfig\((.*?)\)\n", ctor.group(2), re.S).group(1)
    args = [a.strip() for a in call.replace("\n", " ").split(",")]
    want = [n for n, _ in L.CrlConfig._fields_]
    assert len(args) == len(want), (len(args), len(want))
    for a, w in zip(args, want):
        assert a in (w, "config." + w), (a, w)      # every field of crl_ppo_config is fed by the like-named keyword / PPOConfig field
    assert not re.search(r"CrlConfig\([^)]*\b4, 2, 64\b", jl), "shape literals are back in the constructor"
    assert "Matrix{Float32}(undef, actor.n_act, n)" in jl
    assert re.search(r"function ppo\(config::PPOConfig=PPOConfig\(\);.*?shape\.\.\.\)", jl, re.S) and "Agent(config; device, env_id_offset=rank * config.num_envs, shape...)" in jl
    assert "crl_ppo_param_count" in called
    # ppo.jl:77: the logger is installed first; the runner (README.md:24, config_parser.jl:18-40) goes argparse_struct -> ppo
    body = re.search(r"function ppo\(config::PPOConfig=PPOConfig\(\);(.*?)\n  agent\nend", jl, re.S).group(1)
    assert 'run_name::AbstractString="ppo-2-test"' in body and "make_logger=_default_make_logger()" in body
    assert body.index("make_logger(run_name; to_terminal=false)") < body.index("agent = Agent(")
    run = open(os.path.join(ROOT, "julia", "run_ppo.jl")).read()
    assert "ConfigParser.argparse_struct(CleanRLHip.PPOConfig())" in run and "CleanRLHip.ppo(config; make_logger = CleanRL.Logger.make_logger" in run
    assert 'include(joinpath(@__DIR__, "CleanRLHip.jl"))' in run


def _jl_function(jl, head):
    """Source of the Julia function whose definition starts with `head` (up to the first column-0 `end`)."""
    i = jl.index(head)
    return jl[i:jl.index("\nend\n", i)]


def test_julia_shell_cannot_train_a_dead_network(crl):
    """ppo.jl:87 / a2c.jl:37 / dqn.jl:39 on the Julia side of the boundary (verdict r5, row b): every default-keyword path of the three
    entry points builds the reference's networks — the reference's own builder when the shell sits inside the package, the library's
    restatement otherwise — and NO path leaves the zeros of a fresh handle in place. Desk-checked list: VERIFY_WITH_JULIA.md §B."""
    jl = open(os.path.join(ROOT, "julia", "CleanRLHip.jl")).read()
    # the reference builder, exactly as ppo.jl:87 calls it, flattened in Flux.params order (ppo.jl:196)
    ref = _jl_function(jl, "function reference_params(")
    assert "Networks.make_actor_critic(Base.OneTo(n_act), Base.OneTo(obs_dim), Int[hidden, hidden]) .|> Flux.f32" in ref
    assert "vcat(vec.(Flux.params(actor, critic))...)" in ref
    dflt = _jl_function(jl, "function _default_init()")
    assert "isdefined(pm, :Networks)" in dflt and "reference_params(" in dflt
    ppo = _jl_function(jl, "function ppo(config::PPOConfig=PPOConfig();")
    assert "params::Union{Nothing,Vector{Float32}}=nothing" in ppo and "init=_default_init()" in ppo
    # with params === nothing: the builder, else crl_ppo_init_params — an if / else with no third way out — and all of it before the first iterate
    assert "params = init(agent.n_act, agent.obs_dim, agent.hidden)" in ppo
    m = re.search(r"if params === nothing\n\s+init_params!\(agent, init_seed\).*?\n\s+else\n(.*?)\n\s+end", ppo, re.S)
    assert m and "set_params!(agent, params)" in m.group(1)
    assert ppo.index("init_params!(agent, init_seed)") < ppo.index(":crl_ppo_iterate")
    # the loop is the pipelined one: every iterate_async is followed by a drain, and the records of an update are emitted from ITS report (iteration index), not from the loop counter
    assert ppo.count(":crl_ppo_iterate_async") == 1 and ppo.count(":crl_ppo_drain") == 1 and ppo.index(":crl_ppo_iterate_async") < ppo.index(":crl_ppo_drain")
    assert "base = r.iteration * batch_size" in ppo and ppo.count("rep[].iteration >= 0 && emit(rep[])") == 2
    assert "crl_ppo_init_params" in _jl_function(jl + "\nend\n", "init_params!(a::Agent")
    # a2c / dqn: same rule (they used to demand `params`; now the reference's default call works and still cannot reach zeros)
    a2c = _jl_function(jl, "function a2c(config;")
    assert "init=_default_init()" in a2c and "params = init(2, 4, 64)" in a2c and ":crl_a2c_init_params" in a2c and ":crl_a2c_write_params" in a2c
    assert a2c.index(":crl_a2c_init_params") < a2c.index(":crl_a2c_run_until_update")
    assert 'make_logger("a2c|$(config.run_name)")' in a2c              # a2c.jl:30
    dqn = _jl_function(jl, "function dqn(config;")
    assert "init=_default_dqn_init()" in dqn and ":crl_dqn_init_params" in dqn and ":crl_dqn_write_params" in dqn
    assert dqn.index(":crl_dqn_init_params") < dqn.index(":crl_dqn_run")
    assert 'make_logger("dqn|$(config.run_name)")' in dqn              # dqn.jl:35
    assert "make_nn" in _jl_function(jl, "function _default_dqn_init()")
    # the runner hands the reference's own Networks / Flux over (its include is at top level, where the in-package default sees nothing)
    run = open(os.path.join(ROOT, "julia", "run_ppo.jl")).read()
    assert "CleanRLHip.reference_params(CleanRL.Networks, CleanRL.Flux, n_act, obs_dim, hidden)" in run
    # and the header says what a fresh handle is
    hdr = open(os.path.join(ROOT, "include", "cleanrl_hip.h")).read()
    assert "parameters not set" in hdr and "crl_ppo_init_params" in hdr


def test_library_initialisers_are_reference_shaped(crl):
    """crl_make_actor_critic (networks.jl:36-49: orthogonal weights with gains sqrt(2) / 0.01 / 1.0, zero biases, Flux.params order) and
    crl_dqn_make_nn (dqn.jl:22-26: glorot-uniform weights, zero biases) — host-only, so checked without a GPU."""
    L = crl._lib
    from cleanrl_jl_amd import networks
    for d, A, h in ((4, 2, 64), (8, 4, 256), (3, 5, 128)):
        w = L.make_actor_critic_host(d, A, h, seed=11)
        off = networks.param_offsets(A, d, [h, h])
        assert w.size == off[-1]
        shapes = {0: (h, d, 2 ** 0.5), 2: (h, h, 2 ** 0.5), 4: (A, h, 0.01), 6: (h, d, 2 ** 0.5), 8: (h, h, 2 ** 0.5), 10: (1, h, 1.0)}
        for i in range(12):
            seg = w[off[i]:off[i + 1]]
            if i % 2:
                assert not seg.any(), "bias %d is not zero" % i
                continue
            r, c, g = shapes[i]
            W = seg.reshape((r, c), order="F").astype(np.float64) / g
            G = W.T @ W if r >= c else W @ W.T
            assert np.abs(G - np.eye(min(r, c))).max() < 1e-6, (d, A, h, i)
        assert np.array_equal(w, L.make_actor_critic_host(d, A, h, seed=11))          # seeded: the same on every rank
        assert not np.array_equal(w, L.make_actor_critic_host(d, A, h, seed=12))
    with pytest.raises(L.CrlError, match="expected"):
        L.check(L.load().crl_make_actor_critic(4, 2, 64, 0, L._ptr(np.zeros(7, np.float32), C.c_float), 7))
    q = L.dqn_make_nn_host(seed=5)
    lay = [(120, 4), (84, 120), (2, 84)]
    o = 0
    for out_dim, in_dim in lay:
        W = q[o:o + out_dim * in_dim]; b = q[o + out_dim * in_dim:o + out_dim * in_dim + out_dim]
        lim = np.sqrt(6.0 / (in_dim + out_dim))
        assert not b.any() and np.abs(W).max() <= lim * (1 + 1e-6) and np.abs(W).max() > 0.9 * lim
        assert abs(W.std() - lim / np.sqrt(3)) < 0.1 * lim                          # uniform on (-lim, lim)
        o += out_dim * in_dim + out_dim
    assert o == L.DQN_PARAM_COUNT


def _pb_fields(buf):
    """Minimal protobuf reader: yields (field number, wire type, value) — enough for tensorflow.Event / Summary."""
    import struct
    i = 0
    while i < len(buf):
        key = 0; shift = 0
        while True:
            b = buf[i]; i += 1; key |= (b & 0x7F) << shift; shift += 7
            if not b & 0x80: break
        num, wt = key >> 3, key & 7
        if wt == 0:
            v = 0; shift = 0
            while True:
                b = buf[i]; i += 1; v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80: break
        elif wt == 1:
            v = struct.unpack_from("<d", buf, i)[0]; i += 8
        elif wt == 5:
            v = struct.unpack_from("<f", buf, i)[0]; i += 4
        else:
            n = 0; shift = 0
            while True:
                b = buf[i]; i += 1; n |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80: break
            v = bytes(buf[i:i + n]); i += n
        yield num, wt, v


def test_tensorboard_sink_writes_a_valid_event_file(tmp_path):
    """logger.jl:14-16 — TBLogger("logs/<run_name>"): the pure-Python sink must produce TFRecord-framed tensorflow.Event messages
    (masked CRC-32C on length and payload), one scalar "<message>/<key>" per numeric key, steps advanced by log_step_increment."""
    import importlib
    import struct
    lg_mod = importlib.import_module("cleanrl_jl_amd.logger")
    c = 0xFFFFFFFF
    for b in b"123456789":
        c = lg_mod._CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    assert c ^ 0xFFFFFFFF == 0xE3069283                       # CRC-32C (Castagnoli) check value
    lg = lg_mod.make_logger("run|x", to_terminal=False, to_tensorboard=True, to_json=False, log_dir=str(tmp_path))
    lg.info("Training Statistics", extra={"crl": dict(loss=1.5, pg_loss=-0.25, log_step_increment=0)})
    lg.info("Training Statistics", extra={"crl": dict(loss=1.25, pg_loss=-0.5, note="text is skipped", log_step_increment=4096)})
    lg.info("Episode Statistics", extra={"crl": dict(episode_return=37.0, episode_length=37, log_step_increment=128)})
    files = os.listdir(tmp_path / "run|x")
    assert len(files) == 1 and files[0].startswith("events.out.tfevents.")
    data = (tmp_path / "run|x" / files[0]).read_bytes()
    off, events = 0, []
    while off < len(data):
        (n,) = struct.unpack_from("<Q", data, off)
        assert struct.unpack_from("<I", data, off + 8)[0] == lg_mod._masked_crc32c(data[off:off + 8])
        body = data[off + 12:off + 12 + n]
        assert struct.unpack_from("<I", data, off + 12 + n)[0] == lg_mod._masked_crc32c(body)
        events.append({num: v for num, _, v in _pb_fields(body)})
        off += 12 + n + 4
    assert events[0][3] == b"brain.Event:2" and len(events) == 4
    steps = [e.get(2, 0) for e in events[1:]]
    assert steps == [0, 4096, 4224]
    scal = []
    for e in events[1:]:
        vals = [dict((num, v) for num, _, v in _pb_fields(v)) for num, _, v in _pb_fields(e[5]) if num == 1]
        scal.append({v[1].decode(): v[2] for v in vals})
    assert scal[0] == {"Training Statistics/loss": 1.5, "Training Statistics/pg_loss": -0.25}
    assert scal[1] == {"Training Statistics/loss": 1.25, "Training Statistics/pg_loss": -0.5}
    assert scal[2] == {"Episode Statistics/episode_return": 37.0, "Episode Statistics/episode_length": 37.0}
    for hd in list(lg.handlers):
        hd.close(); lg.removeHandler(hd)
