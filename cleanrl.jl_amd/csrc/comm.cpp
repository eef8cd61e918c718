// comm.cpp — data-parallel gradient exchange over RCCL (xGMI). One communicator per handle, one process per GPU.
// The reference has no distributed path at all; the cadence follows ppo.jl:250 — one optimiser step per
// minibatch ⇒ one all-reduce of the flat gradient (+4 loss scalars) per minibatch, enqueued on the handle's stream so
// it stays ordered with the kernels around it and never returns to the host.
// librccl is opened lazily (dlopen) so single-GPU users never load it.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "ppo_ctx.hpp"

namespace crl {

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;

// A librccl that is ALREADY mapped into this process wins over the soname search: a host framework (torch ships its own copy under torch/lib) and this
// library must run their collectives on ONE RCCL — two copies in a process each register with the runtime and crash at exit (seen here: crl first,
// torch second → "double free or corruption"). With the launcher importing torch first, dlopen by soname finds torch's copy anyway (same SONAME,
// librccl.so.1); the scan also covers a copy mapped under another name. The reverse order cannot be mended from here: import torch before the first
// crl_comm_* call (cleanrl.jl_amd/dist.py and bench.py do).
int find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out) {
  const char* name = info->dlpi_name;
  if (!name || !*name) return 0;
  const char* base = std::strrchr(name, '/');
  base = base ? base + 1 : name;
  if (std::strncmp(base, "librccl.so", 10) != 0) return 0;
  *static_cast<std::string*>(out) = name;
  return 1;
}

int load_rccl() {
  if (g_rccl.lib) return 0;
  std::string mapped;
  dl_iterate_phdr(find_mapped_rccl, &mapped);
  if (!mapped.empty()) g_rccl.lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_GLOBAL);
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    if (g_rccl.lib) break;
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_rccl.lib) { set_error(std::string("cannot load librccl: ") + dlerror()); return 1; }
#define CRL_SYM(field, name)                                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name)); \
  if (!g_rccl.field) { set_error(std::string("librccl lacks ") + name); return 1; }
  CRL_SYM(GetUniqueId, "ncclGetUniqueId")
  CRL_SYM(CommInitRank, "ncclCommInitRank")
  CRL_SYM(AllReduce, "ncclAllReduce")
  CRL_SYM(CommDestroy, "ncclCommDestroy")
  CRL_SYM(GetErrorString, "ncclGetErrorString")
  CRL_SYM(GetVersion, "ncclGetVersion")
#undef CRL_SYM
  return 0;
}
int nccl_fail(const char* what, ncclResult_t r) {
  set_error(std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
  return 1;
}
}  // namespace

int comm_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (load_rccl()) return 1;
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  std::memcpy(id, &u, 128);
  return 0;
}

int comm_init(crl_ppo* h, const uint8_t id[128], int world, int rank) {
  if (world < 1 || rank < 0 || rank >= world) { set_error("crl_comm_init: bad world/rank"); return 1; }
  if (h->peer) { set_error("crl_comm_init: a peer communicator is already attached"); return 1; }
  h->world = world; h->rank = rank;
  // world 1 needs no communicator; option comm_force = 1 creates one anyway so a 1-GPU box can exercise the RCCL path
  if (world == 1 && !opt(h, OPT_COMM_FORCE)) return 0;
  if (load_rccl()) return 1;
  CRL_HIP_CHECK(hipSetDevice(h->device));
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclComm_t c = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&c, world, u, rank);
  if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
  h->comm = c;
  return 0;
}

int comm_allreduce(crl_ppo* h, void* buf, size_t count, bool is_double) {
  if (h->peer) return peer_allreduce(h, buf, count, is_double);
  if (!h->comm) {
    if (h->world == 1 || h->external_comm) return 0;
    set_error("all-reduce requested but no communicator attached (crl_comm_init)"); return 1;
  }
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, is_double ? ncclDouble : ncclFloat, ncclSum,
                                    static_cast<ncclComm_t>(h->comm), h->stream);
  if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
  return 0;
}

// Which librccl this process's exchange runs on: the file the loaded ncclAllReduce lives in (dladdr) and ncclGetVersion. A launcher that imports
// torch has torch's own librccl mapped already; dlopen by soname then resolves to whichever copy the loader finds first — the bench line records this
// per rank so that the first real N > 1 run shows at a glance that all ranks, and torch, use ONE RCCL (verdict r5 item 7).
int comm_info(char* path, size_t path_cap, int* version) {
  if (load_rccl()) return 1;
  Dl_info info;
  std::memset(&info, 0, sizeof(info));
  const char* fname = "?";
  if (dladdr(reinterpret_cast<void*>(g_rccl.AllReduce), &info) && info.dli_fname) fname = info.dli_fname;
  if (path && path_cap) { std::strncpy(path, fname, path_cap - 1); path[path_cap - 1] = 0; }
  if (version) {
    int v = 0;
    ncclResult_t r = g_rccl.GetVersion(&v);
    if (r != ncclSuccess) return nccl_fail("ncclGetVersion", r);
    *version = v;
  }
  return 0;
}

void comm_destroy(crl_ppo* h) {
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(h->comm));
  h->comm = nullptr;
}

}  // namespace crl
