// comm.cpp — data-parallel gradient exchange over RCCL (xGMI). One communicator per handle, one process per GPU.
// The reference has no distributed path at all; the cadence follows ppo.jl:250 — one optimiser step per
// minibatch ⇒ one all-reduce of the flat gradient (+4 loss scalars) per minibatch, enqueued on the handle's stream so
// it stays ordered with the kernels around it and never returns to the host.
// librccl is opened lazily (dlopen) so single-GPU users never load it.
#include <dlfcn.h>
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
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
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

void comm_destroy(crl_ppo* h) {
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(h->comm));
  h->comm = nullptr;
}

}  // namespace crl
