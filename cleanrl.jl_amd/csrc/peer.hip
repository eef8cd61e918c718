// peer.hip — one-shot all-reduce over peer-mapped mailboxes (xGMI), the alternative to RCCL for the small per-step messages.
//
// The data-parallel path exchanges one 36.6 KB gradient message per optimiser step (ppo.jl:250 is the step the reference
// takes per minibatch) and one f64 advantage-sums message per iteration. At 8192 envs per GPU the whole iteration is ≈3 ms, so
// a ring all-reduce's per-hop latency is a visible share of it. xGMI is point-to-point and every GPU reaches every other in one
// hop, so for a message this small the cheapest exchange is a single push: every rank writes its vector into its own slot of
// every peer's mailbox, raises a flag there, waits for the W flags in its own mailbox and adds the W slots IN RANK ORDER — all
// ranks compute the same sum in the same order, so the replicas stay bit-identical (the property crl_ppo_iterate's guard window
// and the tests rely on). One launch, no host involvement, no ring.
//
// Mailbox (one per rank, uncached device memory shared through hipIpc handles; the host launcher moves the 64-byte handles):
//   flags [2 parities][W ranks][nblk] u32   — the sequence number of the message whose 256-element chunk has landed
//   slots [2 parities][W ranks][slot_bytes] — the chunks themselves
// A workgroup owns one 256-element chunk end to end (push → flag → wait → sum), so chunks never wait on each other and a grid
// larger than the chip still drains. Parity = sequence & 1: a rank can only be two messages ahead of a peer's slot after that
// peer has raised the flag of the message in between, which it does after it finished reading the older one (stream order).
// A lost peer turns into a time-out (option peer_timeout_ms, default 20 s) that raises a sticky error word, never a hang.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ppo_ctx.hpp"

namespace crl {

constexpr int PEER_CHUNK = 256;

struct PeerState {
  char* box[PEER_MAX] = {};
  bool opened[PEER_MAX] = {};
  int world = 0, rank = 0, nblk = 0;     // nblk: flag words per (parity, rank) — one per 64-float chunk of the largest message (the one-launch
                                         // optimiser step exchanges 64-float chunks; peer_allreduce_kernel's 256-element chunks use the first of them)
  int cap256 = 0;                        // 256-element chunks a slot holds
  int on_my_device = 1;                  // ranks whose mailbox lives on THIS device (1 = one process per GPU; more = a shared GPU, functional runs)
  uint32_t seq = 0;
  size_t slot_bytes = 0, data_off = 0, box_bytes = 0;
  uint32_t* err = nullptr;
  bool attached = false;
};

template <typename T>
__global__ __launch_bounds__(PEER_CHUNK) void peer_allreduce_kernel(PeerArgs a, T* __restrict__ buf, size_t count) {
  const int par = (int)(a.seq & 1u), W = a.world;
  const size_t i = (size_t)blockIdx.x * PEER_CHUNK + threadIdx.x;
  const bool live = i < count;
  const T v = live ? buf[i] : T(0);
  // push: my chunk into my slot of every mailbox, starting with my right-hand neighbour so the W ranks spread over the links
  for (int d = 1; d <= W; ++d) {
    const int p = (a.rank + d) % W;
    T* dst = reinterpret_cast<T*>(a.box[p] + a.data_off + ((size_t)par * W + a.rank) * a.slot_bytes);
    if (live) __builtin_nontemporal_store(v, dst + i);
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < W) {
    const int p = threadIdx.x;
    uint32_t* f = reinterpret_cast<uint32_t*>(a.box[p]) + ((size_t)par * W + a.rank) * a.nblk + blockIdx.x;
    __hip_atomic_store(f, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for rank p's chunk in MY mailbox
    const uint32_t* g = reinterpret_cast<const uint32_t*>(a.box[a.rank]) + ((size_t)par * W + p) * a.nblk + blockIdx.x;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeout_ticks) { atomicExch(a.err, 1u); break; }
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (!live) return;
  const char* mine = a.box[a.rank] + a.data_off + (size_t)par * W * a.slot_bytes;
  // The slots were written by OTHER devices over xGMI while this kernel was already running, and this rank read the same parity's
  // slots two messages ago: every slot word is read with a system-scope atomic load, which is served at the memory side and can
  // never be a stale line of this device's L2 or of this CU's vector L1 (the mailbox is fine-grained / uncached memory on top).
  using U = typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type;
  T s = T(0);
  for (int r = 0; r < W; ++r) {
    const U bits = __hip_atomic_load(reinterpret_cast<const U*>(mine + (size_t)r * a.slot_bytes) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    T v;
    __builtin_memcpy(&v, &bits, sizeof(T));
    s += v;
  }
  buf[i] = s;
}

static PeerState* peer_of(crl_ppo* h) { return static_cast<PeerState*>(h->peer); }

// Allocates this rank's mailbox and returns its IPC handle. slot_bytes covers the largest message the handle can send:
// the gradient message (P + 8 floats) and the advantage sums (update_epochs × num_minibatches × 2 doubles).
int peer_export(crl_ppo* h, int world, int rank, uint8_t handle[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (world < 1 || world > PEER_MAX || rank < 0 || rank >= world) {
    set_error("crl_comm_peer_export: need 1 <= world_size <= 16 and 0 <= rank < world_size"); return 1;
  }
  if (h->peer || h->comm || h->external_comm) { set_error("crl_comm_peer_export: a communicator is already attached (crl_comm_destroy detaches it)"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(h->device));
  PeerState* s = new PeerState;
  s->world = world; s->rank = rank;
  size_t msg = ((size_t)h->P + 8) * sizeof(float);
  const size_t adv = (size_t)h->cfg.update_epochs * h->cfg.num_minibatches * 2 * sizeof(double);
  if (adv > msg) msg = adv;
  s->cap256 = (int)((msg / 4 + PEER_CHUNK - 1) / PEER_CHUNK);
  s->nblk = (int)((msg / 4 + 63) / 64);
  s->slot_bytes = (size_t)s->cap256 * PEER_CHUNK * sizeof(double);   // a chunk is 256 ELEMENTS of either type
  s->data_off = (((size_t)2 * world * s->nblk * sizeof(uint32_t)) + 255) & ~(size_t)255;
  s->box_bytes = s->data_off + (size_t)2 * world * s->slot_bytes;
  // The protocol needs a peer's stores over xGMI to become visible inside an already-running kernel of the home GPU: the mailbox
  // must be uncached or fine-grained device memory. Ordinary (coarse-grained) hipMalloc memory is NOT a fallback — the home L2
  // could serve stale slot lines and the sums would be silently wrong — so a failed allocation is an error.
  void* box = nullptr;
  hipError_t e = hipExtMallocWithFlags(&box, s->box_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); box = nullptr; e = hipExtMallocWithFlags(&box, s->box_bytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    delete s;
    set_error(std::string("peer mailbox: neither uncached nor fine-grained device memory could be allocated (") + hipGetErrorString(e) +
              "); the peer all-reduce is refused rather than run on cached memory — use crl_comm_init (RCCL)");
    return 1;
  }
  s->box[rank] = static_cast<char*>(box);
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&s->err), sizeof(uint32_t));
  if (e2 != hipSuccess) { (void)hipFree(box); delete s; set_error("peer error word allocation failed"); return 1; }
  CRL_HIP_CHECK(hipMemset(box, 0, s->box_bytes));
  CRL_HIP_CHECK(hipMemset(s->err, 0, sizeof(uint32_t)));
  CRL_HIP_CHECK(hipDeviceSynchronize());
  hipIpcMemHandle_t ih;
  e = hipIpcGetMemHandle(&ih, box);
  if (e != hipSuccess) {
    (void)hipFree(box); (void)hipFree(s->err); delete s;
    set_error(std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)"); return 1;
  }
  std::memcpy(handle, &ih, 64);
  h->peer = s;
  h->world = world; h->rank = rank;
  return 0;
}

// handles: world × 64 bytes in rank order (this rank's own entry is ignored). Every rank must have exported before any rank
// attaches — the launcher's all-gather of the handles is that barrier.
int peer_attach(crl_ppo* h, const uint8_t* handles) {
  PeerState* s = peer_of(h);
  if (!s) { set_error("crl_comm_peer_attach: call crl_comm_peer_export first"); return 1; }
  if (s->attached) { set_error("crl_comm_peer_attach: already attached"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(h->device));
  for (int p = 0; p < s->world; ++p) {
    if (p == s->rank) continue;
    hipIpcMemHandle_t ih;
    std::memcpy(&ih, handles + (size_t)p * 64, 64);
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, ih, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error(std::string("hipIpcOpenMemHandle (rank ") + std::to_string(p) + "): " + hipGetErrorString(e)); return 1;
    }
    s->box[p] = static_cast<char*>(ptr); s->opened[p] = true;
    // a mailbox that lives on another device must be reachable from this one over xGMI / PCIe peer access
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.device == h->device) s->on_my_device += 1;
    if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.device >= 0 && at.device != h->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, h->device, at.device) == hipSuccess && !can) {
        set_error("crl_comm_peer_attach: device " + std::to_string(h->device) + " cannot access the mailbox of rank " + std::to_string(p) +
                  " on device " + std::to_string(at.device) + " (hipDeviceCanAccessPeer = 0); use crl_comm_init (RCCL)");
        return 1;
      }
    } else (void)hipGetLastError();
  }
  s->attached = true;
  return 0;
}

bool peer_active(const crl_ppo* h) { return h->peer && static_cast<const PeerState*>(h->peer)->attached; }
int peer_ranks_on_my_device(const crl_ppo* h) { return h->peer ? static_cast<const PeerState*>(h->peer)->on_my_device : 1; }

int peer_allreduce(crl_ppo* h, void* buf, size_t count, bool is_double) {
  PeerState* s = peer_of(h);
  if (!s || !s->attached) { set_error("peer all-reduce before crl_comm_peer_attach"); return 1; }
  const size_t bytes = count * (is_double ? sizeof(double) : sizeof(float));
  const int nblk = (int)((count + PEER_CHUNK - 1) / PEER_CHUNK);
  if (nblk > s->cap256 || bytes > s->slot_bytes) { set_error("peer all-reduce: message larger than the mailbox slot"); return 1; }
  if (count == 0) return 0;
  const double timeout_s = (double)opt(h, OPT_PEER_TIMEOUT_MS) * 1e-3;
  PeerArgs a;
  for (int p = 0; p < PEER_MAX; ++p) a.box[p] = s->box[p];
  a.world = s->world; a.rank = s->rank; a.seq = ++s->seq; a.nblk = s->nblk; a.slot_bytes = s->slot_bytes; a.data_off = s->data_off;
  a.err = s->err; a.timeout_ticks = (long long)(timeout_s * 1e8);   // wall_clock64 ticks at 100 MHz
  if (is_double)
    hipLaunchKernelGGL(peer_allreduce_kernel<double>, dim3(nblk), dim3(PEER_CHUNK), 0, h->stream, a, static_cast<double*>(buf), count);
  else
    hipLaunchKernelGGL(peer_allreduce_kernel<float>, dim3(nblk), dim3(PEER_CHUNK), 0, h->stream, a, static_cast<float*>(buf), count);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int peer_next_args(crl_ppo* h, PeerArgs* a, int chunks, size_t floats) {
  PeerState* s = peer_of(h);
  if (!s || !s->attached) { set_error("peer exchange before crl_comm_peer_attach"); return 1; }
  if (chunks > s->nblk || floats * sizeof(float) > s->slot_bytes) { set_error("peer exchange: message larger than the mailbox slot"); return 1; }
  for (int p = 0; p < PEER_MAX; ++p) a->box[p] = s->box[p];
  a->world = s->world; a->rank = s->rank; a->seq = ++s->seq; a->nblk = s->nblk; a->slot_bytes = s->slot_bytes; a->data_off = s->data_off;
  a->err = s->err; a->timeout_ticks = (long long)((double)opt(h, OPT_PEER_TIMEOUT_MS) * 1e-3 * 1e8);
  return 0;
}

// Sticky time-out word (read where the other sticky flags are read: crl_sync and the ends of guard windows).
const uint32_t* peer_err_word(const crl_ppo* h) {
  const PeerState* s = static_cast<const PeerState*>(h->peer);
  return (s && s->attached) ? s->err : nullptr;
}
int peer_check(crl_ppo* h) {
  PeerState* s = peer_of(h);
  if (!s || !s->attached) return 0;
  uint32_t e = 0;
  CRL_HIP_CHECK(hipMemcpyAsync(&e, s->err, sizeof(e), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (e) { set_error("peer all-reduce timed out waiting for another rank (a rank died or the ranks issued different collectives)"); return 1; }
  return 0;
}

void peer_destroy(crl_ppo* h) {
  PeerState* s = peer_of(h);
  if (!s) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();      // my last all-reduce has finished => no peer writes into my mailbox any more (see header)
  for (int p = 0; p < s->world; ++p)
    if (s->opened[p]) (void)hipIpcCloseMemHandle(s->box[p]);
  if (s->box[s->rank]) (void)hipFree(s->box[s->rank]);
  if (s->err) (void)hipFree(s->err);
  delete s;
  h->peer = nullptr;
}

}  // namespace crl
