// The path's only exchange step: one all-gather of the predicted corner offsets x[B_local, 8] per rank,
// on RCCL over xGMI, behind the C ABI so that a host without torch.distributed can gather too.
//
// RCCL is bound at run time (dlopen of its SONAME "librccl.so.1"): inside a PyTorch process this resolves to the copy
// torch already loaded — one RCCL per process, one HIP runtime — and a plain C host gets /opt/rocm/lib's.  The library
// itself therefore loads (and every other entry point works) on a box without RCCL.
//
// Payload: 32 B per pair, 2 KB per rank at B=512 over 8 GPUs — latency-bound; one collective per step, no bucketing.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "hdn_common.h"

namespace hdn {
namespace {

// the few RCCL declarations used, spelled out so the build does not depend on where rccl.h lives (ABI: rccl.h 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[HDN_RCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;          // ncclSuccess == 0
constexpr int kNcclFloat32 = 7;    // ncclFloat32 in ncclDataType_t

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("HDN_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    auto sym = [&](const char* s) { return dlsym(r.handle, s); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.AllGather;
  });
  return r;
}

inline int rccl_status(ncclResult_t e) { return e == 0 ? HDN_OK : -(2000 + (int)e); }
inline int hip_status(hipError_t e) { return e == hipSuccess ? HDN_OK : -(1000 + (int)e); }

// ---- one-shot direct-write gather --------------------------------------------------------------------------------------------
// A ring (or any RCCL algorithm) pays several hops and a proxy hand-off for a payload of 2 KB per rank; on a fully connected
// xGMI node every rank can instead STORE its slice straight into every peer's window and raise a flag there: one kernel per
// rank, one hop, no host thread in the data path.  Each rank owns a window of uncached device memory, exported by
// hipIpcGetMemHandle and mapped by the peers:
//
//   [ control: epoch, done, pad ]  [ flags[2][world] (u64) ]  [ slots[2][world][slot_bytes] ]
//
// Call e (device-resident counter, so a hipGraph replay is a valid call) uses parity e & 1.  Workgroup p of rank r: store the
// local slice into peer p's slots[parity][r], release-store e into peer p's flags[parity][r]; spin (acquire, system scope)
// on the own flags[parity][p] until it holds e, copy the own slots[parity][p] to the output.  Two parities suffice: a rank
// reaches call e + 2 only after every peer raised its flag for e + 1, which a peer does after its call e has left the stream.
// That argument needs every wait to END WITH THE FLAG: a wait that gives up (2 s) breaks it, so a timeout is terminal — the
// rows of the peer that did not show up are filled with NaN (never with whatever the slot held), a STICKY bit is set in the
// pinned status word, and hdn_gather_offsets_oneshot refuses every later call on that context (HDN_E_PEER).
constexpr int kGatherMaxWorld = 16;
constexpr unsigned long long kGatherSpinLimit = 200000000ull;   // 2 s of the 100 MHz wall clock, then status |= 1 and give up

struct GatherPeers {
  char* win[kGatherMaxWorld];
};

struct GatherCtx {
  int world = 0, rank = 0, device = 0;
  size_t slot_bytes = 0, window_bytes = 0;
  char* window = nullptr;              // own window (device, uncached)
  GatherPeers peers{};                 // [rank] = window, others = hipIpcOpenMemHandle mappings
  bool connected = false;
  unsigned* status = nullptr;          // pinned host word the kernel ORs error bits into
};

__host__ __device__ inline size_t gather_flags_off() { return 64; }
__host__ __device__ inline size_t gather_slots_off(int world) { return 64 + round_up(2 * world * 8, 64); }

__global__ __launch_bounds__(HDN_BLOCK) void gather_oneshot_kernel(const float4* __restrict__ local, float4* __restrict__ all, GatherPeers peers,
                                                                   int world, int rank, int n16, unsigned long long slot_bytes,
                                                                   unsigned* status) {
  const int p = blockIdx.x, tid = threadIdx.x;
  __shared__ int timed_out;
  if (tid == 0) timed_out = 0;
  // A context whose status word is set (an earlier call gave up on a peer) is unusable: the host entry point refuses it, and a
  // launch that reaches the device anyway — the replay of a hipGraph captured before the timeout — hands out NaN rows, never
  // the slots' stale contents, and leaves the window alone.
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
    const float qnan = __builtin_nanf("");
    for (int i = tid; i < n16; i += HDN_BLOCK) all[size_t(p) * n16 + i] = float4{qnan, qnan, qnan, qnan};
    return;
  }
  char* mine = peers.win[rank];
  unsigned long long* ctl = reinterpret_cast<unsigned long long*>(mine);
  const unsigned long long epoch = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;   // the same for every workgroup
  const int parity = (int)(epoch & 1);
  // push
  char* theirs = peers.win[p];
  float4* dst = reinterpret_cast<float4*>(theirs + gather_slots_off(world) + (size_t(parity) * world + rank) * slot_bytes);
  for (int i = tid; i < n16; i += HDN_BLOCK) dst[i] = local[i];
  __atomic_thread_fence(__ATOMIC_RELEASE);   // (clang: system scope)
  __syncthreads();
  if (tid == 0) {
    unsigned long long* f = reinterpret_cast<unsigned long long*>(theirs + gather_flags_off()) + (parity * world + rank);
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for peer p's slice in the own window
    unsigned long long* g = reinterpret_cast<unsigned long long*>(mine + gather_flags_off()) + (parity * world + p);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      if (wall_clock64() - t0 > kGatherSpinLimit) {
        __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (timed_out) {   // the peer never delivered: its rows are NaN, not stale slot contents
    const float qnan = __builtin_nanf("");
    for (int i = tid; i < n16; i += HDN_BLOCK) all[size_t(p) * n16 + i] = float4{qnan, qnan, qnan, qnan};
  } else {
    const float4* src = reinterpret_cast<const float4*>(mine + gather_slots_off(world) + (size_t(parity) * world + p) * slot_bytes);
    for (int i = tid; i < n16; i += HDN_BLOCK) all[size_t(p) * n16 + i] = src[i];
  }
  // the last workgroup to finish publishes the epoch for the next call (stream order makes it visible to that launch)
  __syncthreads();
  if (tid == 0) {
    const unsigned long long d = __hip_atomic_fetch_add(ctl + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (d + 1 == (unsigned long long)world) {
      __hip_atomic_store(ctl + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace
}  // namespace hdn

extern "C" {

int hdn_rccl_available(void) { return hdn::rccl().ok ? 1 : 0; }

int hdn_rccl_unique_id(void* id128) {
  if (!id128) return HDN_E_NULL;
  const hdn::Rccl& r = hdn::rccl();
  if (!r.ok) return HDN_E_NORCCL;
  return hdn::rccl_status(r.GetUniqueId(static_cast<hdn::ncclUniqueId*>(id128)));
}

int hdn_rccl_comm_create(void** comm_out, int world, int rank, const void* id128) {
  if (!comm_out || !id128) return HDN_E_NULL;
  if (world <= 0 || rank < 0 || rank >= world) return HDN_E_SHAPE;
  const hdn::Rccl& r = hdn::rccl();
  if (!r.ok) return HDN_E_NORCCL;
  hdn::ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  hdn::ncclComm_t c = nullptr;
  const int rc = hdn::rccl_status(r.CommInitRank(&c, world, id, rank));  // binds the comm to the CURRENT device
  if (rc == HDN_OK) *comm_out = c;
  return rc;
}

// ranks in the communicator, as RCCL itself reports it (ncclCommCount): what bench.py --gpus N prints as collective.comm_ranks
int hdn_rccl_comm_count(void* comm, int* count) {
  if (!comm || !count) return HDN_E_NULL;
  const hdn::Rccl& r = hdn::rccl();
  if (!r.ok) return HDN_E_NORCCL;
  return hdn::rccl_status(r.CommCount(static_cast<hdn::ncclComm_t>(comm), count));
}

int hdn_rccl_comm_destroy(void* comm) {
  if (!comm) return HDN_E_NULL;
  const hdn::Rccl& r = hdn::rccl();
  if (!r.ok) return HDN_E_NORCCL;
  return hdn::rccl_status(r.CommDestroy(static_cast<hdn::ncclComm_t>(comm)));
}

int hdn_allgather_offsets(const float* local, float* all, int Bl, void* rccl_comm, void* stream) {
  if (!local || !all || !rccl_comm) return HDN_E_NULL;
  if (Bl <= 0) return HDN_E_SHAPE;
  if (Bl > (1 << 24)) return HDN_E_LIMIT;
  const hdn::Rccl& r = hdn::rccl();
  if (!r.ok) return HDN_E_NORCCL;
  hdn::ncclComm_t c = static_cast<hdn::ncclComm_t>(rccl_comm);
  int world = 0, rank = 0;
  int rc = hdn::rccl_status(r.CommCount(c, &world));
  if (rc != HDN_OK) return rc;
  rc = hdn::rccl_status(r.CommUserRank(c, &rank));
  if (rc != HDN_OK) return rc;
  // in place is allowed exactly as ncclAllGather defines it (local == all + rank * Bl * 8); any other overlap is an error
  const float* mine = all + (size_t)rank * Bl * 8;
  if (local != mine && local < all + (size_t)world * Bl * 8 && all < local + (size_t)Bl * 8) return HDN_E_ALIAS;
  return hdn::rccl_status(r.AllGather(local, all, (size_t)Bl * 8, hdn::kNcclFloat32, c, static_cast<hipStream_t>(stream)));
}


int hdn_gather_create(void** ctx_out, int world, int rank, long long slot_bytes) {
  if (!ctx_out) return HDN_E_NULL;
  if (world <= 0 || rank < 0 || rank >= world || slot_bytes <= 0 || slot_bytes % 16 != 0) return HDN_E_SHAPE;
  if (world > hdn::kGatherMaxWorld || slot_bytes > (1ll << 26)) return HDN_E_LIMIT;
  hdn::GatherCtx* c = new hdn::GatherCtx;
  c->world = world; c->rank = rank; c->slot_bytes = (size_t)slot_bytes;
  (void)hipGetDevice(&c->device);
  c->window_bytes = hdn::gather_slots_off(world) + 2 * (size_t)world * c->slot_bytes;
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->window), c->window_bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(c->window, 0, c->window_bytes);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->status), sizeof(unsigned), hipHostMallocMapped);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->window) (void)hipFree(c->window);
    if (c->status) (void)hipHostFree(c->status);
    delete c;
    return hdn::hip_status(e);
  }
  *c->status = 0;
  c->peers.win[rank] = c->window;
  *ctx_out = c;
  return HDN_OK;
}

int hdn_gather_handle(void* ctx, void* handle64) {
  if (!ctx || !handle64) return HDN_E_NULL;
  hdn::GatherCtx* c = static_cast<hdn::GatherCtx*>(ctx);
  static_assert(sizeof(hipIpcMemHandle_t) == HDN_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, c->window);
  if (e != hipSuccess) return hdn::hip_status(e);
  memcpy(handle64, &h, sizeof h);
  return HDN_OK;
}

int hdn_gather_connect(void* ctx, const void* handles) {
  if (!ctx || !handles) return HDN_E_NULL;
  hdn::GatherCtx* c = static_cast<hdn::GatherCtx*>(ctx);
  if (c->connected) return HDN_E_SHAPE;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(handles) + (size_t)p * HDN_IPC_HANDLE_BYTES, sizeof h);
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      for (int q = 0; q < p; ++q)
        if (q != c->rank && c->peers.win[q]) { (void)hipIpcCloseMemHandle(c->peers.win[q]); c->peers.win[q] = nullptr; }
      return hdn::hip_status(e);
    }
    c->peers.win[p] = static_cast<char*>(ptr);
  }
  c->connected = true;
  return HDN_OK;
}

int hdn_gather_offsets_oneshot(void* ctx, const float* local, float* all, int Bl, void* stream) {
  if (!ctx || !local || !all) return HDN_E_NULL;
  hdn::GatherCtx* c = static_cast<hdn::GatherCtx*>(ctx);
  if (Bl <= 0 || !c->connected) return HDN_E_SHAPE;
  if (__atomic_load_n(c->status, __ATOMIC_ACQUIRE) != 0) return HDN_E_PEER;   // an earlier call timed out: the epoch protocol is broken for good
  const size_t bytes = (size_t)Bl * 8 * sizeof(float);
  if (bytes > c->slot_bytes) return HDN_E_LIMIT;
  if (!hdn::aligned16(local) || !hdn::aligned16(all)) return HDN_E_SHAPE;
  if (local < all + (size_t)c->world * Bl * 8 && all < local + (size_t)Bl * 8) return HDN_E_ALIAS;
  hipLaunchKernelGGL(hdn::gather_oneshot_kernel, dim3(c->world), dim3(HDN_BLOCK), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(local), reinterpret_cast<float4*>(all), c->peers, c->world, c->rank, (int)(bytes / 16),
                     (unsigned long long)c->slot_bytes, c->status);
  return hdn::launch_status();
}

int hdn_gather_status(void* ctx) {
  if (!ctx) return HDN_E_NULL;
  return (int)__atomic_load_n(static_cast<hdn::GatherCtx*>(ctx)->status, __ATOMIC_ACQUIRE);
}

// 1 / 0: can `device` of this process read and write the memory of the device with PCI bus id `peer_pci_bus_id` (as
// hdn_device_pci_bus_id printed it in the peer's process)?  The same device counts as accessible.  HDN_E_SHAPE: that device is not
// visible to this process (HIP_VISIBLE_DEVICES isolation): unknown.
int hdn_gather_peer_access(int device, const char* peer_pci_bus_id) {
  if (!peer_pci_bus_id) return HDN_E_NULL;
  int peer = -1;
  if (hipDeviceGetByPCIBusId(&peer, peer_pci_bus_id) != hipSuccess || peer < 0) {
    (void)hipGetLastError();
    return HDN_E_SHAPE;
  }
  if (peer == device) return 1;
  int can = 0;
  const hipError_t e = hipDeviceCanAccessPeer(&can, device, peer);
  if (e != hipSuccess) return hdn::hip_status(e);
  return can ? 1 : 0;
}

int hdn_device_pci_bus_id(int device, char* out, int len) {
  if (!out) return HDN_E_NULL;
  if (len < 16) return HDN_E_SHAPE;
  return hdn::hip_status(hipDeviceGetPCIBusId(out, len, device));
}

// Every rank must have stopped calling AND every launch that stores into this window must have completed: run a group barrier
// after a device synchronisation on every rank first (hdn_amd.dist.OneShotGather.destroy does).  The own device is synchronised here.
int hdn_gather_destroy(void* ctx) {
  if (!ctx) return HDN_E_NULL;
  hdn::GatherCtx* c = static_cast<hdn::GatherCtx*>(ctx);
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();          // no launch of this rank is still reading the window or storing into a peer's
  hipError_t e = hipSuccess;
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->peers.win[p]) {
      const hipError_t e2 = hipIpcCloseMemHandle(c->peers.win[p]);
      if (e == hipSuccess) e = e2;
    }
  const hipError_t e3 = hipFree(c->window);
  if (e == hipSuccess) e = e3;
  (void)hipHostFree(c->status);
  if (cur != c->device) (void)hipSetDevice(cur);
  delete c;
  return hdn::hip_status(e);
}

}  // extern "C"
