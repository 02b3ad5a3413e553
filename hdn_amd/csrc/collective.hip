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

}  // extern "C"
