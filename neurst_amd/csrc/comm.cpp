// a19 over the C ABI (SURVEY 8(b) export list: neurst_comm_init / neurst_allreduce_bucket / neurst_comm_destroy): the gradient
// exchange of st/tools/tftools/gradientTools.py:7-37 (average_gradients: the towers' gradients summed, the 1/N folded into
// the optimizer) for a host WITHOUT torch.distributed - one process per GPU, one RCCL communicator per process, buckets of the
// gradient arena summed in place on a stream of the caller's.  The Python host of this repository uses torch.distributed
// (backend "nccl" = RCCL) for the same collective, as the launch contract prescribes; this file is the same exchange for any
// other host language.
//
// RCCL is bound at RUN time (dlopen / dlsym on the first neurst_comm_* call), not linked: a process whose Python side already
// carries torch's own librccl.so must not see a second copy's symbols, and hosts that never exchange gradients need no RCCL
// at all.  NEURST_RCCL_LIB names the library (default: librccl.so.1, then librccl.so).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/neurst_hip.h"

void neurst_set_error(const char* fmt, ...);   // elementwise.hip

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

bool bind_rccl() {
  if (g_rccl.lib) return true;
  const char* env = getenv("NEURST_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    neurst_set_error("comm: cannot load RCCL (%s)", dlerror());
    return false;
  }
  Rccl r;
  r.lib = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
    neurst_set_error("comm: the RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString");
    dlclose(h);
    return false;
  }
  g_rccl = r;
  return true;
}
}  // namespace

extern "C" int neurst_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int neurst_comm_unique_id(void* id_out) {
  if (!id_out) { neurst_set_error("comm_unique_id: null buffer"); return NEURST_E_INVALID; }
  if (!bind_rccl()) return NEURST_E_UNSUPPORTED;
  ncclUniqueId id;
  const ncclResult_t rc = g_rccl.GetUniqueId(&id);
  if (rc != ncclSuccess) { neurst_set_error("comm_unique_id: %s", g_rccl.GetErrorString(rc)); return NEURST_E_LAUNCH; }
  memcpy(id_out, &id, sizeof(id));
  return NEURST_OK;
}

extern "C" int neurst_comm_init(int rank, int world, const void* unique_id) {
  if (world < 1 || rank < 0 || rank >= world || !unique_id) { neurst_set_error("comm_init: bad rank %d of %d or null id", rank, world); return NEURST_E_INVALID; }
  if (g_comm) { neurst_set_error("comm_init: a communicator already exists (one per process; neurst_comm_destroy first)"); return NEURST_E_INVALID; }
  if (!bind_rccl()) return NEURST_E_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  const ncclResult_t rc = g_rccl.CommInitRank(&g_comm, world, id, rank);   // binds the communicator to the CURRENT device
  if (rc != ncclSuccess) { g_comm = nullptr; neurst_set_error("comm_init: %s", g_rccl.GetErrorString(rc)); return NEURST_E_LAUNCH; }
  g_world = world;
  g_rank = rank;
  return NEURST_OK;
}

extern "C" int neurst_comm_world(void) { return g_comm ? g_world : 0; }

extern "C" int neurst_allreduce_bucket(void* ptr, int64_t count, int dtype, void* stream) {
  if (!g_comm) { neurst_set_error("allreduce_bucket: no communicator (neurst_comm_init)"); return NEURST_E_INVALID; }
  if (!ptr || count < 0 || (dtype != NEURST_F32 && dtype != NEURST_BF16)) { neurst_set_error("allreduce_bucket: bad arguments"); return NEURST_E_INVALID; }
  if (count == 0) return NEURST_OK;
  const ncclResult_t rc = g_rccl.AllReduce(ptr, ptr, (size_t)count, dtype == NEURST_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, g_comm,
                                           reinterpret_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) { neurst_set_error("allreduce_bucket: %s", g_rccl.GetErrorString(rc)); return NEURST_E_LAUNCH; }
  return NEURST_OK;
}

extern "C" int neurst_comm_destroy(void) {
  if (!g_comm) return NEURST_OK;
  const ncclResult_t rc = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  g_rank = -1;
  if (rc != ncclSuccess) { neurst_set_error("comm_destroy: %s", g_rccl.GetErrorString(rc)); return NEURST_E_LAUNCH; }
  return NEURST_OK;
}
