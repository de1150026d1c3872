// RCCL behind the C-ABI: the one data-path collective of the training step -- the SUM all-reduce of the flat
// fp32 projector-gradient bucket, issued in completion-ordered ranges on a side HIP stream (ps_slm_amd/engine.py) -- replaces the
// DeepSpeed engine's ZeRO-2 reduce-scatter / all-gather pair (Multitask/finetune_deepspeed.py:147-149, conf/ds_config.json:15-21).
//
// RCCL is bound at RUN time (dlopen): the library is already in the process when the host framework has loaded it, and a
// second copy must not be linked in beside it; building libtasu_hip.so needs no RCCL headers.  The few declarations below restate
// rccl.h (RCCL 2.22+: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclInt32 = 2, ncclSum = 0, ncclMin = 3).
// Host-side code; one communicator per process (one process per GPU), created on the CURRENT device.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

struct UniqueId {
  char internal[128];
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* forced = getenv("TASU_RCCL_PATH");
    // the copy the host framework has already loaded, if any (RTLD_NOLOAD), else the ROCm installation's
    const char* names[] = {forced, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !x.handle; ++pass)
      for (const char* n : names) {
        if (!n) continue;
        x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (x.handle) break;
      }
    if (!x.handle) return x;
    x.get_unique_id = (GetUniqueIdFn)dlsym(x.handle, "ncclGetUniqueId");
    x.comm_init_rank = (CommInitRankFn)dlsym(x.handle, "ncclCommInitRank");
    x.comm_destroy = (CommDestroyFn)dlsym(x.handle, "ncclCommDestroy");
    x.all_reduce = (AllReduceFn)dlsym(x.handle, "ncclAllReduce");
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce;
    return x;
  }();
  return r;
}

constexpr int NCCL_FLOAT32 = 7, NCCL_INT32 = 2, NCCL_SUM = 0, NCCL_MIN = 3;

}  // namespace

extern "C" int tasu_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int tasu_comm_unique_id(uint8_t* id128) {
  if (!id128) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  UniqueId id;
  if (rccl().get_unique_id(&id) != 0) return TASU_ERR_LAUNCH;
  memcpy(id128, id.internal, 128);
  return TASU_OK;
}

extern "C" int tasu_comm_init(const uint8_t* id128, int rank, int world, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  UniqueId id;
  memcpy(id.internal, id128, 128);
  Comm c = nullptr;
  if (rccl().comm_init_rank(&c, world, id, rank) != 0 || !c) return TASU_ERR_LAUNCH;
  *comm = c;
  return TASU_OK;
}

extern "C" int tasu_comm_destroy(void* comm) {
  if (!comm) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  return rccl().comm_destroy((Comm)comm) == 0 ? TASU_OK : TASU_ERR_LAUNCH;
}

extern "C" int tasu_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  return rccl().all_reduce(buf, buf, (size_t)n, NCCL_FLOAT32, NCCL_SUM, (Comm)comm, (hipStream_t)stream) == 0 ? TASU_OK : TASU_ERR_LAUNCH;
}

extern "C" int tasu_allreduce_min_i32(void* comm, int32_t* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  return rccl().all_reduce(buf, buf, (size_t)n, NCCL_INT32, NCCL_MIN, (Comm)comm, (hipStream_t)stream) == 0 ? TASU_OK : TASU_ERR_LAUNCH;
}
