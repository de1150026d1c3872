// RCCL behind the C-ABI: the one data-path collective of the training step -- the SUM all-reduce of the flat
// fp32 projector-gradient bucket, issued in completion-ordered ranges on a side HIP stream (ps_slm_amd/engine.py) -- replaces the
// DeepSpeed engine's ZeRO-2 reduce-scatter / all-gather pair (Multitask/finetune_deepspeed.py:147-149, conf/ds_config.json:15-21).
//
// RCCL is bound at RUN time (dlopen): the library is already in the process when the host framework has loaded it, and a
// second copy must not be linked in beside it; building libtasu_hip.so needs no RCCL headers.  The few declarations below restate
// rccl.h (RCCL 2.22+: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclInt32 = 2, ncclSum = 0, ncclMin = 3).
// Host-side code; one communicator per process (one process per GPU), created on the CURRENT device.
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
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
typedef int (*CommCountFn)(Comm, int*);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommCountFn comm_count = nullptr;
  bool ok = false;
  char path[512] = {0};      // the file the functions were bound from
  char error[640] = {0};     // why ok is false
};

// Shared objects already mapped into the process whose file name contains "librccl" (the host framework's copy: torch ships
// its own under torch/lib, which no dlopen("librccl.so.1", RTLD_NOLOAD) by soname would find when it was loaded by path).
struct Loaded {
  char paths[4][512];
  int n = 0;
};
int note_rccl(struct dl_phdr_info* info, size_t, void* data) {
  Loaded* l = (Loaded*)data;
  const char* name = info->dlpi_name;
  if (!name || !*name) return 0;
  const char* base = strrchr(name, '/');
  base = base ? base + 1 : name;
  if (strstr(base, "librccl") != base || l->n >= 4) return 0;
  for (int i = 0; i < l->n; ++i)
    if (!strcmp(l->paths[i], name)) return 0;
  snprintf(l->paths[l->n++], sizeof l->paths[0], "%s", name);
  return 0;
}

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    Loaded have;
    dl_iterate_phdr(note_rccl, &have);
    const char* forced = getenv("TASU_RCCL_PATH");
    if (forced && !*forced) forced = nullptr;
    // Two RCCL copies in one process each keep their own device state, topology cache and proxy threads: refuse instead of
    // binding a second one.  (a) more than one copy is already mapped; (b) TASU_RCCL_PATH names a file other than the mapped copy.
    if (have.n > 1) {
      snprintf(x.error, sizeof x.error, "%d RCCL copies are mapped into this process (%s, %s, ...): refusing to pick one", have.n,
               have.paths[0], have.paths[1]);
      return x;
    }
    char want[512] = {0};
    if (forced) {
      char* real = realpath(forced, nullptr);
      snprintf(want, sizeof want, "%s", real ? real : forced);
      free(real);
      if (have.n == 1) {
        char* mapped = realpath(have.paths[0], nullptr);
        const bool same = mapped && !strcmp(mapped, want);
        free(mapped);
        if (!same) {
          snprintf(x.error, sizeof x.error, "TASU_RCCL_PATH=%s but the process already maps %s: binding it would put a second RCCL "
                   "beside the host framework's", forced, have.paths[0]);
          return x;
        }
      }
    }
    if (have.n == 1) {                                   // the copy already in the process, by its exact path
      x.handle = dlopen(have.paths[0], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      snprintf(x.path, sizeof x.path, "%s", have.paths[0]);
    }
    if (!x.handle) {                                     // nothing mapped: the forced file, else the ROCm installation's
      const char* names[] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names) {
        if (!n) continue;
        x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (x.handle) {
          snprintf(x.path, sizeof x.path, "%s", n);
          break;
        }
        if (n == forced) {                               // a path given explicitly must load
          snprintf(x.error, sizeof x.error, "TASU_RCCL_PATH=%s: %s", forced, dlerror());
          return x;
        }
      }
    }
    if (!x.handle) {
      snprintf(x.error, sizeof x.error, "no librccl.so found (TASU_RCCL_PATH overrides the search)");
      return x;
    }
    x.get_unique_id = (GetUniqueIdFn)dlsym(x.handle, "ncclGetUniqueId");
    x.comm_init_rank = (CommInitRankFn)dlsym(x.handle, "ncclCommInitRank");
    x.comm_destroy = (CommDestroyFn)dlsym(x.handle, "ncclCommDestroy");
    x.all_reduce = (AllReduceFn)dlsym(x.handle, "ncclAllReduce");
    x.comm_count = (CommCountFn)dlsym(x.handle, "ncclCommCount");
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce && x.comm_count;
    if (!x.ok) snprintf(x.error, sizeof x.error, "%s lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / CommCount", x.path);
    return x;
  }();
  return r;
}

constexpr int NCCL_FLOAT32 = 7, NCCL_INT32 = 2, NCCL_SUM = 0, NCCL_MIN = 3;

}  // namespace

extern "C" int tasu_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int tasu_comm_library(char* out, int n) {
  if (!out || n <= 0) return TASU_ERR_ARG;
  const Rccl& r = rccl();
  snprintf(out, (size_t)n, "%s", r.ok ? r.path : r.error);
  return r.ok ? TASU_OK : TASU_ERR_LAUNCH;
}

extern "C" int tasu_comm_count(void* comm, int* count) {
  if (!comm || !count) return TASU_ERR_ARG;
  if (!rccl().ok) return TASU_ERR_LAUNCH;
  return rccl().comm_count((Comm)comm, count) == 0 ? TASU_OK : TASU_ERR_LAUNCH;
}

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
