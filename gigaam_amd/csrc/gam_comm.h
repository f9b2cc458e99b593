// gam_comm.h -- the multi-GPU exchange of the hot path behind the C ABI: one grouped ncclAllGather of the
// decode buffers.  RCCL is resolved at run time (dlopen by soname), so the library has no
// link-time dependency on it and, inside a torch process, shares the RCCL instance torch already loaded.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/gigaam_hip.h"

namespace gam_rccl {
// the handful of RCCL declarations used (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclAllGather :678)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int Result;          // ncclSuccess = 0
constexpr int kInt32 = 2;    // ncclInt32
struct Api {
  void* lib = nullptr;
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
  Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  Result (*GroupStart)() = nullptr;
  Result (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(Result) = nullptr;
  std::string err;
};

static inline void load(Api& a);
static inline Api* api() {   // resolved once per process (thread-safe)
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] { load(a); });
  return &a;
}
static inline void load(Api& a) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {      // first the instance the process already holds (inside torch: torch's own RCCL), whatever file it came from
    a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (a.lib) break;
  }
  if (!a.lib)
    for (const char* n : names) {
      a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
  if (!a.lib) { a.err = std::string("cannot load librccl: ") + dlerror(); return; }
#define GAM_RCCL_SYM(field, name)                                         \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));     \
  if (!a.field) { a.err = std::string("librccl lacks ") + name; a.lib = nullptr; return; }
  GAM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  GAM_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  GAM_RCCL_SYM(CommDestroy, "ncclCommDestroy");
  GAM_RCCL_SYM(AllGather, "ncclAllGather");
  GAM_RCCL_SYM(GroupStart, "ncclGroupStart");
  GAM_RCCL_SYM(GroupEnd, "ncclGroupEnd");
  GAM_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef GAM_RCCL_SYM
}
}  // namespace gam_rccl

struct gam_comm {
  gam_rccl::Comm comm = nullptr;
  int rank = 0, world = 1, device = 0;
  std::string err;
};

static thread_local std::string gam_comm_static_err;

extern "C" {

int gam_comm_unique_id(char id_out[GAM_COMM_ID_BYTES]) {
  gam_rccl::Api* a = gam_rccl::api();
  if (!a->lib) { gam_comm_static_err = a->err; return -2; }
  gam_rccl::UniqueId id;
  const int r = a->GetUniqueId(&id);
  if (r != 0) { gam_comm_static_err = std::string("ncclGetUniqueId: ") + a->GetErrorString(r); return -2; }
  static_assert(sizeof id.internal == GAM_COMM_ID_BYTES, "RCCL unique id size");
  memcpy(id_out, id.internal, GAM_COMM_ID_BYTES);
  return 0;
}

int gam_comm_create(const char id[GAM_COMM_ID_BYTES], int rank, int world, int device_id, gam_comm** out) {
  if (!id || !out || world <= 0 || rank < 0 || rank >= world) return -1;
  gam_comm* c = new gam_comm();
  c->rank = rank; c->world = world; c->device = device_id;
  *out = c;
  gam_rccl::Api* a = gam_rccl::api();
  if (!a->lib) { c->err = a->err; return -2; }
  if (hipSetDevice(device_id) != hipSuccess) { c->err = "hipSetDevice failed"; return -2; }
  gam_rccl::UniqueId uid;
  memcpy(uid.internal, id, GAM_COMM_ID_BYTES);
  const int r = a->CommInitRank(&c->comm, world, uid, rank);
  if (r != 0) { c->err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); c->comm = nullptr; return -2; }
  return 0;
}

int gam_comm_world(const gam_comm* c) { return c ? c->world : -1; }

const char* gam_comm_last_error(const gam_comm* c) { return c ? c->err.c_str() : gam_comm_static_err.c_str(); }

int gam_gather_ids(gam_comm* c, const int32_t* index, const int32_t* counts, const int32_t* ids, const int32_t* frames,
                   int rows, int cap, int32_t* all_index, int32_t* all_counts, int32_t* all_ids, int32_t* all_frames,
                   void* stream) {
  if (!c) return -1;
  if (!c->comm) { c->err = "communicator was not created"; return -1; }
  if (rows <= 0 || cap <= 0 || !counts || !ids || !frames || !all_counts || !all_ids || !all_frames || (index == nullptr) != (all_index == nullptr)) {
    c->err = "bad gam_gather_ids arguments";
    return -1;
  }
  gam_rccl::Api* a = gam_rccl::api();
  hipStream_t s = (hipStream_t)stream;
  if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return -2; }
  int r = a->GroupStart();
  if (r == 0 && index) r = a->AllGather(index, all_index, (size_t)rows, gam_rccl::kInt32, c->comm, s);
  if (r == 0) r = a->AllGather(counts, all_counts, (size_t)rows, gam_rccl::kInt32, c->comm, s);
  if (r == 0) r = a->AllGather(ids, all_ids, (size_t)rows * cap, gam_rccl::kInt32, c->comm, s);
  if (r == 0) r = a->AllGather(frames, all_frames, (size_t)rows * cap, gam_rccl::kInt32, c->comm, s);
  const int r2 = a->GroupEnd();
  if (r == 0) r = r2;
  if (r != 0) { c->err = std::string("ncclAllGather: ") + a->GetErrorString(r); return -2; }
  return 0;
}

void gam_comm_destroy(gam_comm* c) {
  if (!c) return;
  if (c->comm) {
    gam_rccl::Api* a = gam_rccl::api();
    if (a->lib) a->CommDestroy(c->comm);
  }
  delete c;
}

}  // extern "C"
