// Gradient all-reduce for callers that have no torch.distributed: a thin C-ABI wrapper over RCCL (the reference sums
// its clones' gradients with tf.add_n inside ONE process, deployment/model_deploy.py:473-503; with one process per GPU
// that sum is an all-reduce over xGMI).  RCCL is bound lazily with dlopen / dlsym -- the library itself links nothing
// but the HIP runtime, and a process that never calls these entry points never loads RCCL.  TG_RCCL_PATH names the
// library to load (default: librccl.so.1, then librccl.so, by the loader's search path).
#include "tg_common.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool rccl_bind(Rccl& r) {
  const char* env = getenv("TG_RCCL_PATH");
  const char* names[] = {env, "librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) return false;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
    dlclose(r.handle);
    r.handle = nullptr;
    return false;
  }
  return true;
}

// bound once, by whichever thread asks first: a function-local static's initialiser is thread-safe in C++11
Rccl* rccl() {
  static Rccl r;
  static const bool ok = rccl_bind(r);
  return ok ? &r : nullptr;
}

int fail(const char* who, Rccl* r, ncclResult_t rc) {
  tg_set_error("%s: RCCL error %d (%s)", who, (int)rc, (r && r->GetErrorString) ? r->GetErrorString(rc) : "?");
  return TG_ECOMM;
}

}  // namespace

extern "C" {

int tg_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int tg_comm_unique_id(void* id) {
  TG_CHECK(id, TG_EINVAL, "tg_comm_unique_id: null pointer");
  Rccl* r = rccl();
  TG_CHECK(r, TG_ENOSUP, "tg_comm_unique_id: RCCL could not be loaded (TG_RCCL_PATH, librccl.so.1, librccl.so)");
  ncclResult_t rc = r->GetUniqueId((ncclUniqueId*)id);
  return rc == ncclSuccess ? TG_OK : fail("tg_comm_unique_id", r, rc);
}

int tg_comm_init(const void* id, int nranks, int rank, void** comm) {
  TG_CHECK(id && comm && nranks > 0 && rank >= 0 && rank < nranks, TG_EINVAL, "tg_comm_init: bad arguments");
  Rccl* r = rccl();
  TG_CHECK(r, TG_ENOSUP, "tg_comm_init: RCCL could not be loaded (TG_RCCL_PATH, librccl.so.1, librccl.so)");
  ncclUniqueId uid;
  ::memcpy((void*)&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  ncclResult_t rc = r->CommInitRank(&c, nranks, uid, rank);      // the calling thread's current HIP device
  if (rc != ncclSuccess) return fail("tg_comm_init", r, rc);
  *comm = (void*)c;
  return TG_OK;
}

int tg_allreduce(void* comm, void* buf, int64_t count, int dtype, void* stream) {
  TG_CHECK(comm && buf && count > 0, TG_EINVAL, "tg_allreduce: bad arguments");
  TG_CHECK(dtype == TG_F32 || dtype == TG_BF16 || dtype == TG_F16, TG_EINVAL, "tg_allreduce: dtype %d", dtype);
  const ncclDataType_t nt = dtype == TG_F32 ? ncclFloat32 : (dtype == TG_BF16 ? ncclBfloat16 : ncclFloat16);
  Rccl* r = rccl();
  TG_CHECK(r, TG_ENOSUP, "tg_allreduce: RCCL is not loaded");
  ncclResult_t rc = r->AllReduce(buf, buf, (size_t)count, nt, ncclSum,
                                 (ncclComm_t)comm, (hipStream_t)stream);
  return rc == ncclSuccess ? TG_OK : fail("tg_allreduce", r, rc);
}

int tg_comm_destroy(void* comm) {
  if (!comm) return TG_OK;
  Rccl* r = rccl();
  TG_CHECK(r, TG_ENOSUP, "tg_comm_destroy: RCCL is not loaded");
  ncclResult_t rc = r->CommDestroy((ncclComm_t)comm);
  return rc == ncclSuccess ? TG_OK : fail("tg_comm_destroy", r, rc);
}

}  // extern "C"
