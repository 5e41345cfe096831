// comm.hip — the halo exchange of a node-range partitioned graph on RCCL, behind the C ABI
// (include/stag_hip.h: stag_comm_* / stag_halo_allgather / stag_halo_exchange).
//
// The reference has no multi-GPU code; BASELINE.json's north_star adds "RCCL all-gather of halo features
// over xGMI".  RCCL is bound at run time (dlopen), preferring the copy already mapped into the process
// (PyTorch ships one), so libstag_hip.so carries no link-time dependency on it and still loads on a
// machine without RCCL — the entry points then return STAG_ENOSYS.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/stag_hip.h"

namespace {

// the few declarations of rccl.h this file needs (the header is not required to build)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {                       // the copy already in the process, if any
      x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (x.handle) break;
    }
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (size_t i = 0; !x.handle && i < sizeof(paths) / sizeof(paths[0]); ++i) x.handle = dlopen(paths[i], RTLD_NOW);
    if (!x.handle) return x;
#define STAG_SYM(field, name) *(void**)(&x.field) = dlsym(x.handle, name)
    STAG_SYM(GetUniqueId, "ncclGetUniqueId");
    STAG_SYM(CommInitRank, "ncclCommInitRank");
    STAG_SYM(CommDestroy, "ncclCommDestroy");
    STAG_SYM(AllGather, "ncclAllGather");
    STAG_SYM(Send, "ncclSend");
    STAG_SYM(Recv, "ncclRecv");
    STAG_SYM(GroupStart, "ncclGroupStart");
    STAG_SYM(GroupEnd, "ncclGroupEnd");
#undef STAG_SYM
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.Send && x.Recv && x.GroupStart &&
           x.GroupEnd;
    return x;
  }();
  return r;
}

struct StagComm {
  ncclComm_t comm;
  int rank, world;
};

}  // namespace

extern "C" {

int stag_comm_unique_id(void* id_out_host) {
  if (!id_out_host) return STAG_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return STAG_ENOSYS;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return STAG_EIO;
  memcpy(id_out_host, &id, sizeof(id));
  return STAG_OK;
}

int stag_comm_init(const void* id_host, int32_t rank, int32_t world, void** comm_out) {
  if (!id_host || !comm_out || world < 1 || rank < 0 || rank >= world) return STAG_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return STAG_ENOSYS;
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  if (r.CommInitRank(&c, world, id, rank) != ncclSuccess) return STAG_EIO;
  *comm_out = new StagComm{c, rank, world};
  return STAG_OK;
}

int stag_comm_destroy(void* comm) {
  if (!comm) return STAG_EINVAL;
  StagComm* c = static_cast<StagComm*>(comm);
  const int rc = rccl().CommDestroy(c->comm);
  delete c;
  return rc == ncclSuccess ? STAG_OK : STAG_EIO;
}

int stag_halo_allgather(void* comm, const float* x_local, int64_t n_floats, float* x_full, void* stream) {
  if (!comm || n_floats < 0 || (n_floats > 0 && (!x_local || !x_full))) return STAG_EINVAL;
  StagComm* c = static_cast<StagComm*>(comm);
  if (n_floats == 0) return STAG_OK;
  return rccl().AllGather(x_local, x_full, (size_t)n_floats, ncclFloat32, c->comm, (hipStream_t)stream) == ncclSuccess
             ? STAG_OK : STAG_EIO;
}

int stag_halo_exchange(void* comm, const float* send, const int64_t* send_counts_host, float* recv,
                       const int64_t* recv_counts_host, void* stream) {
  if (!comm || !send_counts_host || !recv_counts_host) return STAG_EINVAL;
  StagComm* c = static_cast<StagComm*>(comm);
  Rccl& r = rccl();
  int64_t so = 0, ro = 0;
  for (int p = 0; p < c->world; ++p) {
    if (send_counts_host[p] < 0 || recv_counts_host[p] < 0) return STAG_EINVAL;
    if (p == c->rank && (send_counts_host[p] || recv_counts_host[p])) return STAG_EINVAL;   // own rows never travel
    so += send_counts_host[p]; ro += recv_counts_host[p];
  }
  if ((so > 0 && !send) || (ro > 0 && !recv)) return STAG_EINVAL;
  // one group: every pair's send and receive progress together, all xGMI links at once
  if (r.GroupStart() != ncclSuccess) return STAG_EIO;
  so = ro = 0;
  int bad = 0;
  for (int p = 0; p < c->world; ++p) {
    if (send_counts_host[p] > 0)
      bad |= r.Send(send + so, (size_t)send_counts_host[p], ncclFloat32, p, c->comm, (hipStream_t)stream) != ncclSuccess;
    if (recv_counts_host[p] > 0)
      bad |= r.Recv(recv + ro, (size_t)recv_counts_host[p], ncclFloat32, p, c->comm, (hipStream_t)stream) != ncclSuccess;
    so += send_counts_host[p]; ro += recv_counts_host[p];
  }
  if (r.GroupEnd() != ncclSuccess || bad) return STAG_EIO;
  return STAG_OK;
}

int stag_halo_exchange_multi(void* comm, int32_t n_tables, const float* const* send, float* const* recv,
                             const int32_t* widths, const int64_t* send_rows_host,
                             const int64_t* recv_rows_host, void* stream) {
  if (!comm || n_tables < 1 || !send || !recv || !widths || !send_rows_host || !recv_rows_host) return STAG_EINVAL;
  StagComm* c = static_cast<StagComm*>(comm);
  Rccl& r = rccl();
  int64_t so = 0, ro = 0;
  for (int p = 0; p < c->world; ++p) {
    if (send_rows_host[p] < 0 || recv_rows_host[p] < 0) return STAG_EINVAL;
    if (p == c->rank && (send_rows_host[p] || recv_rows_host[p])) return STAG_EINVAL;
    so += send_rows_host[p]; ro += recv_rows_host[p];
  }
  for (int t = 0; t < n_tables; ++t) {
    if (widths[t] <= 0) return STAG_EINVAL;
    if ((so > 0 && !send[t]) || (ro > 0 && !recv[t])) return STAG_EINVAL;
  }
  if (r.GroupStart() != ncclSuccess) return STAG_EIO;
  int bad = 0;
  for (int t = 0; t < n_tables; ++t) {
    const int64_t w = widths[t];
    so = ro = 0;
    for (int p = 0; p < c->world; ++p) {
      if (send_rows_host[p] > 0)
        bad |= r.Send(send[t] + so * w, (size_t)(send_rows_host[p] * w), ncclFloat32, p, c->comm, (hipStream_t)stream) != ncclSuccess;
      if (recv_rows_host[p] > 0)
        bad |= r.Recv(recv[t] + ro * w, (size_t)(recv_rows_host[p] * w), ncclFloat32, p, c->comm, (hipStream_t)stream) != ncclSuccess;
      so += send_rows_host[p]; ro += recv_rows_host[p];
    }
  }
  if (r.GroupEnd() != ncclSuccess || bad) return STAG_EIO;
  return STAG_OK;
}

}  // extern "C"
