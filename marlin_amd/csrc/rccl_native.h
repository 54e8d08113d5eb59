// RCCL called from the library itself, on the library's own stream -- the native transport of the multi-GPU prover.
//
// The exchanges of a sharded proof (one all-gather of partial points per commit round, one all-to-all per
// distributed transform, two all-gathers of round polynomials) were first wired through a caller-supplied callback
// (marlin_amd/dist.py: C -> Python CFUNCTYPE -> torch.distributed -> RCCL -> C).  Here the same three collectives are issued
// from C++ as ncclAllGather / ncclAllToAll on `Context::stream`: they are ordered with the kernels around them on the
// device, nothing synchronises the host except where the host needs the bytes (the partial points, for Fiat-Shamir), and no
// interpreter sits in the path.  The reference has no counterpart (a single-process CPU prover); what is exchanged follows
// /root/reference src/ahp/prover.rs:351-366,532-535,655-688 (the transforms that are distributed) and src/lib.rs:172,193,213
// (the commitments whose MSMs are sharded).
//
// librccl is resolved at run time (dlopen by SONAME: the instance torch already mapped when the caller is a torch process,
// /opt/rocm/lib's otherwise), so libmarlin_hip.so keeps loading on a box without RCCL and a one-GPU caller never touches it.
// The communicator is the library's own: rank 0 draws an ncclUniqueId (mh_rccl_unique_id), the caller hands the 128 bytes to
// every rank by whatever means it has (bench.py: one torch.distributed broadcast; a C caller: its own bootstrap), and every
// rank calls mh_marlin_set_rccl(rank, world, id).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: every function is reached through dlsym
#include <mutex>
#include <string>
#include "context.h"

namespace rcclnative {
using mh::Context;
using mh::fail;

struct Api {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;          // RCCL extension; grouped send / recv when absent
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string path;
};

// the function table is the process's; a communicator (and its staging) belongs to a context (prover.hip: ProverState::rccl)
inline Api& api() { static Api a; return a; }
struct State {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  mh::Scratch d_send, d_recv;                 // device staging of the host all-gather (partial points: <= a few KB)
  mh::Pinned h_send, h_recv;
  uint64_t n_allgather_host = 0, n_alltoall = 0, n_allgather_dev = 0, bytes_moved = 0;
};

inline int load_api() {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  Api& a = api();
  if (a.lib) return MH_OK;
  // the copy that is already mapped (a torch process) first, then the loader's search path, then the ROCm tree.
  // MH_RCCL_LIB=<path>: that library and no other -- the hook the tests use to run this transport at N > 1 on ONE GPU over a
  // shared-memory stand-in (tests/mock_rccl/), since RCCL itself refuses two ranks on one device
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  const char* forced = getenv("MH_RCCL_LIB");
  void* h = nullptr;
  if (forced && *forced) {
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
  } else {
    h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) {
    const char* why = dlerror();               // read ONCE: the call clears the message, a second one returns NULL
    return fail(MH_EINVAL, std::string("native RCCL transport: ") + (forced && *forced ? forced : "librccl.so.1") + " not found (" + (why ? why : "?") + ")");
  }
#define RCCL_SYM(field, name, required)                                                                         \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                                \
  if (required && !a.field) { dlclose(h); return fail(MH_EINVAL, std::string("native RCCL transport: symbol missing: ") + name); }
  RCCL_SYM(GetUniqueId, "ncclGetUniqueId", true)
  RCCL_SYM(CommInitRank, "ncclCommInitRank", true)
  RCCL_SYM(CommDestroy, "ncclCommDestroy", true)
  RCCL_SYM(CommAbort, "ncclCommAbort", false)
  RCCL_SYM(AllGather, "ncclAllGather", true)
  RCCL_SYM(AllToAll, "ncclAllToAll", false)
  RCCL_SYM(Send, "ncclSend", true)
  RCCL_SYM(Recv, "ncclRecv", true)
  RCCL_SYM(GroupStart, "ncclGroupStart", true)
  RCCL_SYM(GroupEnd, "ncclGroupEnd", true)
  RCCL_SYM(GetErrorString, "ncclGetErrorString", true)
  RCCL_SYM(GetVersion, "ncclGetVersion", false)
#undef RCCL_SYM
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(a.AllGather), &info) && info.dli_fname) a.path = info.dli_fname;
  a.lib = h;
  return MH_OK;
}

#define MH_RCCL(call)                                                                                         \
  do {                                                                                                        \
    ncclResult_t _r = (call);                                                                                 \
    if (_r != ncclSuccess) {                                                                                  \
      char _b[384];                                                                                           \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, api().GetErrorString(_r), __FILE__, __LINE__); \
      return mh::fail(MH_EHIP, _b);                                                                           \
    }                                                                                                         \
  } while (0)

inline int unique_id(uint8_t* out128) {
  MH_TRY(load_api());
  ncclUniqueId id;
  MH_RCCL(api().GetUniqueId(&id));
  static_assert(sizeof(id) == NCCL_UNIQUE_ID_BYTES && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId is 128 opaque bytes");
  memcpy(out128, id.internal, sizeof(id));
  return MH_OK;
}

inline int destroy(State& s) {
  if (s.comm) {
    (void)api().CommDestroy(s.comm);
    s.comm = nullptr;
  }
  s.rank = 0; s.world = 1;
  s.d_send.release(); s.d_recv.release(); s.h_send.release(); s.h_recv.release();
  return MH_OK;
}

// collective over all ranks: returns when this rank's communicator exists
inline int init(Context& c, State& s, int rank, int world, const uint8_t* id128) {
  MH_TRY(load_api());
  if (s.comm) destroy(s);
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id));
  MH_HIP(hipSetDevice(c.device));
  MH_RCCL(api().CommInitRank(&s.comm, world, id, rank));
  s.rank = rank; s.world = world;
  s.d_send.hookable = s.d_recv.hookable = false;          // (mh_debug_fail_scratch never strikes the transport's own staging)
  s.n_allgather_host = s.n_alltoall = s.n_allgather_dev = s.bytes_moved = 0;
  return MH_OK;
}

// mh_allgather_fn: `bytes` bytes of HOST memory from every rank into recv (rank-major).  The host needs the result (the
// partial points feed the transcript), so this is the one exchange that ends in a stream synchronisation.
inline int allgather_host(const void* send, size_t bytes, void* recv, void* user) {
  Context& c = mh::ctx();
  State& s = *static_cast<State*>(user);
  if (!s.comm) return fail(MH_EINVAL, "native RCCL transport: no communicator (mh_marlin_set_rccl)");
  MH_TRY(s.d_send.ensure(bytes)); MH_TRY(s.d_recv.ensure(bytes * s.world));
  MH_TRY(s.h_send.ensure(bytes)); MH_TRY(s.h_recv.ensure(bytes * s.world));
  memcpy(s.h_send.ptr, send, bytes);
  MH_HIP(hipMemcpyAsync(s.d_send.ptr, s.h_send.ptr, bytes, hipMemcpyHostToDevice, c.stream));
  MH_RCCL(api().AllGather(s.d_send.ptr, s.d_recv.ptr, bytes, ncclUint8, s.comm, c.stream));
  MH_HIP(hipMemcpyAsync(s.h_recv.ptr, s.d_recv.ptr, bytes * s.world, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  memcpy(recv, s.h_recv.ptr, bytes * s.world);
  s.n_allgather_host++; s.bytes_moved += bytes * (s.world - 1);
  return MH_OK;
}

// mh_alltoall_fn on DEVICE buffers of the library, stream-ordered: chunk q of d_send goes to rank q, chunk q of d_recv comes
// from rank q.  Returns as soon as the collective is enqueued.
inline int alltoall_dev(const void* d_send, size_t bytes_per_peer, void* d_recv, void* user) {
  Context& c = mh::ctx();
  State& s = *static_cast<State*>(user);
  if (!s.comm) return fail(MH_EINVAL, "native RCCL transport: no communicator (mh_marlin_set_rccl)");
  if (api().AllToAll) {
    MH_RCCL(api().AllToAll(d_send, d_recv, bytes_per_peer, ncclUint8, s.comm, c.stream));
  } else {
    MH_RCCL(api().GroupStart());
    for (int q = 0; q < s.world; q++) {
      MH_RCCL(api().Send((const char*)d_send + (size_t)q * bytes_per_peer, bytes_per_peer, ncclUint8, q, s.comm, c.stream));
      MH_RCCL(api().Recv((char*)d_recv + (size_t)q * bytes_per_peer, bytes_per_peer, ncclUint8, q, s.comm, c.stream));
    }
    MH_RCCL(api().GroupEnd());
  }
  s.n_alltoall++; s.bytes_moved += bytes_per_peer * (s.world - 1);
  return MH_OK;
}

// all-gather of DEVICE buffers, stream-ordered (the round polynomials of the sliced sections: the callback transport has
// to express this as an all-to-all of `world` copies of the same chunk)
inline int allgather_dev(const void* d_send, size_t bytes, void* d_recv, void* user) {
  Context& c = mh::ctx();
  State& s = *static_cast<State*>(user);
  if (!s.comm) return fail(MH_EINVAL, "native RCCL transport: no communicator (mh_marlin_set_rccl)");
  MH_RCCL(api().AllGather(d_send, d_recv, bytes, ncclUint8, s.comm, c.stream));
  s.n_allgather_dev++; s.bytes_moved += bytes * (s.world - 1);
  return MH_OK;
}

}  // namespace rcclnative
