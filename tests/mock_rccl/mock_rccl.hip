// TEST INFRASTRUCTURE (never loaded by the product unless MH_RCCL_LIB points at it): a stand-in for librccl that lets SEVERAL
// processes sharing ONE GPU run the library's native transport (marlin_amd/csrc/rccl_native.h) -- RCCL itself refuses two ranks on
// one device, and a gpurun lease has one device.  It implements exactly the entry points rccl_native.h resolves, with RCCL's
// signatures and semantics, over a POSIX shared-memory segment: a collective drains the caller's stream, stages through the
// segment (device -> host slot, barrier, host slots -> device, barrier) and returns.  What this exercises is everything on OUR
// side of the call at N = 2 / 4 / 8 -- buffer sizes, counts, staging, chunk order, the device all-gather of round polynomials,
// bench.py's transport set-up and self-tests -- not RCCL (AMD's code; executed for real at world 1 by
// tests/test_gpu_rccl_native.py).
// MH_MOCK_RCCL_ASYNC=1 (round 5): the collectives are ENQUEUED like RCCL's -- asynchronous copies through the (then
// page-locked) segment and the barriers as host functions in stream order; the call returns at once -- so that the library's
// stream-ordered use of its exchange buffers (sl_send / sl_recv / ntt_dist_buf reused while a collective is in flight) is
// exercised at N > 1.  -DMOCK_NO_ALLTOALL builds the variant WITHOUT ncclAllToAll (libmock_rccl_noa2a.so): rccl_native.h then
// takes its grouped ncclSend / ncclRecv path.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
struct Header { std::atomic<uint32_t> arrived; std::atomic<uint32_t> generation; std::atomic<uint32_t> attached; uint32_t nranks; };
struct ncclComm { int rank, nranks; size_t slot; char name[64]; Header* hdr; char* slots; size_t map_bytes; bool solo; void* filled[16]; size_t filled_bytes[16];
                  bool async; };
typedef ncclComm* ncclComm_t;

// per rank; the tests move a few MB, a 2^20 rehearsal 34 MB, a 2^22 one with 4 ranks 134 MB: MH_MOCK_RCCL_SLOT_MB (the same on
// every rank) raises it (the segment is sparse: only what is written is ever backed by memory)
static size_t slot_bytes() { const char* e = getenv("MH_MOCK_RCCL_SLOT_MB"); const long mb = e ? atol(e) : 96; return (size_t)(mb > 0 ? mb : 96) << 20; }

static bool async_mode() { const char* e = getenv("MH_MOCK_RCCL_ASYNC"); return e && atoi(e) == 1; }
static void barrier(ncclComm* c);
static void barrier_hostfn(void* p) { barrier((ncclComm*)p); }
// one collective in stream order: own chunk(s) into the own slot, barrier, the peers' chunks out of their slots, barrier
static ncclResult_t enqueue_exchange(ncclComm* c, hipStream_t s, const void* send, size_t send_bytes, void* recv, size_t count, bool alltoall) {
  if (hipMemcpyAsync(c->slots + (size_t)c->rank * c->slot, send, send_bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(s, barrier_hostfn, c) != hipSuccess) return ncclUnhandledCudaError;
  for (int p = 0; p < c->nranks; p++) {
    const char* src = c->slots + (size_t)p * c->slot + (alltoall ? (size_t)c->rank * count : 0);
    if (hipMemcpyAsync((char*)recv + (size_t)p * count, src, count, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
  }
  if (hipLaunchHostFunc(s, barrier_hostfn, c) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
static void barrier(ncclComm* c) {
  Header* h = c->hdr;
  const uint32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_release);
  } else {
    while (h->generation.load(std::memory_order_acquire) == gen) usleep(20);
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, sizeof(id->internal));
  snprintf(id->internal, sizeof(id->internal), "/mh_mock_rccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}
// MH_MOCK_RCCL_SOLO=1 (MEASUREMENT ONLY, bench.py --simulate-rank R/G --transport native): this process is rank R of G WITHOUT
// peers.  A collective is then a stream-ordered local copy of the caller's own chunk into its slot -- what the native transport
// costs the HOST of one rank, with no interpreter and no synchronisation, on the one GPU of a lease.  Small payloads (the partial
// points) are replicated into every slot; the foreign slots of large receive buffers are filled ONCE with pseudo-random field
// elements, so that what the prover feeds to its MSMs afterwards has realistic digits.  Results are meaningless; timings are not.
static bool solo_mode() { const char* e = getenv("MH_MOCK_RCCL_SOLO"); return e && atoi(e) == 1; }
static ncclResult_t solo_fill(ncclComm* c, void* recv, size_t total) {
  for (int i = 0; i < 16; i++) if (c->filled[i] == recv && c->filled_bytes[i] >= total) return ncclSuccess;
  uint64_t* h = (uint64_t*)malloc(total + 32);
  if (!h) return ncclSystemError;
  uint64_t x = 0x9e3779b97f4a7c15ull ^ (uint64_t)total;
  for (size_t i = 0; i < total / 8; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    h[i] = (i & 3) == 3 ? (x & ((1ull << 60) - 1)) : x;       // the top word of every 32-byte element below 2^60: a valid Fr
  }
  const hipError_t e = hipMemcpy(recv, h, total, hipMemcpyHostToDevice);
  free(h);
  if (e != hipSuccess) return ncclUnhandledCudaError;
  static int next = 0;
  c->filled[next & 15] = recv; c->filled_bytes[next & 15] = total; next++;
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank; c->nranks = nranks; c->slot = slot_bytes();
  if (solo_mode()) { c->solo = true; c->hdr = nullptr; memset(c->filled, 0, sizeof(c->filled)); *out = c; return ncclSuccess; }
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = 4096 + (size_t)nranks * c->slot;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->hdr = (Header*)p; c->slots = (char*)p + 4096;
  c->async = async_mode();
  if (c->async && hipHostRegister(c->slots, (size_t)nranks * c->slot, hipHostRegisterDefault) != hipSuccess) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
  // a fresh segment is zero-filled: the counters start at 0 without an initialisation race
  c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
  while (c->hdr->attached.load(std::memory_order_acquire) < (uint32_t)nranks) usleep(100);     // ncclCommInitRank is collective
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->solo) { delete c; return ncclSuccess; }
  if (c->async) { (void)hipDeviceSynchronize(); (void)hipHostUnregister(c->slots); }
  munmap((void*)c->hdr, c->map_bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }
ncclResult_t ncclGetVersion(int* v) { if (v) *v = 0; return ncclSuccess; }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
  if (c->solo) {
    if (count <= (64u << 10)) {                 // partial points: the own chunk in every slot
      for (int p = 0; p < c->nranks; p++)
        if (hipMemcpyAsync((char*)recv + (size_t)p * count, send, count, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
      return ncclSuccess;
    }
    if (solo_fill(c, recv, count * c->nranks) != ncclSuccess) return ncclUnhandledCudaError;
    return hipMemcpyAsync((char*)recv + (size_t)c->rank * count, send, count, hipMemcpyDeviceToDevice, s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
  }
  if (count > c->slot) return ncclInvalidArgument;
  if (c->async && !getenv("MH_MOCK_RCCL_CORRUPT_RANK")) return enqueue_exchange(c, s, send, count, recv, count, false);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * c->slot, send, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  for (int p = 0; p < c->nranks; p++)
    if (hipMemcpy((char*)recv + (size_t)p * count, c->slots + (size_t)p * c->slot, count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);                        // nobody overwrites a slot before every rank has read it
  // MH_MOCK_RCCL_CORRUPT_RANK=<r>: rank r receives a damaged byte -- lets a test see a transport self-test FAIL on one rank
  // and the caller fall back on all of them
  if (const char* e = getenv("MH_MOCK_RCCL_CORRUPT_RANK"))
    if (atoi(e) == c->rank && (hipMemsetAsync(recv, 0xA5, 1, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))   // on the caller's
      return ncclUnhandledCudaError;       // stream: a hipMemset of device memory may return before it lands, and the caller's copy-back is on `s`
  return ncclSuccess;
}
#ifndef MOCK_NO_ALLTOALL
ncclResult_t ncclAllToAll(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
  if (c->solo) {
    if (solo_fill(c, recv, count * c->nranks) != ncclSuccess) return ncclUnhandledCudaError;
    const size_t off = (size_t)c->rank * count;
    return hipMemcpyAsync((char*)recv + off, (const char*)send + off, count, hipMemcpyDeviceToDevice, s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
  }
  if (count * (size_t)c->nranks > c->slot) return ncclInvalidArgument;
  if (c->async) return enqueue_exchange(c, s, send, count * c->nranks, recv, count, true);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * c->slot, send, count * c->nranks, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  for (int p = 0; p < c->nranks; p++)         // chunk `rank` of peer p's send buffer is what p sends to this rank
    if (hipMemcpy((char*)recv + (size_t)p * count, c->slots + (size_t)p * c->slot + (size_t)c->rank * count, count, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}
#endif
// grouped point-to-point (what rccl_native.h issues when the library has no ncclAllToAll): the operations between ncclGroupStart
// and ncclGroupEnd are collected and carried out together -- every send staged into the own slot at the peer's chunk position,
// barrier, every receive taken from the peer's slot at the own chunk position, barrier (stream-ordered when MH_MOCK_RCCL_ASYNC=1)
struct P2P { const void* send; void* recv; size_t count; int peer; ncclComm* c; hipStream_t s; };
static thread_local P2P g_ops[64]; static thread_local int g_nops = 0; static thread_local bool g_in_group = false;
static ncclResult_t run_group() {
  if (g_nops == 0) return ncclSuccess;
  ncclComm* c = g_ops[0].c; hipStream_t s = g_ops[0].s;
  if (c->solo) {
    for (int i = 0; i < g_nops; i++)
      if (g_ops[i].recv && g_ops[i].peer == c->rank)
        for (int j = 0; j < g_nops; j++)
          if (g_ops[j].send && g_ops[j].peer == c->rank && hipMemcpyAsync(g_ops[i].recv, g_ops[j].send, g_ops[i].count, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
  }
  if (!c->async && hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  for (int i = 0; i < g_nops; i++)
    if (g_ops[i].send) {
      char* dst = c->slots + (size_t)c->rank * c->slot + (size_t)g_ops[i].peer * g_ops[i].count;
      if ((size_t)(g_ops[i].peer + 1) * g_ops[i].count > c->slot) return ncclInvalidArgument;
      const hipError_t e = c->async ? hipMemcpyAsync(dst, g_ops[i].send, g_ops[i].count, hipMemcpyDeviceToHost, s) : hipMemcpy(dst, g_ops[i].send, g_ops[i].count, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return ncclUnhandledCudaError;
    }
  if (c->async) { if (hipLaunchHostFunc(s, barrier_hostfn, c) != hipSuccess) return ncclUnhandledCudaError; } else barrier(c);
  for (int i = 0; i < g_nops; i++)
    if (g_ops[i].recv) {
      const char* src = c->slots + (size_t)g_ops[i].peer * c->slot + (size_t)c->rank * g_ops[i].count;
      const hipError_t e = c->async ? hipMemcpyAsync(g_ops[i].recv, src, g_ops[i].count, hipMemcpyHostToDevice, s) : hipMemcpy(g_ops[i].recv, src, g_ops[i].count, hipMemcpyHostToDevice);
      if (e != hipSuccess) return ncclUnhandledCudaError;
    }
  if (c->async) { if (hipLaunchHostFunc(s, barrier_hostfn, c) != hipSuccess) return ncclUnhandledCudaError; } else barrier(c);
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
  if (!g_in_group || g_nops >= 64) return ncclInvalidArgument;          // this stand-in only knows grouped, symmetric exchanges
  g_ops[g_nops++] = P2P{send, nullptr, count, peer, c, s};
  return ncclSuccess;
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
  if (!g_in_group || g_nops >= 64) return ncclInvalidArgument;
  g_ops[g_nops++] = P2P{nullptr, recv, count, peer, c, s};
  return ncclSuccess;
}
ncclResult_t ncclGroupStart(void) { g_in_group = true; g_nops = 0; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { g_in_group = false; const ncclResult_t r = run_group(); g_nops = 0; return r; }
}
