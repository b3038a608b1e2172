// SPDX-License-Identifier: GPL-3.0-or-later
// fake_rccl.cpp -- TEST-ONLY stand-in for the RCCL entry points libmmoore_hip.so calls (csrc/mm_multi.hip):
//
//    ncclGetUniqueId  ncclCommInitRank  ncclCommInitAll  ncclAllGather  ncclGroupStart  ncclGroupEnd
//    ncclCommGetAsyncError  ncclCommDestroy  ncclGetErrorString
//
// Why: RCCL refuses two ranks on one device, and the development / test boxes have ONE GPU -- the library's gather
// (the replacement of the reference's dispatcher + merge, src/core/search_engine.cpp:66-188, :193-197) could never be
// executed with nranks > 1 there.  LD_PRELOADed into a TEST process (tests/test_gpu_multi.py, `bench.py --gpus N
// --allow-shared-device`), this file lets N processes -- or N contexts of one process -- on the SAME GPU form a
// communicator whose all-gather keeps the semantics the library relies on:
//
//   * asynchronous and stream-ordered: ncclAllGather returns at once; the send buffer is read, and the receive buffer
//     written, when the caller's stream gets there (so a product bug that overwrites a send buffer too early, or reads a
//     table too early, shows exactly as it would with the real library);
//   * a collective ends only when every rank has joined it; ranks may be any number of collectives apart (a ring of
//     kRing staging slots, each reused only after every rank has consumed it).
//
// How: a POSIX shared-memory segment named by the unique id, pinned with hipHostRegister in every process.  One
// all-gather of sequence number s on rank r =
//       [wait until every rank has consumed s - kRing]   hipStreamWaitValue64 on consumed[*]
//       device send buffer -> segment slot (s % kRing, r) hipMemcpyAsync D2H
//       posted[r] = s + 1                                 hipStreamWriteValue64
//       [wait until posted[q] >= s + 1 for every q]       hipStreamWaitValue64
//       segment slot (s % kRing, 0 .. n-1) -> recv buffer hipMemcpyAsync H2D (the slots of one sequence are contiguous)
//       consumed[r] = s + 1                               hipStreamWriteValue64
// -- stream memory operations only: no kernels, no host callbacks, nothing blocks the calling thread.
// Communicators of ONE process (ncclCommInitAll) take another route: the high-priority streams of two contexts on one
// device share a hardware queue, where a wait-value packet of rank 0 keeps rank 1's post from ever running (seen as a
// hang).  Their all-gathers are held until ncclGroupEnd (the library issues them inside a group), where every rank's
// call is known, and are ordered with HIP events: send buffer -> staging slot + event per rank, then every rank's stream
// waits for all events and fills its receive buffer -- equally asynchronous, equally stream-ordered.
// The HIP runtime is looked up at run time in the copy the process has already loaded (a torch wheel ships its own
// under the same SONAME): this file links against neither HIP nor RCCL.
//
// NEVER part of the product: nothing under monkey-moore_amd/ refers to it, bench.py refuses ranks that share a device
// unless --allow-shared-device is given, and the library's own RCCL calls are untouched.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace {

constexpr int kRing = 4;                           // staging slots per rank: collectives a rank may run ahead of the slowest
constexpr uint64_t kSlotBytes = 2ull << 20;        // staging per rank and round: longer sends go in pieces
constexpr int kMaxRanks = 64;
constexpr uint64_t kMagic = 0x66616b6572636331ull; // "fakercc1"

struct Header {
   std::atomic<uint64_t> magic;
   std::atomic<uint32_t> joined, left;
   uint32_t nranks, pad;
   uint64_t reserved[12];
   alignas(128) volatile uint64_t posted[kMaxRanks];
   alignas(128) volatile uint64_t consumed[kMaxRanks];
};
static_assert(sizeof(Header) <= 4096, "header page");

struct Segment {
   char name[64];
   Header *h = nullptr;
   uint8_t *data = nullptr;                        // [kRing][nranks][kSlotBytes]
   size_t bytes = 0;
   int nranks = 0;
   int users = 0;                                  // communicators of this process on it
};

struct Hip {
   void *lib = nullptr;
   hipError_t (*GetDevice)(int *) = nullptr;
   hipError_t (*SetDevice)(int) = nullptr;
   hipError_t (*HostRegister)(void *, size_t, unsigned int) = nullptr;
   hipError_t (*HostUnregister)(void *) = nullptr;
   hipError_t (*MemcpyAsync)(void *, const void *, size_t, hipMemcpyKind, hipStream_t) = nullptr;
   hipError_t (*StreamWriteValue64)(hipStream_t, void *, uint64_t, unsigned int) = nullptr;
   hipError_t (*StreamWaitValue64)(hipStream_t, void *, uint64_t, unsigned int, uint64_t) = nullptr;
   hipError_t (*DeviceGetAttribute)(int *, hipDeviceAttribute_t, int) = nullptr;
   hipError_t (*EventCreateWithFlags)(hipEvent_t *, unsigned int) = nullptr;
   hipError_t (*EventDestroy)(hipEvent_t) = nullptr;
   hipError_t (*EventRecord)(hipEvent_t, hipStream_t) = nullptr;
   hipError_t (*StreamWaitEvent)(hipStream_t, hipEvent_t, unsigned int) = nullptr;
   const char *(*GetErrorString)(hipError_t) = nullptr;
};

Hip g_hip;
std::once_flag g_hip_once;
char g_error[256] = "";

template <class F> bool sym(F &f, const char *name)
{
   f = reinterpret_cast<F>(dlsym(g_hip.lib, name));
   if (!f) {
      snprintf(g_error, sizeof g_error, "fake_rccl: %s not found in the HIP runtime", name);
   }
   return f != nullptr;
}

bool hip_runtime()
{
   std::call_once(g_hip_once, [] {
      // the copy this process already runs on (whoever calls an RCCL entry point has initialised HIP), by SONAME
      for (const char *name : {"libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"}) {
         g_hip.lib = dlopen(name, RTLD_NOLOAD | RTLD_LAZY);
         if (g_hip.lib) {
            break;
         }
      }
      if (!g_hip.lib) {
         g_hip.lib = dlopen("libamdhip64.so", RTLD_LAZY | RTLD_GLOBAL);
      }
      if (!g_hip.lib) {
         snprintf(g_error, sizeof g_error, "fake_rccl: no HIP runtime in this process (%s)", dlerror());
         return;
      }
      bool ok = sym(g_hip.GetDevice, "hipGetDevice") && sym(g_hip.SetDevice, "hipSetDevice") &&
                sym(g_hip.HostRegister, "hipHostRegister") && sym(g_hip.HostUnregister, "hipHostUnregister") &&
                sym(g_hip.MemcpyAsync, "hipMemcpyAsync") && sym(g_hip.StreamWriteValue64, "hipStreamWriteValue64") &&
                sym(g_hip.StreamWaitValue64, "hipStreamWaitValue64") && sym(g_hip.DeviceGetAttribute, "hipDeviceGetAttribute") &&
                sym(g_hip.EventCreateWithFlags, "hipEventCreateWithFlags") && sym(g_hip.EventDestroy, "hipEventDestroy") &&
                sym(g_hip.EventRecord, "hipEventRecord") && sym(g_hip.StreamWaitEvent, "hipStreamWaitEvent") &&
                sym(g_hip.GetErrorString, "hipGetErrorString");
      if (!ok) {
         g_hip.lib = nullptr;
      }
   });
   return g_hip.lib != nullptr;
}

bool hip_ok(hipError_t e, const char *what)
{
   if (e == hipSuccess) {
      return true;
   }
   snprintf(g_error, sizeof g_error, "fake_rccl: %s: %s", what, g_hip.GetErrorString ? g_hip.GetErrorString(e) : "?");
   fprintf(stderr, "%s\n", g_error);
   return false;
}

double now_s()
{
   timespec t;
   clock_gettime(CLOCK_MONOTONIC, &t);
   return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

std::mutex g_lock;                                  // segments and communicators of this process
constexpr int kMaxSegments = 16;
Segment g_segments[kMaxSegments];

size_t segment_bytes(int nranks) { return 4096 + (size_t)kRing * (size_t)nranks * kSlotBytes; }

// maps (creating if need be) the segment of an id; every process sizes it the same, tmpfs hands out zeroes
Segment *segment_attach(const char *name, int nranks)
{
   std::lock_guard<std::mutex> hold(g_lock);
   Segment *slot = nullptr;
   for (Segment &s : g_segments) {
      if (s.users && strcmp(s.name, name) == 0) {
         s.users++;
         return &s;
      }
      if (!s.users && !slot) {
         slot = &s;
      }
   }
   if (!slot) {
      snprintf(g_error, sizeof g_error, "fake_rccl: more than %d communicators alive in one process", kMaxSegments);
      return nullptr;
   }
   const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
   if (fd < 0) {
      snprintf(g_error, sizeof g_error, "fake_rccl: shm_open(%s): %s", name, strerror(errno));
      return nullptr;
   }
   const size_t bytes = segment_bytes(nranks);
   if (ftruncate(fd, (off_t)bytes) != 0) {
      snprintf(g_error, sizeof g_error, "fake_rccl: ftruncate(%s, %zu): %s", name, bytes, strerror(errno));
      close(fd);
      return nullptr;
   }
   void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (p == MAP_FAILED) {
      snprintf(g_error, sizeof g_error, "fake_rccl: mmap(%s): %s", name, strerror(errno));
      return nullptr;
   }
   if (!hip_ok(g_hip.HostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped), "hipHostRegister of the segment")) {
      munmap(p, bytes);
      return nullptr;
   }
   snprintf(slot->name, sizeof slot->name, "%s", name);
   slot->h = static_cast<Header *>(p);
   slot->data = static_cast<uint8_t *>(p) + 4096;
   slot->bytes = bytes;
   slot->nranks = nranks;
   slot->users = 1;
   return slot;
}

void segment_release(Segment *s)
{
   std::lock_guard<std::mutex> hold(g_lock);
   if (--s->users > 0) {
      return;
   }
   (void)g_hip.HostUnregister(s->h);
   munmap(s->h, s->bytes);
   shm_unlink(s->name);                             // (whoever is last: the others' mappings stay valid until they unmap)
   s->h = nullptr;
   s->data = nullptr;
}

} // namespace

struct ncclComm {
   Segment *seg;
   int rank, nranks, device;
   uint64_t seq;                                    // all-gathers enqueued so far
   // communicators of ONE process (ncclCommInitAll): their streams may share a hardware queue -- the high-priority
   // streams of two contexts on one device do --, where a wait-value packet of one rank would keep the other rank's post
   // from ever running.  Their collectives are only taken inside ncclGroupStart / ncclGroupEnd (as the library issues
   // them), where every rank's call is known, and are ordered with HIP events instead of values in memory.
   bool local;
   ncclComm **peers;                                // (local) the communicators of the group, by rank
   hipEvent_t posted_ev[kRing], consumed_ev[kRing];
   bool consumed_set[kRing];
};

namespace {

thread_local int g_group_depth = 0;

// an all-gather of a process-local communicator, held until ncclGroupEnd
struct LocalOp {
   ncclComm *comm;
   const void *send;
   void *recv;
   size_t bytes;
   hipStream_t stream;
};
constexpr int kMaxLocalOps = 64;
thread_local LocalOp g_local_ops[kMaxLocalOps];
thread_local int g_local_n = 0;

// The held all-gathers of one process-local group, all ranks present: every rank's stream copies its send buffer into the
// round's staging slot and records an event; every rank's stream then waits for all of those events and fills its receive
// buffer.  A staging slot is reused kRing rounds later, behind the events that mark its last readers' copies.
ncclResult_t run_local_group(LocalOp *ops, int n)
{
   ncclComm *first = ops[0].comm;
   if (n != first->nranks) {
      snprintf(g_error, sizeof g_error, "fake_rccl: %d of %d ranks of a process-local communicator called ncclAllGather inside the group", n,
               first->nranks);
      return ncclInvalidUsage;
   }
   LocalOp *by_rank[kMaxRanks] = {};
   for (int i = 0; i < n; i++) {
      if (ops[i].comm->seg != first->seg || ops[i].bytes != ops[0].bytes || by_rank[ops[i].comm->rank] ||
          ops[i].comm->seq != first->seq) {
         snprintf(g_error, sizeof g_error, "fake_rccl: the all-gathers of one group do not match (communicator, count, order)");
         return ncclInvalidUsage;
      }
      by_rank[ops[i].comm->rank] = &ops[i];
   }
   const size_t bytes = ops[0].bytes;
   int before = 0;
   (void)g_hip.GetDevice(&before);
   bool ok = true;
   for (size_t off = 0; off < bytes && ok; off += kSlotBytes) {
      const size_t len = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
      const uint64_t s = first->seq;
      const int slot = (int)(s % kRing);
      uint8_t *ring = first->seg->data + (size_t)slot * (size_t)n * kSlotBytes;
      for (int r = 0; r < n && ok; r++) {
         LocalOp &o = *by_rank[r];
         ok = hip_ok(g_hip.SetDevice(o.comm->device), "hipSetDevice");
         for (int q = 0; q < n && ok; q++) {
            if (by_rank[q]->comm->consumed_set[slot]) {
               ok = hip_ok(g_hip.StreamWaitEvent(o.stream, by_rank[q]->comm->consumed_ev[slot], 0), "wait consumed (event)");
            }
         }
         ok = ok && hip_ok(g_hip.MemcpyAsync(ring + (size_t)r * len, static_cast<const uint8_t *>(o.send) + off, len, hipMemcpyDeviceToHost, o.stream),
                           "D2H of the send buffer");
         ok = ok && hip_ok(g_hip.EventRecord(o.comm->posted_ev[slot], o.stream), "post (event)");
      }
      for (int r = 0; r < n && ok; r++) {
         LocalOp &o = *by_rank[r];
         ok = hip_ok(g_hip.SetDevice(o.comm->device), "hipSetDevice");
         for (int q = 0; q < n && ok; q++) {
            if (q != r) {
               ok = hip_ok(g_hip.StreamWaitEvent(o.stream, by_rank[q]->comm->posted_ev[slot], 0), "wait posted (event)");
            }
         }
         for (int q = 0; q < n && ok; q++) {
            ok = hip_ok(g_hip.MemcpyAsync(static_cast<uint8_t *>(o.recv) + (size_t)q * bytes + off, ring + (size_t)q * len, len, hipMemcpyHostToDevice,
                                          o.stream), "H2D of a piece");
         }
         ok = ok && hip_ok(g_hip.EventRecord(o.comm->consumed_ev[slot], o.stream), "consumed (event)");
      }
      for (int r = 0; r < n; r++) {
         by_rank[r]->comm->consumed_set[slot] = true;
         by_rank[r]->comm->seq++;
      }
   }
   (void)g_hip.SetDevice(before);
   if (getenv("FAKE_RCCL_TRACE")) {
      fprintf(stderr, "fake_rccl: process-local all-gather of %d ranks, %zu bytes per rank%s\n", n, bytes, ok ? "" : " FAILED");
   }
   return ok ? ncclSuccess : ncclUnhandledCudaError;
}

size_t dtype_bytes(ncclDataType_t t)
{
   switch (t) {
   case ncclInt8: case ncclUint8: return 1;
   case ncclFloat16: case ncclBfloat16: return 2;
   case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
   case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
   default: return 0;
   }
}

ncclResult_t comm_create(ncclComm_t *out, const char *name, int nranks, int rank, bool wait_for_peers)
{
   if (!hip_runtime()) {
      fprintf(stderr, "%s\n", g_error);
      return ncclSystemError;
   }
   int device = 0;
   if (!hip_ok(g_hip.GetDevice(&device), "hipGetDevice")) {
      return ncclUnhandledCudaError;
   }
   int can_wait = 0;
   if (g_hip.DeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !can_wait) {
      snprintf(g_error, sizeof g_error, "fake_rccl: device %d has no stream wait-value support", device);
      fprintf(stderr, "%s\n", g_error);
      return ncclSystemError;
   }
   Segment *seg = segment_attach(name, nranks);
   if (!seg) {
      fprintf(stderr, "%s\n", g_error);
      return ncclSystemError;
   }
   Header *h = seg->h;
   h->nranks = (uint32_t)nranks;
   h->magic.store(kMagic);
   h->joined.fetch_add(1);
   if (wait_for_peers) {
      const double t0 = now_s();
      while (h->joined.load() < (uint32_t)nranks) {
         if (now_s() - t0 > 120.0) {
            snprintf(g_error, sizeof g_error, "fake_rccl: rank %d of %d: only %u ranks joined within 120 s", rank, nranks, h->joined.load());
            fprintf(stderr, "%s\n", g_error);
            segment_release(seg);
            return ncclSystemError;
         }
         usleep(200);
      }
   }
   ncclComm *c = new ncclComm();
   c->seg = seg;
   c->rank = rank;
   c->nranks = nranks;
   c->device = device;
   c->seq = 0;
   c->local = false;
   c->peers = nullptr;
   for (int k = 0; k < kRing; k++) {
      c->posted_ev[k] = c->consumed_ev[k] = nullptr;
      c->consumed_set[k] = false;
   }
   *out = c;
   if (getenv("FAKE_RCCL_TRACE")) {
      fprintf(stderr, "fake_rccl: rank %d / %d up on device %d (pid %d, segment %s)\n", rank, nranks, device, (int)getpid(), name);
   }
   return ncclSuccess;
}

} // namespace

extern "C" {

// what a test asks to make sure the stand-in (not librccl) served the process
int fake_rccl_loaded(void) { return 1; }

const char *ncclGetErrorString(ncclResult_t r)
{
   switch (r) {
   case ncclSuccess: return "no error";
   case ncclUnhandledCudaError: return g_error[0] ? g_error : "fake_rccl: unhandled HIP error";
   case ncclSystemError: return g_error[0] ? g_error : "fake_rccl: system error";
   case ncclInvalidArgument: return g_error[0] ? g_error : "fake_rccl: invalid argument";
   case ncclInvalidUsage: return g_error[0] ? g_error : "fake_rccl: invalid usage";
   default: return "fake_rccl: error";
   }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
   static std::atomic<uint32_t> counter{0};
   if (!id) {
      return ncclInvalidArgument;
   }
   timespec t;
   clock_gettime(CLOCK_REALTIME, &t);
   memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
   snprintf(id->internal, 60, "/fake-rccl-%d-%lld%09ld-%u", (int)getpid(), (long long)t.tv_sec, t.tv_nsec, counter.fetch_add(1));
   return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
   if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/' || strnlen(id.internal, 64) >= 60) {
      snprintf(g_error, sizeof g_error, "fake_rccl: ncclCommInitRank: bad argument (the id must come from this stand-in's ncclGetUniqueId)");
      return ncclInvalidArgument;
   }
   return comm_create(comm, id.internal, nranks, rank, true);
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
   if (!comms || ndev < 1 || ndev > kMaxRanks) {
      return ncclInvalidArgument;
   }
   if (!hip_runtime()) {
      return ncclSystemError;
   }
   ncclUniqueId id;
   ncclGetUniqueId(&id);
   int before = 0;
   (void)g_hip.GetDevice(&before);
   for (int i = 0; i < ndev; i++) {
      // (unlike librccl, the same device may appear more than once: that is what this stand-in is for)
      if (!hip_ok(g_hip.SetDevice(devlist ? devlist[i] : i), "hipSetDevice")) {
         return ncclUnhandledCudaError;
      }
      const ncclResult_t r = comm_create(&comms[i], id.internal, ndev, i, false);
      if (r != ncclSuccess) {
         return r;
      }
   }
   if (ndev > 1) {
      ncclComm **peers = new ncclComm *[ndev];      // (shared by the group, released with rank 0)
      for (int i = 0; i < ndev; i++) {
         peers[i] = comms[i];
      }
      for (int i = 0; i < ndev; i++) {
         comms[i]->local = true;
         comms[i]->peers = peers;
         for (int k = 0; k < kRing; k++) {
            if (!hip_ok(g_hip.SetDevice(comms[i]->device), "hipSetDevice") ||
                !hip_ok(g_hip.EventCreateWithFlags(&comms[i]->posted_ev[k], hipEventDisableTiming), "hipEventCreate") ||
                !hip_ok(g_hip.EventCreateWithFlags(&comms[i]->consumed_ev[k], hipEventDisableTiming), "hipEventCreate")) {
               return ncclUnhandledCudaError;
            }
         }
      }
   }
   (void)g_hip.SetDevice(before);
   return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
   if (!comm) {
      return ncclInvalidArgument;
   }
   comm->seg->h->left.fetch_add(1);
   for (int k = 0; k < kRing; k++) {
      if (comm->posted_ev[k]) (void)g_hip.EventDestroy(comm->posted_ev[k]);
      if (comm->consumed_ev[k]) (void)g_hip.EventDestroy(comm->consumed_ev[k]);
   }
   if (comm->local && comm->rank == 0) {
      delete[] comm->peers;
   }
   segment_release(comm->seg);
   delete comm;
   return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *async_error)
{
   if (!comm || !async_error) {
      return ncclInvalidArgument;
   }
   *async_error = ncclSuccess;
   return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
   g_group_depth++;
   return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
   if (g_group_depth <= 0) {
      return ncclInvalidUsage;
   }
   if (--g_group_depth > 0) {
      return ncclSuccess;
   }
   // (calls of communicators that span processes were enqueued at once; the process-local ones were held until here)
   ncclResult_t result = ncclSuccess;
   while (g_local_n > 0 && result == ncclSuccess) {
      LocalOp group[kMaxLocalOps];
      int n = 0, kept = 0;
      Segment *seg = g_local_ops[0].comm->seg;
      const uint64_t seq = g_local_ops[0].comm->seq;
      for (int i = 0; i < g_local_n; i++) {
         // one collective per rank and pass: a rank's second call inside the group belongs to the next round
         bool rank_taken = false;
         for (int k = 0; k < n; k++) {
            rank_taken = rank_taken || group[k].comm == g_local_ops[i].comm;
         }
         if (g_local_ops[i].comm->seg == seg && g_local_ops[i].comm->seq == seq && !rank_taken) {
            group[n++] = g_local_ops[i];
         }
         else {
            g_local_ops[kept++] = g_local_ops[i];
         }
      }
      g_local_n = kept;
      result = run_local_group(group, n);
   }
   g_local_n = 0;
   return result;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
   const size_t bytes = sendcount * dtype_bytes(datatype);
   if (!comm || !sendbuff || !recvbuff || dtype_bytes(datatype) == 0) {
      snprintf(g_error, sizeof g_error, "fake_rccl: ncclAllGather: bad argument");
      return ncclInvalidArgument;
   }
   if (bytes == 0) {
      return ncclSuccess;
   }
   if (comm->local) {
      if (g_group_depth <= 0 || g_local_n >= kMaxLocalOps) {
         snprintf(g_error, sizeof g_error, "fake_rccl: the collectives of a process-local communicator (ncclCommInitAll) are only taken inside "
                                           "ncclGroupStart / ncclGroupEnd");
         return ncclInvalidUsage;
      }
      g_local_ops[g_local_n++] = LocalOp{comm, sendbuff, recvbuff, bytes, stream};
      return ncclSuccess;
   }
   int before = 0;
   (void)g_hip.GetDevice(&before);
   if (before != comm->device && !hip_ok(g_hip.SetDevice(comm->device), "hipSetDevice")) {
      return ncclUnhandledCudaError;
   }
   Header *h = comm->seg->h;
   const int n = comm->nranks, r = comm->rank;
   const uint64_t first_seq = comm->seq;
   bool ok = true;
   // pieces of at most a staging slot, each a round of its own (every rank cuts the same way: the counts are equal by contract)
   for (size_t off = 0; off < bytes && ok; off += kSlotBytes) {
      const size_t len = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
      const uint64_t s = comm->seq++;
      // slots of one round lie one behind the other at THIS piece's width
      uint8_t *ring = comm->seg->data + (size_t)(s % kRing) * (size_t)n * kSlotBytes;
      if (s >= (uint64_t)kRing) {
         for (int q = 0; q < n && ok; q++) {
            ok = hip_ok(g_hip.StreamWaitValue64(stream, (void *)&h->consumed[q], s + 1 - kRing, hipStreamWaitValueGte, ~0ull), "wait consumed");
         }
      }
      ok = ok && hip_ok(g_hip.MemcpyAsync(ring + (size_t)r * len, static_cast<const uint8_t *>(sendbuff) + off, len, hipMemcpyDeviceToHost, stream),
                        "D2H of the send buffer");
      ok = ok && hip_ok(g_hip.StreamWriteValue64(stream, (void *)&h->posted[r], s + 1, 0), "post");
      for (int q = 0; q < n && ok; q++) {
         if (q != r) {
            ok = hip_ok(g_hip.StreamWaitValue64(stream, (void *)&h->posted[q], s + 1, hipStreamWaitValueGte, ~0ull), "wait posted");
         }
      }
      if (len == bytes) {
         ok = ok && hip_ok(g_hip.MemcpyAsync(recvbuff, ring, (size_t)n * len, hipMemcpyHostToDevice, stream), "H2D of the table");
      }
      else {
         for (int q = 0; q < n && ok; q++) {
            ok = hip_ok(g_hip.MemcpyAsync(static_cast<uint8_t *>(recvbuff) + (size_t)q * bytes + off, ring + (size_t)q * len, len, hipMemcpyHostToDevice,
                                          stream), "H2D of a piece");
         }
      }
      ok = ok && hip_ok(g_hip.StreamWriteValue64(stream, (void *)&h->consumed[r], s + 1, 0), "consumed");
   }
   if (before != comm->device) {
      (void)g_hip.SetDevice(before);
   }
   const uint64_t s = first_seq;
   if (getenv("FAKE_RCCL_TRACE")) {
      fprintf(stderr, "fake_rccl: rank %d all-gather %llu, %zu bytes per rank%s\n", r, (unsigned long long)s, bytes, ok ? "" : " FAILED");
   }
   return ok ? ncclSuccess : ncclUnhandledCudaError;
}

} // extern "C"
