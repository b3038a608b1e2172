// SPDX-License-Identifier: GPL-3.0-or-later
// mm_multi.hip -- the multi-GPU half of the C ABI: block-aligned partitions and the RCCL
// gather of the per-GPU offset lists over xGMI.
//
// Replaces the reference's thread dispatcher + merge (src/core/search_engine.cpp:66-188,
// :193-197): every SearchBlock (x byte alignment) is an independent chain, so GPU g of G scans
// the blocks [g*nb/G, (g+1)*nb/G) of the file -- a block-aligned partition plus (L-1)*S bytes of
// pattern-length overlap -- with no exchange at all, and the only communication is the gather
// of the ascending per-GPU offset lists.  Partitions are in rank order, so the concatenation
// is globally ascending: there is no merge step.
//
// The collective is issued by this library itself (librccl), from DEVICE memory: mm_rank_scatter
// leaves every scan's ordered list in HBM next to the copy it publishes to the host, and
// ncclAllGather sends that block as it is -- a fixed-width record [8 counters][<= 16 K offsets]
// per rank, one collective, latency bound (8 B per match).  A small kernel then packs the
// records of all ranks into ONE ascending list in pinned host memory; the host only waits for an
// event.  Lists longer than a record (short keywords, padding floods: rare) take a second,
// padded all-gather; every rank sees every count in the first one, so all ranks agree on that
// without further traffic.
//
// Two deployment shapes, same code:
//   * one process per GPU (bench.py under torch.distributed.run): rank 0 makes an id with
//     mmh_comm_unique_id, the launcher's rendezvous distributes it, every rank calls
//     mmh_comm_init_rank;
//   * one process driving several GPUs (SearchEngine<T>::run): mmh_comm_init_all over the
//     contexts, mmh_scan_multi runs the scans on one host thread per device and the collective
//     inside ncclGroupStart / ncclGroupEnd.
// The gather runs on its own stream and is split into start / finish, so that the collective of
// scan k overlaps scan k+1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mm_context.h"
#include "mm_internal.h"
#include "mm_kernels.h"

namespace {

constexpr uint64_t kRecordWords = MM_RESULT_HEADER_WORDS + MM_MAX_RANK_SORT;   // the widest record: what a scan leaves in HBM, sent as it is
// The record's width follows the lists (round 5): every rank learns every rank's extent from the table a gather
// delivers, and a gather slot is only reused once its previous gather has been finished ON EVERY RANK (two slots, at most
// two gathers outstanding, finished oldest first) -- so "the longest extent this slot saw last time" is a number all ranks
// agree on without further traffic.  While it stays below four fifths of a narrow record the next gather of the slot sends
// 8 KiB per rank instead of 128 KiB (one ROM dealt over eight GPUs: ~530 offsets per rank, 64 KiB instead of 1 MiB received
// per rank and gather over xGMI); a list that does not fit the width in use takes the second, padded phase as ever.
constexpr uint32_t kNarrowSlots = 1024 - MM_RESULT_HEADER_WORDS;            // 1016 offsets, 8 KiB per rank
constexpr uint32_t kNarrowKeep = kNarrowSlots * 4 / 5;
constexpr int kMaxRanks = 64;
constexpr uint64_t kMergedHeader = 8 + kMaxRanks;     // [total, longest, nranks, long flag, ...][count of every rank]

bool hip_ok(hipError_t e, const char *what)
{
   if (e == hipSuccess) {
      return true;
   }
   mmh_set_error("%s: %s", what, hipGetErrorString(e));
   return false;
}

bool nccl_ok(ncclResult_t r, const char *what)
{
   if (r == ncclSuccess) {
      return true;
   }
   mmh_set_error("%s: %s", what, ncclGetErrorString(r));
   return false;
}

#define HIP_TRY(expr)                       \
   do {                                     \
      if (!hip_ok((expr), #expr)) {         \
         return MMH_E_DEVICE;               \
      }                                     \
   } while (0)

#define NCCL_TRY(expr)                      \
   do {                                     \
      if (!nccl_ok((expr), #expr)) {        \
         return MMH_E_DEVICE;               \
      }                                     \
   } while (0)

double now_s()
{
   return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

// --------------------------------------------------------------------------
// kernels
// --------------------------------------------------------------------------

// Record of a list that lives in host memory (already copied to rec + 8 when it fits): only the
// word the receivers read needs writing -- "count + 1", as mm_rank_scatter publishes it.
__global__ void mm_gather_stamp(uint64_t *rec, uint64_t count)
{
   if (threadIdx.x < MM_RESULT_HEADER_WORDS) {
      rec[threadIdx.x] = threadIdx.x == 6 ? count + 1 : (threadIdx.x == 0 ? count : 0);
   }
}

// table: nranks records of `words` words as the all-gather left them.  Block r copies the list of
// rank r behind the lists of the ranks before it -- one ascending list in pinned host memory --
// and block 0 writes the header: [0] total, [1] longest list, [2] nranks, [3] 1 when some list is
// longer than a record (then nothing is copied: the second phase follows), [8 + r] count of rank r.
// limit: the longest list (or, for a list with holes, slot count) a record of this table can hold -- MM_MAX_RANK_SORT for
// the records an all-gather delivered; anything longer only exists in full in its rank's own published block
// (up to MM_MAX_PUBLISH slots), which the second phase packs with limit = MM_MAX_PUBLISH.
__global__ __launch_bounds__(256) void mm_gather_pack(const uint64_t *table, uint32_t nranks, uint64_t words, uint64_t *merged,
                                                      uint64_t merged_cap, uint32_t want_list, uint32_t limit)
{
   __shared__ unsigned long long sh_before, sh_total, sh_longest;
   const uint32_t r = blockIdx.x;
   if (threadIdx.x == 0) {
      unsigned long long before = 0, total = 0, longest = 0;
      for (uint32_t q = 0; q < nranks; q++) {
         const unsigned long long w6 = table[(uint64_t)q * words + 6];
         const unsigned long long n = w6 ? w6 - 1 : 0;
         // what the record would have to hold: the list, or one slot per candidate when it has holes
         const unsigned long long slots_q = (table[(uint64_t)q * words + 4] & 1) ? table[(uint64_t)q * words] : 0;
         const unsigned long long extent = slots_q > n ? slots_q : n;
         before += q < r ? n : 0;
         total += n;
         longest = extent > longest ? extent : longest;
         if (r == 0) {
            merged[8 + q] = n;
         }
      }
      sh_before = before; sh_total = total; sh_longest = longest;
      if (r == 0) {
         merged[0] = total; merged[1] = longest; merged[2] = nranks;
         merged[3] = longest > limit ? 1 : 0;
      }
   }
   __syncthreads();
   if (!want_list || sh_longest > limit || sh_total > merged_cap) {
      return;
   }
   const uint64_t *rec = table + (uint64_t)r * words;
   const unsigned long long n = rec[6] ? rec[6] - 1 : 0;
   uint64_t *dst = merged + kMergedHeader + sh_before;
   if ((rec[4] & 1) == 0 || rec[0] == n) {
      for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) {
         dst[i] = rec[MM_RESULT_HEADER_WORDS + i];
      }
      return;
   }
   // a fused scan's record with holes: one slot per candidate (rec[0] of them), ~0 = not reported.
   // Ordered compaction, 256 slots at a time.
   __shared__ unsigned int wave_count[4];
   unsigned long long kept = 0;                     // block uniform
   const unsigned long long slots = rec[0];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   for (unsigned long long base = 0; base < slots; base += blockDim.x) {
      const unsigned long long i = base + threadIdx.x;
      const uint64_t v = i < slots ? rec[MM_RESULT_HEADER_WORDS + i] : ~0ull;
      const bool keep = v != ~0ull;
      const unsigned long long mask = __ballot(keep);
      if (lane == 0) {
         wave_count[wave] = (unsigned int)__popcll(mask);
      }
      __syncthreads();
      unsigned int before = 0, all = 0;
      for (int w = 0; w < 4; w++) {
         before += w < wave ? wave_count[w] : 0;
         all += wave_count[w];
      }
      if (keep) {
         dst[kept + before + __popcll(mask & ((1ull << lane) - 1))] = v;
      }
      kept += all;
      __syncthreads();
   }
}

// --------------------------------------------------------------------------
// partitions (host only)
// --------------------------------------------------------------------------

extern "C" int mmh_partition(uint64_t total_bytes, uint64_t block_bytes, uint32_t keyword_len, uint32_t elem_bytes, int rank,
                             int nranks, uint64_t *first_byte, uint64_t *nbytes)
{
   if (!first_byte || !nbytes || block_bytes == 0 || keyword_len == 0 || (elem_bytes != 1 && elem_bytes != 2) || nranks < 1 ||
       rank < 0 || rank >= nranks) {
      mmh_set_error("mmh_partition: bad argument");
      return MMH_E_ARG;
   }
   // whole reference blocks (search_engine.cpp:218-253), dealt out evenly in rank order
   const uint64_t nblocks = (total_bytes + block_bytes - 1) / block_bytes;
   const uint64_t b0 = (unsigned __int128)nblocks * (uint64_t)rank / (uint64_t)nranks;
   const uint64_t b1 = (unsigned __int128)nblocks * (uint64_t)(rank + 1) / (uint64_t)nranks;
   const uint64_t first = b0 * block_bytes;
   // pattern-length overlap into the next partition: the last block of this one reads it (:227-230)
   uint64_t end = b1 * block_bytes + (uint64_t)(keyword_len - 1) * elem_bytes;
   end = end > total_bytes ? total_bytes : end;
   *first_byte = first;
   // (more ranks than blocks: a rank without a block scans nothing -- the overlap alone would be
   // taken for a block of its own and report its matches a second time)
   *nbytes = (b1 > b0 && end > first) ? end - first : 0;
   return MMH_OK;
}

// --------------------------------------------------------------------------
// communicator
// --------------------------------------------------------------------------

namespace {

int comm_buffers(mmh_ctx *c)
{
   MmComm &m = c->mg;
   HIP_TRY(hipSetDevice(c->device));
   int rc = mmh_workspace(c);                       // the scan workspace (and its device-side result copies)
   if (rc != MMH_OK) {
      return rc;
   }
   if (!m.stream) {
      // highest priority: when a wave slot frees up while a scan is streaming, the collective's
      // kernel gets it first
      int least = 0, greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_TRY(hipStreamCreateWithPriority(&m.stream, hipStreamNonBlocking, greatest));
   }
   if (!m.d_send) {
      HIP_TRY(hipMalloc(&m.d_send, kRecordWords * sizeof(uint64_t)));
   }
   for (auto &s : m.slot) {
      if (!s.d_table) {
         HIP_TRY(hipMalloc(&s.d_table, (uint64_t)m.nranks * kRecordWords * sizeof(uint64_t)));
         s.merged_cap = (uint64_t)m.nranks * MM_MAX_RANK_SORT;
         HIP_TRY(hipHostMalloc(&s.h_merged, (kMergedHeader + s.merged_cap) * sizeof(uint64_t), hipHostMallocDefault));
         HIP_TRY(hipEventCreate(&s.begin));
         HIP_TRY(hipEventCreate(&s.end));
      }
      s.busy = false;
      s.limit = MM_MAX_RANK_SORT;
      s.last_longest = ~0ull;                         // (nothing seen yet: the wide record)
   }
   m.turn = m.oldest = 0;
   return MMH_OK;
}

void comm_release(mmh_ctx *c)
{
   MmComm &m = c->mg;
   (void)hipSetDevice(c->device);
   if (m.stream) {
      (void)hipStreamSynchronize(m.stream);
   }
   if (m.comm) {
      (void)ncclCommDestroy(static_cast<ncclComm_t>(m.comm));
      m.comm = nullptr;
   }
   for (auto &s : m.slot) {
      if (s.d_table) (void)hipFree(s.d_table);
      if (s.d_keep) (void)hipFree(s.d_keep);
      if (s.h_merged) (void)hipHostFree(s.h_merged);
      if (s.begin) (void)hipEventDestroy(s.begin);
      if (s.end) (void)hipEventDestroy(s.end);
      s = MmGatherSlot();
   }
   if (m.d_send) (void)hipFree(m.d_send);
   if (m.d_long) (void)hipFree(m.d_long);
   if (m.d_long_table) (void)hipFree(m.d_long_table);
   if (m.stream) (void)hipStreamDestroy(m.stream);
   m.d_send = m.d_long = m.d_long_table = nullptr;
   m.long_cap = m.long_table_cap = 0;
   m.stream = nullptr;
   m.rank = 0;
   m.nranks = 1;
   m.last_list.clear();
   m.last_src = nullptr;
   m.last_end = nullptr;
}

} // namespace

extern "C" int mmh_comm_unique_id(void *id128)
{
   if (!id128) {
      mmh_set_error("mmh_comm_unique_id: null argument");
      return MMH_E_ARG;
   }
   static_assert(MMH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
   ncclUniqueId id;
   NCCL_TRY(ncclGetUniqueId(&id));
   std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
   return MMH_OK;
}

extern "C" int mmh_comm_init_rank(mmh_ctx *c, const void *id128, int nranks, int rank)
{
   if (!c || !id128 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) {
      mmh_set_error("mmh_comm_init_rank: bad argument (1 <= nranks <= %d)", kMaxRanks);
      return MMH_E_ARG;
   }
   comm_release(c);
   HIP_TRY(hipSetDevice(c->device));
   ncclUniqueId id;
   std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
   ncclComm_t comm = nullptr;
   NCCL_TRY(ncclCommInitRank(&comm, nranks, id, rank));
   c->mg.comm = comm;
   c->mg.rank = rank;
   c->mg.nranks = nranks;
   return comm_buffers(c);
}

extern "C" int mmh_comm_init_all(mmh_ctx *const *ctxs, int n)
{
   if (!ctxs || n < 1 || n > kMaxRanks) {
      mmh_set_error("mmh_comm_init_all: bad argument (1 <= n <= %d)", kMaxRanks);
      return MMH_E_ARG;
   }
   std::vector<int> devices(n);
   for (int i = 0; i < n; i++) {
      if (!ctxs[i]) {
         mmh_set_error("mmh_comm_init_all: null context %d", i);
         return MMH_E_ARG;
      }
      devices[i] = ctxs[i]->device;
      comm_release(ctxs[i]);
   }
   std::vector<ncclComm_t> comms(n, nullptr);
   // (one context per device: RCCL itself refuses a device that appears twice -- "invalid usage")
   const ncclResult_t made = ncclCommInitAll(comms.data(), n, devices.data());
   if (made != ncclSuccess) {
      bool twice = false;
      for (int i = 0; i < n; i++) {
         for (int k = 0; k < i; k++) {
            twice = twice || devices[k] == devices[i];
         }
      }
      mmh_set_error("ncclCommInitAll over %d contexts: %s%s", n, ncclGetErrorString(made),
                    twice ? " (two of the contexts are on the same device: one rank per GPU)" : "");
      return made == ncclInvalidUsage || made == ncclInvalidArgument ? MMH_E_ARG : MMH_E_DEVICE;
   }
   for (int i = 0; i < n; i++) {
      ctxs[i]->mg.comm = comms[i];
      ctxs[i]->mg.rank = i;
      ctxs[i]->mg.nranks = n;
   }
   for (int i = 0; i < n; i++) {
      int rc = comm_buffers(ctxs[i]);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   return MMH_OK;
}

extern "C" int mmh_comm_info(mmh_ctx *c, int *rank, int *nranks)
{
   if (!c || !rank || !nranks) {
      mmh_set_error("mmh_comm_info: bad argument");
      return MMH_E_ARG;
   }
   *rank = c->mg.rank;
   *nranks = c->mg.comm ? c->mg.nranks : 0;
   return MMH_OK;
}

extern "C" void mmh_comm_destroy(mmh_ctx *c)
{
   if (c && (c->mg.comm || c->mg.stream)) {
      comm_release(c);
   }
}

// --------------------------------------------------------------------------
// the gather, step by step (so that several devices of one process can take the
// collective step together inside ncclGroupStart / ncclGroupEnd)
// --------------------------------------------------------------------------

namespace {

// After a failure the slots' next gathers send the WIDE record whatever their previous ones saw: a rank that left a gather
// early (a time-out, an error between two contexts' steps) no longer knows what its peers saw, and ranks that disagree
// about the width would issue all-gathers of different counts.
void gather_reset_width(mmh_ctx *c)
{
   for (MmGatherSlot &s : c->mg.slot) {
      s.last_longest = ~0ull;
   }
}

// step 1: the record to send.  offsets == nullptr: the list of the most recent mmh_scan -- in
// place in HBM when the scan ordered it there, else from the host copy the scan kept.
int gather_prepare(mmh_ctx *c, const uint64_t *offsets, uint64_t n, const uint64_t **send)
{
   MmComm &m = c->mg;
   if (!m.comm) {
      mmh_set_error("mmh_gather_start: the context has no communicator (mmh_comm_init_rank / mmh_comm_init_all)");
      return MMH_E_STATE;
   }
   MmGatherSlot &s = m.slot[m.turn];
   if (s.busy) {
      mmh_set_error("mmh_gather_start: two gathers are already outstanding, finish the older one first");
      return MMH_E_STATE;
   }
   HIP_TRY(hipSetDevice(c->device));
   const double t0 = now_s();
   // (the width is the slot's state, the same on every rank while every rank completes every gather: what the previous
   // gather's table said.  A failure anywhere puts the slots back to the wide record, gather_reset_width.)
   s.limit = s.last_longest <= kNarrowKeep ? kNarrowSlots : (uint32_t)MM_MAX_RANK_SORT;
   s.from_host = true;
   s.kept = false;                                  // (of the slot's previous gather)
   if (!offsets && n == 0) {
      if (m.last_src) {
         s.from_host = false;
         s.src = m.last_src;
         s.local_count = m.last_count;
         *send = m.last_src;
         if (m.last_end) {
            // the scan's kernels ran on another stream; its end shows in pinned memory before the kernel retires
            HIP_TRY(hipStreamWaitEvent(m.stream, m.last_end, 0));
         }
         // A list (or slot count) beyond a record only travels in the second phase, which mmh_gather_finish enqueues:
         // keep the block now -- the result copy itself is only protected until the first phase has ended
         // (wait_for_gather_reading in mm_capi.hip), and a scan that retries, the scan after next or a second
         // outstanding gather publish into it again before this gather is finished.
         const uint64_t extent = std::max(m.last_count, m.last_slots);
         s.kept = extent > s.limit;
         if (s.kept) {
            const uint64_t words = MM_RESULT_HEADER_WORDS + std::min<uint64_t>(extent, MM_MAX_PUBLISH);
            if (words > s.keep_cap) {
               if (s.d_keep) {
                  HIP_TRY(hipFree(s.d_keep));
                  s.d_keep = nullptr;
                  s.keep_cap = 0;
               }
               HIP_TRY(hipMalloc(&s.d_keep, (words + words / 4) * sizeof(uint64_t)));
               s.keep_cap = words + words / 4;
            }
            HIP_TRY(hipMemcpyAsync(s.d_keep, m.last_src, words * sizeof(uint64_t), hipMemcpyDeviceToDevice, m.stream));
         }
      }
      else {
         offsets = m.last_list.data();
         n = m.last_list.size();
      }
   }
   HIP_TRY(hipEventRecord(s.begin, m.stream));
   if (s.from_host) {
      s.local_count = n;
      if (n && n <= s.limit) {
         HIP_TRY(hipMemcpyAsync(m.d_send + MM_RESULT_HEADER_WORDS, offsets, n * sizeof(uint64_t), hipMemcpyHostToDevice, m.stream));
      }
      hipLaunchKernelGGL(mm_gather_stamp, dim3(1), dim3(64), 0, m.stream, m.d_send, n);
      HIP_TRY(hipGetLastError());
      if (n > s.limit) {
         s.host_list.assign(offsets, offsets + n);      // the second phase sends it from here, whatever scans run in between
      }
      else {
         s.host_list.clear();
      }
      *send = m.d_send;
   }
   s.start_wall_s = now_s() - t0;
   return MMH_OK;
}

// step 2: the collective (non-blocking enqueue on the comm stream)
int gather_collective(mmh_ctx *c, const uint64_t *send)
{
   MmComm &m = c->mg;
   MmGatherSlot &s = m.slot[m.turn];
   NCCL_TRY(ncclAllGather(send, s.d_table, MM_RESULT_HEADER_WORDS + (uint64_t)s.limit, ncclUint64, static_cast<ncclComm_t>(m.comm), m.stream));
   return MMH_OK;
}

// step 3: pack the records into one list in pinned memory
int gather_pack(mmh_ctx *c, int want_list)
{
   MmComm &m = c->mg;
   MmGatherSlot &s = m.slot[m.turn];
   HIP_TRY(hipSetDevice(c->device));
   const double t0 = now_s();
   hipLaunchKernelGGL(mm_gather_pack, dim3((unsigned)m.nranks), dim3(256), 0, m.stream, s.d_table, (uint32_t)m.nranks,
                      MM_RESULT_HEADER_WORDS + (uint64_t)s.limit, s.h_merged, s.merged_cap, want_list ? 1u : 0u, s.limit);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipEventRecord(s.end, m.stream));
   s.busy = true;
   s.start_wall_s += now_s() - t0;
   m.turn ^= 1;
   return MMH_OK;
}

// second phase, step 1: this rank's whole list, padded to the longest one, in d_long
int long_prepare(mmh_ctx *c, MmGatherSlot &s, uint64_t longest)
{
   MmComm &m = c->mg;
   HIP_TRY(hipSetDevice(c->device));
   if (longest > m.long_cap) {
      if (m.d_long) {
         HIP_TRY(hipFree(m.d_long));
         m.d_long = nullptr;
      }
      HIP_TRY(hipMalloc(&m.d_long, (kMergedHeader + longest) * sizeof(uint64_t)));
      m.long_cap = longest;
   }
   const uint64_t need = (uint64_t)m.nranks * longest;
   if (need > m.long_table_cap) {
      if (m.d_long_table) {
         HIP_TRY(hipFree(m.d_long_table));
         m.d_long_table = nullptr;
      }
      HIP_TRY(hipMalloc(&m.d_long_table, need * sizeof(uint64_t)));
      m.long_table_cap = need;
   }
   if (s.local_count) {
      // (a host list that fitted its record -- ANOTHER rank's list is the long one -- was not kept on the host: it sits in
      // this rank's record of the gathered table like a short device list.  Found by the first run with two ranks:
      // round 4 kept no such list and failed here with "lost its host list".)
      if (s.from_host && s.local_count > s.limit) {
         if (s.host_list.size() != s.local_count) {
            mmh_set_error("mmh_gather_finish: the gather slot lost its host list (internal error)");
            return MMH_E_STATE;
         }
         HIP_TRY(hipMemcpyAsync(m.d_long + kMergedHeader, s.host_list.data(), s.local_count * sizeof(uint64_t), hipMemcpyHostToDevice,
                                m.stream));
      }
      else {
         // The scan's published block as it was when the gather started -- NOT the workspace's result copy, which later
         // scans may have published into since the first phase ended: a long list was copied to the slot's own buffer
         // then, a short one sits intact in this rank's record of the table the first phase gathered.  It may have
         // holes (one slot per candidate): the packing kernel on that one block leaves the list behind a header nobody reads.
         const uint64_t *block = s.kept ? s.d_keep : s.d_table + (uint64_t)m.rank * (MM_RESULT_HEADER_WORDS + (uint64_t)s.limit);
         hipLaunchKernelGGL(mm_gather_pack, dim3(1), dim3(256), 0, m.stream, block, 1u, (uint64_t)MM_RESULT_BLOCK_WORDS,
                            m.d_long, longest, 1u, (uint32_t)MM_MAX_PUBLISH);
         HIP_TRY(hipGetLastError());
      }
   }
   return MMH_OK;
}

int long_collective(mmh_ctx *c, uint64_t longest)
{
   MmComm &m = c->mg;
   NCCL_TRY(ncclAllGather(m.d_long + kMergedHeader, m.d_long_table, longest, ncclUint64, static_cast<ncclComm_t>(m.comm), m.stream));
   return MMH_OK;
}

// A collective only ends when every rank has joined it: a peer that died or never called
// mmh_gather_start would leave hipEventSynchronize waiting for good.  Poll instead, watch the
// communicator's asynchronous error state, and give up loudly after MMOORE_GATHER_TIMEOUT_S
// (default 120 s).  After a time-out the communicator is unusable: destroy the context.
double gather_timeout_s()
{
   static const double limit = [] {
      const char *e = getenv("MMOORE_GATHER_TIMEOUT_S");
      const double v = e && *e ? atof(e) : 0;
      return v > 0 ? v : 120.0;
   }();
   return limit;
}

int wait_collective(MmComm &m, hipEvent_t done)
{
   const double t0 = now_s();
   for (uint64_t spins = 1;; spins++) {
      const hipError_t e = hipEventQuery(done);
      if (e == hipSuccess) {
         return MMH_OK;
      }
      if (e != hipErrorNotReady) {
         (void)hip_ok(e, "hipEventQuery (gather)");
         return MMH_E_DEVICE;
      }
      if (spins % 512 == 0) {
         ncclResult_t async = ncclSuccess;
         if (ncclCommGetAsyncError(static_cast<ncclComm_t>(m.comm), &async) == ncclSuccess && async != ncclSuccess &&
             async != ncclInProgress) {
            mmh_set_error("mmh_gather_finish: the communicator reports %s", ncclGetErrorString(async));
            return MMH_E_DEVICE;
         }
         const double waited = now_s() - t0;
         if (waited > gather_timeout_s()) {
            mmh_set_error("mmh_gather_finish: the collective of rank %d / %d did not end within %.0f s -- a peer rank is missing "
                          "or stuck (MMOORE_GATHER_TIMEOUT_S)", m.rank, m.nranks, gather_timeout_s());
            return MMH_E_DEVICE;
         }
         if (waited > 0.002) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
         }
      }
   }
}

// wait for the first phase of the oldest gather; *longest > the slot's record limit: second phase needed
int gather_wait(mmh_ctx *c, uint64_t *total, uint64_t *longest)
{
   MmComm &m = c->mg;
   MmGatherSlot &s = m.slot[m.oldest];
   if (!m.comm || !s.busy) {
      mmh_set_error("mmh_gather_finish: no gather is outstanding");
      return MMH_E_STATE;
   }
   HIP_TRY(hipSetDevice(c->device));
   const int rc = wait_collective(m, s.end);
   if (rc != MMH_OK) {
      return rc;
   }
   *total = s.h_merged[0];
   *longest = s.h_merged[1];
   s.last_longest = *longest;                       // (what the slot's NEXT gather sizes its record by: the same on every rank)
   return MMH_OK;
}

// deliver the oldest gather (first phase waited for; second phase, if any, enqueued) and retire it
int gather_deliver(mmh_ctx *c, uint64_t *out, uint64_t cap, uint64_t *out_count, double t0)
{
   MmComm &m = c->mg;
   MmGatherSlot &s = m.slot[m.oldest];
   const uint64_t total = s.h_merged[0], longest = s.h_merged[1];
   *out_count = total;
   if (out && total > cap) {
      mmh_set_error("mmh_gather_finish: %llu offsets do not fit the caller's buffer of %llu (finish again)",
                    (unsigned long long)total, (unsigned long long)cap);
      return MMH_E_CAPACITY;                         // the gather stays outstanding
   }
   if (longest > s.limit) {
      if (out) {
         uint64_t at = 0;
         for (int r = 0; r < m.nranks; r++) {
            const uint64_t n = s.h_merged[8 + r];
            if (n) {
               HIP_TRY(hipMemcpyAsync(out + at, m.d_long_table + (uint64_t)r * longest, n * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                      m.stream));
            }
            at += n;
         }
      }
      HIP_TRY(hipEventRecord(s.end, m.stream));      // (the gather's device time then covers both phases)
      const int rc = wait_collective(m, s.end);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   else if (out && total) {
      std::memcpy(out, s.h_merged + kMergedHeader, total * sizeof(uint64_t));
   }
   float ms = 0;
   if (hipEventElapsedTime(&ms, s.begin, s.end) == hipSuccess) {
      m.last_device_ms = ms;
   }
   m.last_wall_ms = (s.start_wall_s + (now_s() - t0)) * 1e3;
   s.busy = false;
   m.oldest ^= 1;
   return MMH_OK;
}

} // namespace

extern "C" int mmh_gather_start(mmh_ctx *c, const uint64_t *offsets, uint64_t n, int want_list)
{
   if (!c || (!offsets && n)) {
      mmh_set_error("mmh_gather_start: bad argument");
      return MMH_E_ARG;
   }
   const uint64_t *send = nullptr;
   int rc = gather_prepare(c, offsets, n, &send);
   if (rc == MMH_OK) {
      rc = gather_collective(c, send);
   }
   if (rc == MMH_OK) {
      rc = gather_pack(c, want_list);
   }
   if (rc != MMH_OK && rc != MMH_E_STATE) {                  // (MMH_E_STATE: refused before anything was touched)
      gather_reset_width(c);
   }
   return rc;
}

extern "C" int mmh_gather_finish(mmh_ctx *c, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   if (!c || !out_count || (!out && cap)) {
      mmh_set_error("mmh_gather_finish: bad argument");
      return MMH_E_ARG;
   }
   *out_count = 0;
   const double t0 = now_s();
   uint64_t total = 0, longest = 0;
   int rc = gather_wait(c, &total, &longest);
   if (rc != MMH_OK) {
      if (rc != MMH_E_STATE) {
         gather_reset_width(c);
      }
      return rc;
   }
   MmGatherSlot &s = c->mg.slot[c->mg.oldest];
   if (longest > s.limit && !(out && total > cap)) {
      // some rank's list did not fit its record: every rank saw that in the table it received,
      // so all of them are here now -- the whole lists, padded to the longest
      rc = long_prepare(c, s, longest);
      if (rc == MMH_OK) {
         rc = long_collective(c, longest);
      }
      if (rc != MMH_OK) {
         gather_reset_width(c);
         return rc;
      }
   }
   rc = gather_deliver(c, out, cap, out_count, t0);
   if (rc != MMH_OK && rc != MMH_E_CAPACITY) {               // (MMH_E_CAPACITY: the gather stays outstanding, nothing is lost)
      gather_reset_width(c);
   }
   return rc;
}

extern "C" int mmh_last_gather_timings(mmh_ctx *c, float *ms2)
{
   if (!c || !ms2) {
      mmh_set_error("mmh_last_gather_timings: bad argument");
      return MMH_E_ARG;
   }
   ms2[0] = c->mg.last_device_ms;
   ms2[1] = (float)c->mg.last_wall_ms;
   return MMH_OK;
}

// --------------------------------------------------------------------------
// self-test hook: the packing kernel on records of MANY ranks without a wire
// --------------------------------------------------------------------------
//
// The development box has one GPU, so a real all-gather never delivers more than one record there.
// This entry feeds mm_gather_pack the table an N-rank all-gather WOULD have left (nranks records of
// MMH_GATHER_RECORD_WORDS words, host memory) and returns what mmh_gather_finish would deliver.

extern "C" int mmh_selftest_gather_pack(mmh_ctx *c, const uint64_t *records, int nranks, uint64_t *out, uint64_t cap,
                                        uint64_t *out_count, uint64_t *longest)
{
   if (!c || !records || nranks < 1 || nranks > kMaxRanks || !out_count || !longest || (!out && cap)) {
      mmh_set_error("mmh_selftest_gather_pack: bad argument");
      return MMH_E_ARG;
   }
   static_assert(MMH_GATHER_RECORD_WORDS == kRecordWords, "record width");
   HIP_TRY(hipSetDevice(c->device));
   const uint64_t merged_cap = (uint64_t)nranks * MM_MAX_RANK_SORT;
   uint64_t *d_table = nullptr, *h_merged = nullptr;
   HIP_TRY(hipMalloc(&d_table, (uint64_t)nranks * kRecordWords * sizeof(uint64_t)));
   int rc = MMH_OK;
   if (!hip_ok(hipHostMalloc(&h_merged, (kMergedHeader + merged_cap) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc")) {
      rc = MMH_E_DEVICE;
   }
   if (rc == MMH_OK && !hip_ok(hipMemcpyAsync(d_table, records, (uint64_t)nranks * kRecordWords * sizeof(uint64_t), hipMemcpyHostToDevice,
                                              c->stream), "hipMemcpyAsync")) {
      rc = MMH_E_DEVICE;
   }
   if (rc == MMH_OK) {
      hipLaunchKernelGGL(mm_gather_pack, dim3((unsigned)nranks), dim3(256), 0, c->stream, d_table, (uint32_t)nranks, kRecordWords, h_merged,
                         merged_cap, 1u, (uint32_t)MM_MAX_RANK_SORT);
      if (!hip_ok(hipGetLastError(), "mm_gather_pack") || !hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) {
         rc = MMH_E_DEVICE;
      }
   }
   if (rc == MMH_OK) {
      *out_count = h_merged[0];
      *longest = h_merged[1];
      if (*longest <= MM_MAX_RANK_SORT) {
         if (*out_count > cap) {
            mmh_set_error("mmh_selftest_gather_pack: %llu offsets do not fit", (unsigned long long)*out_count);
            rc = MMH_E_CAPACITY;
         }
         else if (*out_count) {
            std::memcpy(out, h_merged + kMergedHeader, *out_count * sizeof(uint64_t));
         }
      }
   }
   (void)hipFree(d_table);
   if (h_merged) (void)hipHostFree(h_merged);
   return rc;
}

// --------------------------------------------------------------------------
// one process, several GPUs
// --------------------------------------------------------------------------

static int scan_multi_checked(mmh_ctx *const *ctxs, int n, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                              const uint64_t *base_offsets, uint64_t *out, uint64_t cap, uint64_t *out_count);

extern "C" int mmh_scan_multi(mmh_ctx *const *ctxs, int n, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                              const uint64_t *base_offsets, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   if (!ctxs || n < 1 || n > kMaxRanks || !plan || !base_offsets || !out_count || (!out && cap)) {
      mmh_set_error("mmh_scan_multi: bad argument");
      return MMH_E_ARG;
   }
   *out_count = 0;
   for (int i = 0; i < n; i++) {
      if (!ctxs[i] || !ctxs[i]->mg.comm || ctxs[i]->mg.nranks != n || ctxs[i]->mg.rank != i) {
         mmh_set_error("mmh_scan_multi: context %d is not rank %d of a communicator of %d (mmh_comm_init_all)", i, i, n);
         return MMH_E_STATE;
      }
   }
   const int rc = scan_multi_checked(ctxs, n, plan, block_bytes, big_endian, base_offsets, out, cap, out_count);
   if (rc != MMH_OK && rc != MMH_E_CAPACITY) {
      // a step failed after some contexts had taken it: their slots no longer agree about the record width
      for (int i = 0; i < n; i++) {
         gather_reset_width(ctxs[i]);
      }
   }
   return rc;
}

static int scan_multi_checked(mmh_ctx *const *ctxs, int n, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                              const uint64_t *base_offsets, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   // the scans: one host thread per device (a scan ends in a host wait); nothing is exchanged
   std::vector<int> rcs(n, MMH_OK);
   std::vector<std::string> errors(n);
   auto scan_one = [&](int i) {
      std::vector<uint64_t> local(4096);
      uint64_t count = 0;
      for (;;) {
         rcs[i] = mmh_scan(ctxs[i], plan, block_bytes, big_endian, base_offsets[i], local.data(), local.size(), &count);
         if (rcs[i] != MMH_E_CAPACITY) {
            break;
         }
         local.resize(count + 16);
      }
      if (rcs[i] != MMH_OK) {
         errors[i] = mmh_last_error();
      }
   };
   std::vector<std::thread> workers;
   for (int i = 1; i < n; i++) {
      workers.emplace_back(scan_one, i);
   }
   scan_one(0);
   for (auto &t : workers) {
      t.join();
   }
   for (int i = 0; i < n; i++) {
      if (rcs[i] != MMH_OK) {
         mmh_set_error("mmh_scan_multi: device %d: %s", ctxs[i]->device, errors[i].c_str());
         return rcs[i];                               // before any collective: nobody is left waiting
      }
   }
   // the gather: every device's collective call inside one group
   std::vector<const uint64_t *> send(n, nullptr);
   for (int i = 0; i < n; i++) {
      int rc = gather_prepare(ctxs[i], nullptr, 0, &send[i]);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   NCCL_TRY(ncclGroupStart());
   for (int i = 0; i < n; i++) {
      int rc = gather_collective(ctxs[i], send[i]);
      if (rc != MMH_OK) {
         (void)ncclGroupEnd();
         return rc;
      }
   }
   NCCL_TRY(ncclGroupEnd());
   for (int i = 0; i < n; i++) {
      int rc = gather_pack(ctxs[i], i == 0);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   const double t0 = now_s();
   uint64_t total = 0, longest = 0;
   for (int i = 0; i < n; i++) {
      int rc = gather_wait(ctxs[i], &total, &longest);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   const bool fits = !(out && total > cap);
   if (longest > ctxs[0]->mg.slot[ctxs[0]->mg.oldest].limit && fits) {
      for (int i = 0; i < n; i++) {
         int rc = long_prepare(ctxs[i], ctxs[i]->mg.slot[ctxs[i]->mg.oldest], longest);
         if (rc != MMH_OK) {
            return rc;
         }
      }
      NCCL_TRY(ncclGroupStart());
      for (int i = 0; i < n; i++) {
         int rc = long_collective(ctxs[i], longest);
         if (rc != MMH_OK) {
            (void)ncclGroupEnd();
            return rc;
         }
      }
      NCCL_TRY(ncclGroupEnd());
   }
   int result = MMH_OK;
   for (int i = n - 1; i >= 0; i--) {
      uint64_t count = 0;
      int rc = gather_deliver(ctxs[i], i == 0 ? out : nullptr, i == 0 ? cap : 0, &count, t0);
      if (i == 0) {
         *out_count = count;
         if (rc == MMH_E_CAPACITY) {
            // the caller scans again with a larger buffer: retire the gather
            ctxs[0]->mg.slot[ctxs[0]->mg.oldest].busy = false;
            ctxs[0]->mg.oldest ^= 1;
         }
      }
      if (rc != MMH_OK) {
         result = rc;
      }
   }
   return result;
}
