// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi.hip -- the C ABI of include/mmoore_hip.h: device context, ROM
// ownership, and the orchestration of one scan on one HIP stream.
//
// There is no CPU compute path here: every entry point that touches data needs
// a HIP device and fails with MMH_E_DEVICE otherwise.  Host work: the pattern
// plan (mm_plan.cpp), sizing buffers, choosing the engine from the counters a
// scan publishes (second resolver phase, forward engine on flagged domains or
// on everything), and merging lists of different engines in the rare mixed case.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mm_context.h"
#include "mm_internal.h"
#include "mm_kernels.h"

namespace {

thread_local std::string g_error;

constexpr uint64_t kHeaderWords = MM_RESULT_HEADER_WORDS;   // counters in front of the ordered list (pinned host memory)
constexpr uint32_t kMaxRankSort = MM_MAX_RANK_SORT;         // longest list the device orders
constexpr uint64_t kInitialCap = 1u << 20;
constexpr uint64_t kSplitMinBytes = 1ull << 30;             // ROMs from this size on are scanned as a pipeline of parts (scan_split)
constexpr uint64_t kSplitUnitMin = 256ull << 20;            // ... of at least this many bytes each

bool hip_ok(hipError_t e, const char *what)
{
   if (e == hipSuccess) {
      return true;
   }
   mmh_set_error("%s: %s", what, hipGetErrorString(e));
   return false;
}

#define HIP_TRY(expr)                       \
   do {                                     \
      if (!hip_ok((expr), #expr)) {         \
         return MMH_E_DEVICE;               \
      }                                     \
   } while (0)

} // namespace

// MMOORE_SYNC_TRACE=1 (development): where a synchronous scan's host time goes -- marks taken along mmh_scan, averages
// printed every 256 scans: [0] entry -> launches done, [1] -> flag seen (device time + launch latency), [2] -> block
// validated, [3] -> back at the caller
namespace {
struct SyncTrace {
   bool on = getenv("MMOORE_SYNC_TRACE") != nullptr;
   std::chrono::steady_clock::time_point t[5];
   double sum[4] = {0, 0, 0, 0};
   uint64_t n = 0;
   void mark(int k) { if (on) { t[k] = std::chrono::steady_clock::now(); } }
   void done()
   {
      if (!on) {
         return;
      }
      for (int k = 0; k < 4; k++) {
         sum[k] += std::chrono::duration<double>(t[k + 1] - t[k]).count() * 1e6;
      }
      if (++n % 256 == 0) {
         fprintf(stderr, "mmh_scan host trace, mean of 256 scans: launches %.1f us, until the flag %.1f us, validation %.1f us, hand-over %.1f us\n",
                 sum[0] / 256, sum[1] / 256, sum[2] / 256, sum[3] / 256);
         sum[0] = sum[1] = sum[2] = sum[3] = 0;
      }
   }
};
thread_local SyncTrace g_sync_trace;
} // namespace

extern "C" void mmh_set_error(const char *fmt, ...)
{
   char buf[512];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof(buf), fmt, ap);
   va_end(ap);
   g_error = buf;
}

extern "C" const char *mmh_last_error(void) { return g_error.c_str(); }

namespace {

mm::ResolveBuffers resolve_buffers(const MmWorkspace &w)
{
   mm::ResolveBuffers rb;
   rb.cand = w.d_cand; rb.cand_cap = w.cand_cap; rb.out = w.d_out; rb.out_cap = w.out_cap; rb.ctrl = w.d_ctrl;
   rb.mid_off = w.d_mid_off; rb.mid_hi = w.d_mid_hi; rb.mid_set = w.d_mid_set; rb.mid_slot = w.d_mid_slot;
   rb.hard_off = w.d_hard_off; rb.hard_hi = w.d_hard_hi; rb.hard_set = w.d_hard_set; rb.hard_slot = w.d_hard_slot;
   rb.scratch = w.d_scratch;
   rb.bcand = w.d_bcand; rb.bcount = w.d_bcount;
   return rb;
}

// The bucketed candidate store of a workspace, counters zeroed: MM_BUCKET_CAP slots for each of the buckets THIS ROM is
// cut into (mm::bucket_geom: <= 4096 buckets, 128 MiB for ROMs of >= 16 MiB; a 64 KiB ROM's 17 buckets take 544 KiB),
// grown when a larger ROM arrives.  The counters are always there for all MM_MAX_BUCKETS (16 KiB: mm_scan_tail2 sums
// them all).  (Until round 4 every workspace that met a ROM beyond the single-launch kernel's took the full 128 MiB.)
// (clear = false, prepare_scans: the store for ANY ROM or part of one -- a part of a big ROM is cut into more buckets than
// the ROM itself, 3073 for three eighths of 4 GiB against 2049 -- and no memset on a stream the scan does not run on)
int ensure_buckets(mmh_ctx *c, MmWorkspace &w, hipStream_t st, uint64_t rom_bytes, bool clear = true)
{
   const uint64_t need = clear ? mm::bucket_geom(rom_bytes).nb : (uint64_t)MM_MAX_BUCKETS;
   if (!w.d_bcount) {
      HIP_TRY(hipSetDevice(c->device));
      HIP_TRY(hipMalloc(&w.d_bcount, mm::bucket_count_bytes()));
      w.buckets_clean = false;
   }
   if (need > w.bcand_buckets) {
      HIP_TRY(hipSetDevice(c->device));
      if (w.d_bcand) {
         // (nothing of this workspace is in flight: a workspace runs one scan at a time and the previous one was waited for)
         HIP_TRY(hipFree(w.d_bcand));
         w.d_bcand = nullptr;
         w.bcand_buckets = 0;
      }
      const uint64_t want = std::min<uint64_t>(MM_MAX_BUCKETS, need + need / 4 + 1);
      HIP_TRY(hipMalloc(&w.d_bcand, want * MM_BUCKET_CAP * sizeof(uint64_t)));
      w.bcand_buckets = want;
   }
   if (clear && !w.buckets_clean) {
      HIP_TRY(hipMemsetAsync(w.d_bcount, 0, mm::bucket_count_bytes(), st));
   }
   return MMH_OK;
}

int ensure_workspace(mmh_ctx *c, MmWorkspace &w, uint64_t out_cap)
{
   HIP_TRY(hipSetDevice(c->device));
   if (!w.d_cand) {
      w.cand_cap = kInitialCap;
      HIP_TRY(hipMalloc(&w.d_cand, w.cand_cap * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_ctrl, mm::ctrl_bytes()));
      HIP_TRY(hipMalloc(&w.d_mid_off, mm::mid_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_mid_hi, mm::mid_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_mid_set, mm::mid_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_mid_slot, mm::mid_cap() * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(&w.d_hard_off, mm::hard_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_hard_hi, mm::hard_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_hard_set, mm::hard_cap() * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&w.d_hard_slot, mm::hard_cap() * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(&w.d_scratch, mm::hard_scratch_bytes()));
      HIP_TRY(hipMalloc(&w.d_partials, mm::rank_partials_bytes(kMaxRankSort)));
      HIP_TRY(hipHostMalloc(&w.h_result, MM_RESULT_BLOCK_WORDS * sizeof(uint64_t), hipHostMallocDefault));
      std::memset(w.h_result, 0, kHeaderWords * sizeof(uint64_t));
      std::memset(w.h_result + kHeaderWords, 0xFE, (size_t)MM_MAX_PUBLISH * sizeof(uint64_t));   // MM_SLOT_POISON in every slot
      std::memset(w.h_result + MM_HDR_FLAG_WORD, 0, (MM_RESULT_BLOCK_WORDS - MM_HDR_FLAG_WORD) * sizeof(uint64_t));
      w.dirty_slots = 0;
      for (auto &d : w.d_result) {
         HIP_TRY(hipMalloc(&d, MM_RESULT_BLOCK_WORDS * sizeof(uint64_t)));
         // (the words behind the slots: mm_publish_list's arrival counter lives there, zero between uses)
         HIP_TRY(hipMemset(d + MM_HDR_FLAG_WORD, 0, (MM_RESULT_BLOCK_WORDS - MM_HDR_FLAG_WORD) * sizeof(uint64_t)));
      }
      w.ctrl_clean = false;
   }
   if (out_cap > w.out_cap) {
      if (w.d_out) {
         HIP_TRY(hipFree(w.d_out));
         w.d_out = nullptr;
      }
      HIP_TRY(hipMalloc(&w.d_out, out_cap * sizeof(uint64_t)));
      w.out_cap = out_cap;
   }
   for (auto &triple : c->ring) {
      for (auto &e : triple) {
         if (!e) {
            HIP_TRY(hipEventCreate(&e));
         }
      }
   }
   return MMH_OK;
}

void free_workspace(MmWorkspace &w)
{
   if (w.d_cand) (void)hipFree(w.d_cand);
   if (w.d_out) (void)hipFree(w.d_out);
   if (w.d_ctrl) (void)hipFree(w.d_ctrl);
   if (w.d_mid_off) (void)hipFree(w.d_mid_off);
   if (w.d_mid_hi) (void)hipFree(w.d_mid_hi);
   if (w.d_mid_set) (void)hipFree(w.d_mid_set);
   if (w.d_mid_slot) (void)hipFree(w.d_mid_slot);
   if (w.d_hard_off) (void)hipFree(w.d_hard_off);
   if (w.d_hard_hi) (void)hipFree(w.d_hard_hi);
   if (w.d_hard_set) (void)hipFree(w.d_hard_set);
   if (w.d_hard_slot) (void)hipFree(w.d_hard_slot);
   if (w.d_scratch) (void)hipFree(w.d_scratch);
   if (w.d_partials) (void)hipFree(w.d_partials);
   if (w.d_bcand) (void)hipFree(w.d_bcand);
   if (w.d_bcount) (void)hipFree(w.d_bcount);
   if (w.h_result) (void)hipHostFree(w.h_result);
   for (auto d : w.d_result) {
      if (d) (void)hipFree(d);
   }
   w = MmWorkspace();
}

} // namespace

int mmh_workspace(mmh_ctx *c) { return ensure_workspace(c, c->ws[0], std::max<uint64_t>(c->ws[0].out_cap, kInitialCap)); }

namespace {
int grow(uint64_t **buf, uint64_t *cap, uint64_t need);
int ensure_lane(mmh_ctx *c, int lane);
int ensure_fetch_ring(mmh_ctx *c);
int ensure_sort_temp(mmh_ctx *c, uint64_t n);

// What the FIRST scan of a ROM would otherwise set up inside its own call (round 6: a ROM hacker scans a keyword once --
// the first scan is the product; it cost 1.5 to 15 times a later one).  Every entry point that gives the context a ROM
// ends here: the synchronous workspace; for a ROM in HBM its bucket store; for a ROM the split pipeline takes (>= 1 GiB)
// the three lanes -- streams, events, workspaces, bucket stores --, result slots for the forward engine's lists at one match
// in 128 positions (it ran twice when they overflowed: 36 ms for a first `aaaa`), the ordering buffers and the forward
// engine's maps.  A 4 GiB ROM: ~1 GiB of the 288, once per context.  Nothing here is a memo: no scan leaves anything
// behind that a later scan's route depends on.
int prepare_scans(mmh_ctx *c)
{
   HIP_TRY(hipSetDevice(c->device));
   const bool in_hbm = c->rom && c->rom != c->rom_host && c->rom_bytes != 0;
   const bool big = in_hbm && c->rom_bytes >= kSplitMinBytes;
   uint64_t out_cap = kInitialCap;
   if (big) {
      out_cap = std::min<uint64_t>(std::max<uint64_t>(c->rom_bytes / 128, kInitialCap), 1ull << 25);
   }
   int rc = ensure_workspace(c, c->ws[0], std::max<uint64_t>(c->ws[0].out_cap, out_cap));
   if (rc != MMH_OK || !in_hbm) {
      return rc;
   }
   if (c->rom_bytes > (4ull << 20)) {                      // (beyond the single-launch kernel's ROMs)
      rc = ensure_buckets(c, c->ws[0], c->stream, c->rom_bytes, false);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   if (!big) {
      return MMH_OK;
   }
   for (int lane = 0; lane < mmh_ctx::kLanes; lane++) {
      rc = ensure_lane(c, lane);
      if (rc == MMH_OK) {
         rc = ensure_buckets(c, c->ws[1 + lane], c->stream, c->rom_bytes, false);
      }
      if (rc != MMH_OK) {
         return rc;
      }
   }
   // A stream gets its hardware queue at its first submission and an event its signal at its first record (0.1 - 0.2 ms
   // each, inside a first scan): one fill of a control word per lane stream, every lane event recorded once, now.
   for (int lane = 0; lane < mmh_ctx::kLanes; lane++) {
      const hipStream_t st = c->lane_stream[lane % 2];
      HIP_TRY(hipMemsetAsync(c->ws[1 + lane].d_ctrl, 0, sizeof(unsigned long long), st));
      for (auto &e : c->lane_ev[lane]) {
         HIP_TRY(hipEventRecord(e, st));
      }
   }
   for (auto &triple : c->ring) {
      for (auto &e : triple) {
         HIP_TRY(hipEventRecord(e, c->stream));
      }
   }
   HIP_TRY(hipEventRecord(c->lane_fence, c->stream));
   for (int k = 0; k < 2; k++) {
      HIP_TRY(hipStreamSynchronize(c->lane_stream[k]));
   }
   HIP_TRY(hipStreamSynchronize(c->stream));
   rc = grow(&c->d_sort_in, &c->sort_in_cap, c->ws[0].out_cap);
   if (rc == MMH_OK) {
      rc = grow(&c->d_sort_out, &c->sort_out_cap, c->ws[0].out_cap);
   }
   if (rc == MMH_OK) {
      rc = ensure_sort_temp(c, c->ws[0].out_cap);
   }
   if (rc == MMH_OK) {
      rc = ensure_fetch_ring(c);
   }
   if (rc != MMH_OK) {
      return rc;
   }
   // the forward engine's maps: a look-back word and up to MMH_MAX_KEYWORD bytes of map per 32 Ki elements of every domain
   const size_t dense = (size_t)(c->rom_bytes / 128 + (4u << 20));
   if (dense > c->dense_bytes) {
      if (c->d_dense) {
         HIP_TRY(hipFree(c->d_dense));
         c->d_dense = nullptr;
         c->dense_bytes = 0;
      }
      HIP_TRY(hipMalloc(&c->d_dense, dense));
      c->dense_bytes = dense;
   }
   return MMH_OK;
}
} // namespace

namespace {

// next slot of the event ring
void begin_scan_events(mmh_ctx *c, bool has_filter)
{
   const int slot = (int)(c->scans_recorded % mmh_ctx::kRing);
   c->ev = c->ring[slot];
   c->ring_has_filter[slot] = has_filter;
   c->ring_is_ms[slot] = false;
   c->ring_timed[slot] = c->timing;
   c->ring_parts[slot] = 0;
}

void release_rom(mmh_ctx *c)
{
   (void)mm_ingest_drain(c);                // (copies of an aborted file load may still be writing the ROM)
   if (c->rom_own) {
      (void)hipFree(c->rom_own);
   }
   c->rom = nullptr;
   c->rom_own = nullptr;
   c->rom_bytes = 0;
   c->rom_alloc = 0;
}

// Uploads of up to this many bytes are not copied to HBM at all: they are copied into a pinned host
// buffer and the kernels read that over PCIe (a 128 KiB hipMemcpy from pageable memory costs ~20 us
// of a ~45 us MonkeyMoore<T>::search; a CPU memcpy of it 4 us and the scan reads it once).
// routes the first-use self-test found returning wrong results on this machine: off for the whole process
std::atomic<uint32_t> g_routes_off{0};

uint32_t routes_off(const mmh_ctx *c) { return c->route_off | g_routes_off.load(std::memory_order_relaxed); }

uint64_t zero_copy_limit()
{
   static const uint64_t v = [] {
      const char *e = getenv("MMOORE_ZEROCOPY_MAX_KIB");
      return (uint64_t)(e && *e ? atol(e) : 512) << 10;
   }();
   return v;
}

} // namespace

extern "C" int mmh_device_count(int *count)
{
   if (!count) {
      mmh_set_error("mmh_device_count: null argument");
      return MMH_E_ARG;
   }
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess) {
      *count = 0;
      mmh_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
      return MMH_E_DEVICE;
   }
   *count = n;
   return MMH_OK;
}

namespace {

int create_context(int device, mmh_ctx **out)
{
   *out = nullptr;
   int n = 0;
   if (mmh_device_count(&n) != MMH_OK || n == 0) {
      if (n == 0) {
         mmh_set_error("no HIP device available: the MI355X engine has no CPU fallback");
      }
      return MMH_E_DEVICE;
   }
   if (device < 0 || device >= n) {
      mmh_set_error("mmh_create: device %d out of range (have %d)", device, n);
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(device));
   mmh_ctx *c = new mmh_ctx();
   c->device = device;
   if (!hip_ok(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate")) {
      delete c;
      return MMH_E_DEVICE;
   }
   c->stream = c->own_stream;
   *out = c;
   return MMH_OK;
}

// first-use self-test, once per device and process (mmh_selftest_run at the end of this file)
constexpr int kSelftestDevices = 64;
std::once_flag g_selftest_once[kSelftestDevices];
std::atomic<int> g_selftest_state[kSelftestDevices];        // 0 not run, 1 passed, 2 passed with routes off, 3 failed
std::atomic<uint32_t> g_selftest_off[kSelftestDevices];
std::string g_selftest_error[kSelftestDevices];

bool selftest_enabled()
{
   static const bool on = [] {
      const char *v = getenv("MMOORE_SELFTEST");
      return !(v && *v == '0');
   }();
   return on;
}

} // namespace

namespace {
int sort_on_device(mmh_ctx *c, const uint64_t *keys, uint64_t n);
int fetch_device_list(mmh_ctx *c, const uint64_t *d_list, uint64_t n, uint64_t *dst, uint64_t cap, uint64_t *matches);
std::once_flag g_warm_once[kSelftestDevices];

// Once per device and process, at the first mmh_create: what the HIP runtime does lazily at a first use, done before any
// caller's scan is on the clock (round 6, tools/first_scan_fresh.py: a first scan cost 1.5 to 16 times a later one; the
// API trace, profiles/r06_first_scan_api_trace.txt, shows where: the code object of a translation unit is loaded at the
// first launch of one of its kernels -- 1.5 to 3 ms per unit of streaming kernels, 23 ms for rocPRIM's radix sort --, the
// first asynchronous device-to-host copy sets up its engine in 9 ms, the first streams take 3 to 9 ms each).
// A scratch context scans an 8 MiB synthetic ROM through every engine and element width and sorts a list; every other
// unit's kernels are looked up once (which loads their code objects).  Failures here are not errors: a caller's scan
// then simply pays what this would have paid.
void warm_device(int device)
{
   mmh_ctx *t = nullptr;
   if (create_context(device, &t) != MMH_OK) {
      return;
   }
   mm::preload_kernels();
   const uint64_t n = 8ull << 20;
   uint32_t kw[65];
   for (int i = 0; i < 65; i++) {
      kw[i] = (uint32_t)('a' + (i * 7) % 26);
   }
   std::vector<uint64_t> out(4096);
   uint64_t count = 0;
   if (mmh_rom_alloc(t, n) == MMH_OK && mmh_rom_synth(t, 0x6d6d6f6f7265ull, 0) == MMH_OK) {
      for (int engine : {0, 2}) {
         (void)mmh_set_engine(t, engine);
         for (uint32_t elem : {1u, 2u}) {
            for (uint32_t len : {12u, 65u}) {
               mmh_plan_desc plan;
               if (mmh_plan_relative(elem, kw, len, 0, nullptr, 0, &plan) == MMH_OK) {
                  (void)mmh_scan(t, &plan, 524288, 0, 0, out.data(), out.size(), &count);
               }
            }
         }
      }
      (void)mmh_set_engine(t, 0);
      // three scans in flight: the lanes' streams
      mmh_plan_desc plan;
      int tickets[mmh_ctx::kLanes];
      if (mmh_plan_relative(1, kw, 12, 0, nullptr, 0, &plan) == MMH_OK) {
         int have = 0;
         for (; have < mmh_ctx::kLanes && mmh_scan_submit(t, &plan, 524288, 0, 0, &tickets[have]) == MMH_OK; have++) {
         }
         for (int k = 0; k < have; k++) {
            (void)mmh_scan_collect(t, tickets[k], out.data(), out.size(), &count);
         }
      }
      // long lists: both of the radix sort's regimes, the pinned pieces, an asynchronous copy to the host
      if (grow(&t->d_sort_in, &t->sort_in_cap, 1u << 20) == MMH_OK &&
          hipMemsetAsync(t->d_sort_in, 0x5a, (1u << 20) * sizeof(uint64_t), t->stream) == hipSuccess) {
         uint64_t kept = 0;
         for (uint64_t keys : {4096ull, 1ull << 20}) {
            if (sort_on_device(t, t->d_sort_in, keys) == MMH_OK) {
               (void)fetch_device_list(t, t->d_sort_out, std::min<uint64_t>(keys, 65536), out.data(), out.size(), &kept);
            }
         }
      }
      (void)hipStreamSynchronize(t->stream);
   }
   (void)hipGetLastError();
   mmh_destroy(t);
}
} // namespace

extern "C" int mmh_create(int device, mmh_ctx **out)
{
   if (!out) {
      mmh_set_error("mmh_create: null argument");
      return MMH_E_ARG;
   }
   const int rc = create_context(device, out);
   if (rc != MMH_OK) {
      return rc;
   }
   if (device < kSelftestDevices) {
      std::call_once(g_warm_once[device], [device] { warm_device(device); });
   }
   if (selftest_enabled() && device < kSelftestDevices) {
      std::call_once(g_selftest_once[device], [device] {
         uint32_t off = 0;
         const int st = mmh_selftest_run(device, &off);
         g_selftest_off[device] = off;
         if (st != MMH_OK) {
            g_selftest_error[device] = mmh_last_error();
            g_selftest_state[device] = 3;
            fprintf(stderr, "libmmoore_hip: SELF-TEST FAILED on device %d: %s\n", device, g_selftest_error[device].c_str());
         }
         else if (off) {
            g_routes_off.fetch_or(off);
            g_selftest_state[device] = 2;
            fprintf(stderr, "libmmoore_hip: self-test on device %d: a fast route returned wrong results; routes 0x%x are switched off for "
                            "this process (MMH_ROUTE_*: 1 single-launch kernel, 2 zero-copy upload, 4 bucketed store, 8 polled results)\n",
                    device, off);
         }
         else {
            g_selftest_state[device] = 1;
         }
      });
      if (g_selftest_state[device] == 3) {
         mmh_destroy(*out);
         *out = nullptr;
         mmh_set_error("mmh_create: the device fails the known-answer self-test even on the plain kernels (%s)", g_selftest_error[device].c_str());
         return MMH_E_DEVICE;
      }
   }
   return MMH_OK;
}

extern "C" void mmh_destroy(mmh_ctx *c)
{
   if (!c) {
      return;
   }
   (void)hipSetDevice(c->device);
   (void)hipStreamSynchronize(c->stream);
   mmh_comm_destroy(c);
   release_rom(c);
   for (auto &p : c->pending) {
      p.active = false;
   }
   for (hipStream_t t : c->lane_stream) {
      if (t) {
         (void)hipStreamSynchronize(t);
         (void)hipStreamDestroy(t);
      }
   }
   if (c->lane_fence) (void)hipEventDestroy(c->lane_fence);
   for (auto &triple : c->lane_ev) {
      for (auto &e : triple) {
         if (e) (void)hipEventDestroy(e);
      }
   }
   for (auto &w : c->ws) {
      free_workspace(w);
   }
   if (c->rom_host) (void)hipHostFree(c->rom_host);
   if (c->d_dense) (void)hipFree(c->d_dense);
   if (c->d_sort_in) (void)hipFree(c->d_sort_in);
   if (c->d_domains) (void)hipFree(c->d_domains);
   if (c->d_sort_out) (void)hipFree(c->d_sort_out);
   if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
   for (int k = 0; k < 2; k++) {
      if (c->h_ring[k]) (void)hipHostFree(c->h_ring[k]);
      if (c->ring_ev[k]) (void)hipEventDestroy(c->ring_ev[k]);
   }
   for (auto &triple : c->ring) {
      for (auto &e : triple) {
         if (e) (void)hipEventDestroy(e);
      }
   }
   for (void *p : c->ingest.staging) (void)hipHostFree(p);
   for (hipEvent_t e : c->ingest.events) (void)hipEventDestroy(e);
   for (hipStream_t t : c->ingest.streams) (void)hipStreamDestroy(t);
   if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
   delete c;
}

extern "C" int mmh_set_stream(mmh_ctx *c, void *hip_stream)
{
   if (!c) {
      mmh_set_error("mmh_set_stream: null context");
      return MMH_E_ARG;
   }
   c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
   return MMH_OK;
}

extern "C" int mmh_set_engine(mmh_ctx *c, int engine)
{
   if (!c || engine < 0 || engine > 2) {
      mmh_set_error("mmh_set_engine: bad argument");
      return MMH_E_ARG;
   }
   c->engine = engine;
   return MMH_OK;
}

extern "C" int mmh_set_timing(mmh_ctx *c, int on)
{
   if (!c) {
      mmh_set_error("mmh_set_timing: bad argument");
      return MMH_E_ARG;
   }
   c->timing = on != 0;
   return MMH_OK;
}

extern "C" int mmh_rom_alloc(mmh_ctx *c, uint64_t nbytes)
{
   if (!c) {
      mmh_set_error("mmh_rom_alloc: null context");
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(c->device));
   {
      const int rc = mm_ingest_drain(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   uint64_t need = ((nbytes + 15) / 16) * 16 + 16;
   if (!(c->rom_own && c->rom_alloc >= need)) {
      release_rom(c);
      void *p = nullptr;
      HIP_TRY(hipMalloc(&p, need));
      c->rom_own = static_cast<uint8_t *>(p);
      c->rom_alloc = need;
   }
   c->rom = c->rom_own;
   c->rom_bytes = nbytes;
   // the padding behind the ROM is never interpreted, but keep it defined
   HIP_TRY(hipMemsetAsync(c->rom + (nbytes / 16) * 16, 0, need - (nbytes / 16) * 16, c->stream));
   return prepare_scans(c);
}

extern "C" int mmh_rom_upload(mmh_ctx *c, const void *host, uint64_t nbytes)
{
   if (!c || (!host && nbytes)) {
      mmh_set_error("mmh_rom_upload: bad argument");
      return MMH_E_ARG;
   }
   {
      // (copies of an aborted file load may still be on their way into the old ROM, or into the staging they came from)
      const int rc = mm_ingest_drain(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   if (nbytes && nbytes <= zero_copy_limit() && !(routes_off(c) & MMH_ROUTE_NO_ZERO_COPY)) {
      HIP_TRY(hipSetDevice(c->device));
      if (!c->rom_host) {
         void *p = nullptr;
         HIP_TRY(hipHostMalloc(&p, zero_copy_limit() + 64, hipHostMallocDefault));
         c->rom_host = static_cast<uint8_t *>(p);
      }
      // (nothing of an earlier scan still reads the buffer: scans are synchronous, lanes are collected before the ROM may change)
      std::memcpy(c->rom_host, host, nbytes);
      std::memset(c->rom_host + nbytes, 0, 32 + (16 - nbytes % 16) % 16);     // the padding behind the ROM stays defined
      c->rom = c->rom_host;
      c->rom_bytes = nbytes;
      return prepare_scans(c);
   }
   int rc = mmh_rom_alloc(c, nbytes);
   if (rc != MMH_OK) {
      return rc;
   }
   if (nbytes) {
      HIP_TRY(hipMemcpyAsync(c->rom, host, nbytes, hipMemcpyHostToDevice, c->stream));
   }
   HIP_TRY(hipStreamSynchronize(c->stream));
   return MMH_OK;
}

extern "C" int mmh_rom_attach(mmh_ctx *c, const void *device_ptr, uint64_t nbytes)
{
   if (!c || (!device_ptr && nbytes)) {
      mmh_set_error("mmh_rom_attach: bad argument");
      return MMH_E_ARG;
   }
   if (reinterpret_cast<uintptr_t>(device_ptr) & 15) {
      mmh_set_error("mmh_rom_attach: device pointer must be 16-byte aligned");
      return MMH_E_ARG;
   }
   release_rom(c);
   c->rom = const_cast<uint8_t *>(static_cast<const uint8_t *>(device_ptr));
   c->rom_bytes = nbytes;
   return prepare_scans(c);
}

extern "C" int mmh_rom_download(mmh_ctx *c, uint64_t first_byte, void *host, uint64_t nbytes)
{
   if (!c || !host || !c->rom || first_byte + nbytes > c->rom_bytes) {
      mmh_set_error("mmh_rom_download: bad argument");
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(c->device));
   HIP_TRY(hipMemcpyAsync(host, c->rom + first_byte, nbytes, hipMemcpyDefault, c->stream));   // (the ROM may be the pinned host buffer)
   HIP_TRY(hipStreamSynchronize(c->stream));
   return MMH_OK;
}

extern "C" int mmh_rom_synth(mmh_ctx *c, uint64_t seed, uint64_t rom_base_offset)
{
   if (!c || !c->rom) {
      mmh_set_error("mmh_rom_synth: no ROM");
      return MMH_E_STATE;
   }
   if (rom_base_offset & 15) {
      mmh_set_error("mmh_rom_synth: base offset must be a multiple of 16");
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(c->device));
   {
      const int rc = mm_ingest_drain(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   mm::launch_synth(c->stream, c->rom, c->rom_bytes, seed, rom_base_offset);
   HIP_TRY(hipGetLastError());
   return MMH_OK;
}

extern "C" int mmh_rom_poke(mmh_ctx *c, uint64_t first_byte, const void *host, uint64_t nbytes)
{
   if (!c || !c->rom || !host || first_byte + nbytes > c->rom_bytes) {
      mmh_set_error("mmh_rom_poke: bad argument");
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(c->device));
   {
      const int rc = mm_ingest_drain(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   HIP_TRY(hipMemcpyAsync(c->rom + first_byte, host, nbytes, hipMemcpyDefault, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   return MMH_OK;
}

extern "C" int mmh_rom_fill(mmh_ctx *c, uint64_t first_byte, uint64_t nbytes, int value, int ramp)
{
   if (!c || !c->rom || first_byte + nbytes > c->rom_bytes) {
      mmh_set_error("mmh_rom_fill: bad argument");
      return MMH_E_ARG;
   }
   HIP_TRY(hipSetDevice(c->device));
   {
      const int rc = mm_ingest_drain(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   mm::launch_pattern_fill(c->stream, c->rom, first_byte, nbytes, value, ramp);
   HIP_TRY(hipGetLastError());
   return MMH_OK;
}

namespace {

struct Outcome {
   uint64_t candidates = 0;         // filter survivors (0 on the sequential path)
   uint64_t listed = 0;             // keys in d_out: result slots (fast path) or appended matches (sequential)
   uint64_t matches = 0;            // valid when sorted_on_device
   uint64_t tiles = 0;
   uint32_t hard = 0;
   bool hard_overflow = false;
   bool sorted_on_device = false;
   bool bucket_overflow = false;    // (tickets of scan_split only) the bucketed store overflowed and nothing was run again
   uint32_t limit = 0;              // the candidate limit of the kernels that ran (bucketed store: 2^20, list-based kernels: 2^18)
};

// enqueue [zero counters] -> engine kernels -> ordering into pinned host memory; `ev` = the
// scan's event triple {start, behind the streaming kernel, end}
// One fused scan kernel at a time per process: its grid barrier needs all of its workgroups
// resident, and two such grids in flight could keep each other's stragglers out (mm_fused.h).
// A scan that finds the lock taken simply runs the plain kernels.
std::mutex g_fused_lock;

// MMOORE_BUCKETS=0: big ROMs take mm_scan_tail over the 64 candidate lists (round 2's tail) instead of the bucketed
// store + mm_scan_tail2
bool buckets_enabled()
{
   static const bool on = [] {
      const char *v = getenv("MMOORE_BUCKETS");
      return !(v && *v == '0');
   }();
   return on;
}

bool fused_enabled()
{
   static const bool on = [] {
      const char *v = getenv("MMOORE_FUSED");
      return !(v && *v == '0');
   }();
   return on;
}

// slots of the pinned block that earlier scans wrote go back to the poison before the next launch (see MM_SLOT_POISON)
void poison_dirty_slots(MmWorkspace &w)
{
   if (w.dirty_slots) {
      std::memset(w.h_result + kHeaderWords, 0xFE, (size_t)std::min<uint64_t>(w.dirty_slots, MM_MAX_PUBLISH) * sizeof(uint64_t));
      w.dirty_slots = 0;
   }
}

// (only slots a kernel stores straight into pinned memory can be mistaken: lists beyond that arrive by a copy that
// overwrites every slot it announces)
void note_dirty_slots(MmWorkspace &w, uint64_t n)
{
   w.dirty_slots = std::max<uint64_t>(w.dirty_slots, std::min<uint64_t>(n, kMaxRankSort));
}

// An outstanding gather may still be sending the device-side result copy a pipeline is about to publish into
// (mmh_gather_start(NULL, 0) sends a scan's list from there, and overlaps the scans that follow): wait for its
// collective.  Called for every pipeline launch -- a scan that retries (out_cap grown, left-overs to the flagged-
// domains or flood path) toggles the copies again and would otherwise overwrite the one still being sent.
int wait_for_gather_reading(mmh_ctx *c, const uint64_t *buffer)
{
   for (auto &s : c->mg.slot) {
      if (s.busy && !s.from_host && s.src && s.src == buffer) {
         HIP_TRY(hipSetDevice(c->device));
         HIP_TRY(hipEventSynchronize(s.end));
      }
   }
   return MMH_OK;
}

uint32_t list_candidate_limit(const MmWorkspace &w, uint32_t max_candidates)
{
   return (uint32_t)std::min<uint64_t>(std::min<uint32_t>(max_candidates, mm::tuning().list_candidates), w.cand_cap / 2);
}

int enqueue_pipeline(mmh_ctx *c, MmWorkspace &w, hipStream_t st, hipEvent_t *ev, const MmGeom &g, const mmh_plan_desc &pl,
                     const mm::FilterChoice &fc, bool sequential, uint64_t base_offset, uint32_t max_candidates,
                     const uint32_t *skip_bits = nullptr, bool allow_polled = false, bool allow_single_launch = false,
                     hipStream_t tail_st = nullptr, unsigned tail_blocks = 0)
{
   const int count_index = sequential ? 1 : 0;
   const uint32_t off = routes_off(c);
   // (the event at a scan's start costs its first dispatch ~4.5 us: only for callers that ask for timings, mmh_set_timing)
   hipEvent_t const ev_start = c->timing ? ev[0] : nullptr;
   const bool polled = allow_polled && !sequential && !skip_bits && fused_enabled() && !(off & MMH_ROUTE_NO_POLLED);
   const bool single_launch = polled && allow_single_launch && c->fused_ok && !(off & MMH_ROUTE_NO_SINGLE_LAUNCH) && mm::fused_applies(g);
   const bool bucketed = polled && !single_launch && buckets_enabled() && !(off & MMH_ROUTE_NO_BUCKETS);
   // Only the bucketed store takes the full limit: the list-based kernels keep round 2's (their lists share d_cand, and the
   // callers read "more candidates than this" as "a flood: take it apart domain by domain").
   if (!bucketed) {
      max_candidates = list_candidate_limit(w, max_candidates);
   }
   w.limit = max_candidates;
   if (bucketed) {
      const int rc = ensure_buckets(c, w, st, g.nbytes);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   w.bucketed = false;
   const mm::ResolveBuffers rb = resolve_buffers(w);

   if (!bucketed) {
      poison_dirty_slots(w);                   // (bucketed: behind the streaming kernel's launch, below -- up to 128 KiB of memset)
   }
   w.max_rank = bucketed ? MM_MAX_PUBLISH : kMaxRankSort;
   w.h_result[6] = 0;                          // mm_rank_scatter publishes "matches + 1" here
   w.result_turn ^= 1;                         // the other device-side copy may still be feeding a gather ...
   {
      const int rc = wait_for_gather_reading(c, w.d_result[w.result_turn]);   // ... and so may this one (two gathers outstanding, retries)
      if (rc != MMH_OK) {
         return rc;
      }
   }
   if (!w.ctrl_clean) {
      HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, mm::ctrl_bytes(), st));
   }
   w.ctrl_clean = false;
   w.fused = false;
   w.polled = false;
   if (polled) {
      // the scan's end is announced in pinned memory (finish_pipeline polls): either everything in one
      // launch (small ROMs), or the streaming kernel + ONE tail kernel
      if (single_launch && g_fused_lock.try_lock()) {
         w.seq++;
         if (mm::launch_fused(st, g, pl, fc, rb, base_offset, max_candidates, w.h_result, w.d_result[w.result_turn], kMaxRankSort,
                              w.seq, ev_start, ev[2])) {
            if (!hip_ok(hipGetLastError(), "launching the fused scan kernel")) {
               g_fused_lock.unlock();
               return MMH_E_DEVICE;
            }
            w.fused = true;
            w.polled = true;
            return MMH_OK;
         }
         g_fused_lock.unlock();
         c->fused_ok = false;                   // the occupancy query failed: never try again
      }
      w.seq++;
      if (bucketed) {
         // big ROMs: candidates into buckets of their ROM neighbourhood, mm_scan_tail2 behind (mm_tail2.h).  tail_st:
         // the tail kernel goes to a stream of its own, behind the streaming kernel's end event (scans in flight)
         w.buckets_clean = false;                 // (until the tail kernel has been seen to finish: it zeroes the counters)
         w.bucketed = true;
         mm::launch_filter_buckets(st, g, pl, fc, rb, ev_start, ev[1]);
         poison_dirty_slots(w);                   // (only the tail kernel stores into the pinned block: the device streams meanwhile)
         if (tail_st && tail_st != st) {
            HIP_TRY(hipStreamWaitEvent(tail_st, ev[1], 0));
         }
         mm::launch_tail2(tail_st ? tail_st : st, g, pl, fc, rb, base_offset, max_candidates, w.h_result, w.d_result[w.result_turn], w.seq,
                          ev[2], tail_blocks);
      }
      else {
         mm::launch_filter(st, g, pl, fc, w.d_cand, w.d_ctrl, w.cand_cap, ev_start, ev[1], nullptr, nullptr);
         mm::launch_tail(st, g, pl, fc, rb, base_offset, max_candidates, w.h_result, w.d_result[w.result_turn], kMaxRankSort, w.seq,
                         ev[2]);
      }
      HIP_TRY(hipGetLastError());
      w.polled = true;
      return MMH_OK;
   }
   // The scan's three events ride on kernel dispatches (hipExtLaunchKernelGGL) where they can:
   // a hipEventRecord between dependent kernels costs ~6 us of stream time on this stack.
   if (!sequential) {
      mm::launch_filter(st, g, pl, fc, w.d_cand, w.d_ctrl, w.cand_cap, ev_start, ev[1], nullptr, skip_bits);
      mm::launch_resolve(st, g, pl, rb, base_offset, max_candidates);
   }
   else {
      if (ev_start) {
         HIP_TRY(hipEventRecord(ev_start, st));
      }
      HIP_TRY(hipEventRecord(ev[1], st));
      mm::launch_chain_seq(st, g, pl, w.d_out, w.d_ctrl + 1, w.out_cap, base_offset);
   }
   // (first phase: when mm_resolve leaves candidates over, the ordering kernel keeps the control
   // block for the second phase, see finish_pipeline)
   mm::launch_rank_sort(st, w.d_out, w.d_ctrl, count_index, w.out_cap, kMaxRankSort, w.d_partials, w.h_result,
                        w.d_result[w.result_turn], ev[2], !sequential);
   HIP_TRY(hipGetLastError());
   return MMH_OK;
}

void read_outcome(MmWorkspace &w, bool sequential, Outcome *oc)
{
   const int count_index = sequential ? 1 : 0;
   oc->candidates = w.h_result[0];
   oc->listed = w.h_result[count_index];
   note_dirty_slots(w, oc->listed);            // (the rank kernels' list: up to kMaxRankSort slots)
   oc->tiles = w.h_result[2];
   oc->hard = (uint32_t)(w.h_result[3] & 0xFFFFFFFFu);
   // left-overs beyond what mm_resolve2 / mm_hard_resolve take, or a prefix too long for the latter
   oc->hard_overflow = (w.h_result[3] >> 32) != 0 || (w.h_result[5] & 0xFFFFFFFFu) > mm::mid_cap();
   oc->sorted_on_device = oc->listed <= kMaxRankSort && oc->listed <= w.out_cap;
   oc->matches = w.h_result[6] ? w.h_result[6] - 1 : oc->listed;
}

// ---- what a polled scan published, checked before it is trusted (include/mmoore_hip.h, "route health") -------------

void note_violation(mmh_ctx *c, uint64_t reason, const MmWorkspace &w, const char *what)
{
   MmHealth &h = c->health;
   if (!h.fallback_reason) {
      h.fallback_reason = reason;
   }
   h.last_reason = reason;
   h.fallbacks++;
   // loud, but not endlessly so
   if (h.fallbacks <= 8) {
      fprintf(stderr, "libmmoore_hip: a scan published a block that fails validation (%s, reason %llu): header %llx %llx %llx %llx %llx %llx %llx %llx, "
                      "flag word %llu for sequence %llu; the scan is rerun through the plain kernels\n", what, (unsigned long long)reason,
              (unsigned long long)w.h_result[0], (unsigned long long)w.h_result[1], (unsigned long long)w.h_result[2], (unsigned long long)w.h_result[3],
              (unsigned long long)w.h_result[4], (unsigned long long)w.h_result[5], (unsigned long long)w.h_result[6], (unsigned long long)w.h_result[7],
              (unsigned long long)w.h_result[MM_HDR_FLAG_WORD], (unsigned long long)w.seq);
   }
}

// the header of a polled scan: flag bits, counters against the scan's capacities
uint64_t validate_header(const MmWorkspace &w, bool was_fused, bool was_bucketed)
{
   const uint64_t *h = w.h_result;
   const uint64_t flags = h[4] & 0xFF;
   if ((flags & ~7ull) || h[3] != 0 || h[7] != 0) {
      return MMH_FB_HEADER;
   }
   if (((flags & 2) && (!was_fused || (flags & 1))) ||       // only the single-launch kernel gives up, and resolves nothing then
       ((flags & 4) && !(was_bucketed && (flags & 1))) ||    // only mm_scan_tail2 leaves a resolved list on the device
       (!was_fused && (h[4] >> 8) != 0)) {                   // only the single-launch kernel stamps its streaming phase
      return MMH_FB_HEADER;
   }
   if (!(flags & 1)) {
      return h[6] != 0 ? MMH_FB_HEADER : MMH_FB_NONE;       // nothing ordered: no match count
   }
   const uint64_t n = h[0];
   if (n > w.out_cap || n > w.limit || n > w.max_rank || h[6] == 0 || h[6] - 1 > n) {
      return MMH_FB_CAPACITY;
   }
   if ((h[5] & 0xFFFFFFFFull) > n) {
      return MMH_FB_HEADER;                                 // more left-overs than candidates
   }
   return MMH_FB_NONE;
}

// The n slots of a resolved list without left-overs: none still poisoned (a slot store that has not landed is waited
// for: 2 ms, a thousand PCIe round trips), values strictly ascending and inside the ROM, `matches` of them besides the
// holes.  Holes are dropped on the way (the list is left compact).
uint64_t validate_slots(mmh_ctx *c, MmWorkspace &w, const MmGeom &g, uint64_t base_offset, uint64_t n, uint64_t matches, bool direct)
{
   uint64_t *slots = w.h_result + kHeaderWords;
   const uint64_t lo = g.whole ? 0 : base_offset;
   const uint64_t hi = g.whole ? g.nbytes / g.S : base_offset + g.nbytes;
   uint64_t kept = 0, prev = 0;
   for (uint64_t i = 0; i < n; i++) {
      uint64_t v = slots[i];
      if (v == MM_SLOT_POISON) {
         if (!direct) {
            return MMH_FB_STALE_SLOT;
         }
         volatile uint64_t *slot = slots + i;
         const auto t0 = std::chrono::steady_clock::now();
         while ((v = *slot) == MM_SLOT_POISON) {
            __builtin_ia32_pause();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
               return MMH_FB_STALE_SLOT;
            }
         }
         std::atomic_thread_fence(std::memory_order_acquire);
         c->health.late_slots++;
      }
      if (v == ~0ull) {
         continue;                                          // a candidate the reference does not report
      }
      if (v < lo || v >= hi) {
         return MMH_FB_RANGE;
      }
      if (kept && v <= prev) {
         return MMH_FB_ORDER;
      }
      slots[kept++] = prev = v;
   }
   return kept == matches ? MMH_FB_NONE : MMH_FB_ORDER;
}

// tests: damage the published block on the host the way a lost or reordered write would (mmh_debug_inject)
void inject_header(mmh_ctx *c, MmWorkspace &w)
{
   if (c->health.inject == 1) {
      w.h_result[4] |= 0x40;
      c->health.inject = 0;
   }
   else if (c->health.inject == 5) {
      w.h_result[6] += 1;
      c->health.inject = 0;
   }
}

void inject_slots(mmh_ctx *c, MmWorkspace &w, const MmGeom &g, uint64_t base_offset, uint64_t n)
{
   uint64_t *slots = w.h_result + kHeaderWords;
   const uint32_t kind = c->health.inject;
   if (kind == 2 && n >= 1) {
      slots[n / 2] = MM_SLOT_POISON;
   }
   else if (kind == 3 && n >= 2) {
      std::swap(slots[0], slots[n - 1]);
   }
   else if (kind == 4 && n >= 1) {
      slots[n - 1] = (g.whole ? g.nbytes / g.S : base_offset + g.nbytes) + 5;
   }
   else {
      return;
   }
   c->health.inject = 0;
}

// Wait for an enqueued scan and read what it published.  When mm_resolve left candidates over
// (rare: low-entropy neighbourhoods, degenerate keywords) the second phase runs here:
// mm_resolve2 -> mm_hard_resolve -> the ordering again.  Launching those two kernels with every
// scan cost ~10 us of launch latency for nothing in the usual case.
// A fused scan announces its end by raising its sequence number in pinned memory: the host spins
// on that word (no event, no interrupt: the results are a PCIe write away) and only falls back to
// the kernel's completion event should the word never change.
int wait_fused(MmWorkspace &w, hipEvent_t done)
{
   volatile uint64_t *flag = w.h_result + MM_HDR_FLAG_WORD;
   const auto t0 = std::chrono::steady_clock::now();
   for (uint64_t spins = 1;; spins++) {
      if (*flag == w.seq) {
         break;
      }
      __builtin_ia32_pause();
      if ((spins & 0xFFFF) == 0) {
         const hipError_t q = hipEventQuery(done);
         if (q == hipSuccess) {
            if (*flag == w.seq) {
               break;
            }
            mmh_set_error("fused scan kernel ended without publishing its results");
            return MMH_E_DEVICE;
         }
         if (q != hipErrorNotReady) {
            mmh_set_error("fused scan kernel: %s", hipGetErrorString(q));
            return MMH_E_DEVICE;
         }
         if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
            mmh_set_error("fused scan kernel: no result after 60 s");
            return MMH_E_DEVICE;
         }
      }
   }
   std::atomic_thread_fence(std::memory_order_acquire);
   return MMH_OK;
}

int finish_pipeline(mmh_ctx *c, MmWorkspace &w, hipStream_t st, hipEvent_t *ev, const MmGeom &g, const mmh_plan_desc &pl,
                    uint64_t base_offset, uint32_t max_candidates, bool sequential, Outcome *oc, bool part_of_split = false)
{
   max_candidates = oc->limit = w.limit;        // (what enqueue_pipeline settled on)
   if (w.polled) {
      w.polled = false;
      const bool was_fused = w.fused;
      w.fused = false;
      const int rc = wait_fused(w, ev[2]);
      g_sync_trace.mark(2);
      if (was_fused) {
         g_fused_lock.unlock();
      }
      if (rc != MMH_OK) {
         return rc;
      }
      const uint64_t flags = w.h_result[4];
      w.fused_filter_ms = was_fused ? (float)((double)(flags >> 8) * 1e-5) : 0.0f;   // 100 MHz ticks -> ms
      // (header word 1, bits 40-59: from the end of the streaming phase to the header, same clock)
      w.fused_total_ms = was_fused ? w.fused_filter_ms + (float)((double)((w.h_result[1] >> 40) & 0xFFFFF) * 1e-5) : 0.0f;
      static const bool trace = getenv("MMOORE_FUSED_TRACE") != nullptr;
      if (trace && was_fused) {
         const uint64_t st = w.h_result[1];
         fprintf(stderr, "fused scan: streaming %.2f us; after the last arrival: wg0 past the barrier %.2f us, wg0 done %.2f us, header %.2f us; %llu candidates\n",
                 (double)(flags >> 8) * 1e-2, (double)(st & 0xFFFFF) * 1e-2, (double)((st >> 20) & 0xFFFFF) * 1e-2,
                 (double)((st >> 40) & 0xFFFFF) * 1e-2, (unsigned long long)w.h_result[0]);
      }
      if (flags & 2) {
         // a grid barrier timed out (the GPU is shared with something that kept workgroups out):
         // correct results come from the plain kernels below; do not try again on this context
         c->fused_ok = false;
      }
      const bool was_bucketed = w.bucketed;
      w.bucketed = false;
      if (was_bucketed) {
         w.buckets_clean = true;                  // mm_scan_tail2's last workgroup zeroed the bucket counters before it raised the flag
      }
      // Nothing of the block is trusted before it has been validated (include/mmoore_hip.h, "route health"); a block that
      // fails is not repaired: the scan runs again through the plain kernels, whose end is a HIP event.
      inject_header(c, w);
      c->health.validated++;
      uint64_t violation = validate_header(w, was_fused, was_bucketed);
      const char *what = "header";
      if (!violation && (flags & 1)) {
         // one slot per candidate, in offset order; ~0 = a candidate the reference does not report
         oc->candidates = w.h_result[0];
         const bool leftovers = (w.h_result[5] & 0xFFFFFFFFu) != 0;
         const bool direct = !(flags & 4);
         if (!direct && !leftovers && oc->candidates != 0) {
            // a long list: the slots were only written to the device-side copy of the block (a PCIe write per slot
            // would take longer than the scan): one copy brings them over.  On the context's own stream, behind the
            // tail kernel's end event (the flag word shows before the kernel has retired and its stores are visible to
            // a copy engine) -- NOT on the scan's stream: with scans in flight the next scan's streaming kernel is
            // already queued there, and waiting for the copy would mean waiting for that scan (measured: C4 / C5,
            // 8.2 - 8.5 K candidates, ran one scan at a time with three tickets outstanding).
            static const bool fetch_trace = getenv("MMOORE_SPLIT_TRACE") != nullptr;
            const auto t_fetch = std::chrono::steady_clock::now();
            HIP_TRY(hipStreamWaitEvent(c->own_stream, ev[2], 0));
            // Round 5: up to a megabyte comes over by a KERNEL that stores it into the pinned block and raises a word behind
            // the scan's flag, which this thread polls -- hipMemcpyAsync + hipStreamSynchronize took 82-120 us for the 130-200 KiB
            // of a part with 16-24 K matches (the runtime's time, not the DMA engine's; MMOORE_PUBLISH_KERNEL=0: as before).
            static const bool by_kernel = [] { const char *e = getenv("MMOORE_PUBLISH_KERNEL"); return !(e && *e == '0'); }();
            bool fetched = false;
            // (only with the device to itself: beside another part's streaming kernel the copy kernel waits for wave slots --
            // 153 us measured -- where the copy engine's 100 us at least overlap the device's work)
            if (by_kernel && c->device_idle_hint && oc->candidates <= 131072) {
               w.pub_seq++;
               unsigned long long *arrive = reinterpret_cast<unsigned long long *>(w.d_result[w.result_turn] + MM_HDR_FLAG_WORD + 1);
               volatile uint64_t *word = w.h_result + MM_HDR_FLAG_WORD + 1;
               mm::launch_publish_list(c->own_stream, w.d_result[w.result_turn] + kHeaderWords, w.h_result + kHeaderWords,
                                       (uint32_t)oc->candidates, arrive, reinterpret_cast<unsigned long long *>(w.h_result + MM_HDR_FLAG_WORD + 1),
                                       w.pub_seq);
               HIP_TRY(hipGetLastError());
               for (uint64_t spins = 1; !fetched; spins++) {
                  if (*word == w.pub_seq) {
                     fetched = true;
                     break;
                  }
                  __builtin_ia32_pause();
                  if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->own_stream) == hipSuccess) {
                     fetched = *word == w.pub_seq;
                     break;                          // (the kernel has retired: the word is there, or the copy below repairs it)
                  }
               }
               std::atomic_thread_fence(std::memory_order_acquire);
               (void)hipGetLastError();               // (hipErrorNotReady of the queries)
            }
            if (!fetched) {
               HIP_TRY(hipMemcpyAsync(w.h_result + kHeaderWords, w.d_result[w.result_turn] + kHeaderWords, oc->candidates * sizeof(uint64_t),
                                      hipMemcpyDeviceToHost, c->own_stream));
               HIP_TRY(hipStreamSynchronize(c->own_stream));
            }
            if (fetch_trace) {
               fprintf(stderr, "   a list of %llu slots fetched from the device in %.1f us\n", (unsigned long long)oc->candidates,
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fetch).count() * 1e6);
            }
         }
         note_dirty_slots(w, oc->candidates);
         oc->listed = oc->candidates;
         oc->tiles = w.h_result[2];
         oc->hard = 0;
         oc->hard_overflow = (w.h_result[5] & 0xFFFFFFFFu) > mm::mid_cap();
         oc->sorted_on_device = true;
         oc->matches = w.h_result[6] - 1;
         if (!leftovers) {
            inject_slots(c, w, g, base_offset, oc->candidates);
            static const bool slots_trace = getenv("MMOORE_SPLIT_TRACE") != nullptr;
            const auto t_slots = std::chrono::steady_clock::now();
            violation = validate_slots(c, w, g, base_offset, oc->candidates, oc->matches, direct);
            if (slots_trace && oc->candidates > 4096) {
               fprintf(stderr, "   %llu slots validated in %.1f us (%s)\n", (unsigned long long)oc->candidates,
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t_slots).count() * 1e6,
                       direct ? "published straight into pinned memory" : "fetched");
            }
            what = "result slots";
            if (!violation) {
               w.ctrl_clean = true;               // the kernel's last workgroup re-zeroed the control block
               g_sync_trace.mark(3);
               return MMH_OK;
            }
         }
         // left-overs: the second phase below orders the slots again with the rank kernels
      }
      if (violation) {
         note_violation(c, violation, w, what);
         HIP_TRY(hipEventSynchronize(ev[2]));     // (whatever published that block has retired)
         // (the rejected scan's slots: whatever it wrote goes back to the poison before the next polled launch)
         note_dirty_slots(w, std::min<uint64_t>(w.h_result[0], w.max_rank));
         mm::FilterChoice fc;
         mm::choose_filter(pl, &fc);
         w.ctrl_clean = false;
         w.buckets_clean = false;
         const int again = enqueue_pipeline(c, w, st, ev, g, pl, fc, false, base_offset, max_candidates, nullptr, false, false);
         if (again != MMH_OK) {
            return again;
         }
         max_candidates = oc->limit = w.limit;
         HIP_TRY(hipEventSynchronize(ev[2]));
         read_outcome(w, sequential, oc);
      }
      else if (flags & 1) {
         // (left-overs: fall through to the second phase)
      }
      else if (was_bucketed && part_of_split) {
         // (a part of scan_split: the pipeline is given up and the caller decides what to do about the flood -- finer
         // parts, whose buckets are narrower, or the whole ROM the usual way; running the list-based kernels over this
         // part would be another pass over it for a list nobody reads)
         HIP_TRY(hipEventSynchronize(ev[2]));
         oc->candidates = ~0ull;
         oc->bucket_overflow = true;
         w.ctrl_clean = false;
         return MMH_OK;
      }
      else if (was_bucketed) {
         // a bucket overflowed (a flood of candidates in one ROM neighbourhood) or there are more candidates than the
         // published block holds: the list-based kernels, from the start (they take floods apart domain by domain)
         mm::FilterChoice fc;
         mm::choose_filter(pl, &fc);
         w.ctrl_clean = false;
         const int again = enqueue_pipeline(c, w, st, ev, g, pl, fc, false, base_offset, max_candidates, nullptr, false, false);
         if (again != MMH_OK) {
            return again;
         }
         max_candidates = oc->limit = w.limit;
         HIP_TRY(hipEventSynchronize(ev[2]));
         read_outcome(w, sequential, oc);
      }
      else {
         // too many candidates for the in-kernel ranking, or the kernel gave up: the plain
         // kernels take over on the candidate lists it left (control block kept)
         const mm::ResolveBuffers rb = resolve_buffers(w);
         poison_dirty_slots(w);
         w.h_result[6] = 0;
         mm::launch_resolve(st, g, pl, rb, base_offset, max_candidates);
         mm::launch_rank_sort(st, w.d_out, w.d_ctrl, 0, w.out_cap, kMaxRankSort, w.d_partials, w.h_result, w.d_result[w.result_turn],
                              nullptr, true);
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipEventRecord(ev[2], st));
         HIP_TRY(hipEventSynchronize(ev[2]));
         read_outcome(w, sequential, oc);
      }
   }
   else {
      // waiting on the scan's last event returns ~6 us sooner than hipStreamSynchronize on this
      // stack (measured: 12 vs 18-20 us between the end of the device work and the caller)
      HIP_TRY(hipEventSynchronize(ev[2]));
      read_outcome(w, sequential, oc);
   }
   const uint64_t leftovers = sequential ? 0 : (w.h_result[5] & 0xFFFFFFFFu);
   if (leftovers == 0) {
      w.ctrl_clean = true;                      // mm_rank_scatter's last block re-zeroed the control block
      return MMH_OK;
   }
   w.ctrl_clean = false;                        // kept for the second phase
   if (leftovers > mm::mid_cap() || oc->candidates > w.out_cap || oc->candidates > max_candidates) {
      return MMH_OK;                            // the caller switches engines (hard_overflow / too many candidates)
   }
   const mm::ResolveBuffers rb = resolve_buffers(w);
   w.h_result[6] = 0;
   mm::launch_leftovers(st, g, pl, rb, base_offset);
   mm::launch_rank_sort(st, w.d_out, w.d_ctrl, 0, w.out_cap, kMaxRankSort, w.d_partials, w.h_result, w.d_result[w.result_turn]);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipEventRecord(ev[2], st));
   HIP_TRY(hipEventSynchronize(ev[2]));
   read_outcome(w, sequential, oc);
   w.ctrl_clean = true;
   return MMH_OK;
}

int run_pipeline(mmh_ctx *c, const MmGeom &g, const mmh_plan_desc &pl, const mm::FilterChoice &fc, bool sequential,
                 uint64_t base_offset, uint32_t max_candidates, Outcome *oc, const uint32_t *skip_bits = nullptr)
{
   begin_scan_events(c, !sequential);
   c->scans_recorded++;
   const int slot = (int)((c->scans_recorded - 1) % mmh_ctx::kRing);
   c->ring_filter_ms[slot] = 0;
   int rc = enqueue_pipeline(c, c->ws[0], c->stream, c->ev, g, pl, fc, sequential, base_offset, max_candidates, skip_bits, true, true);
   if (rc != MMH_OK) {
      return rc;
   }
   g_sync_trace.mark(1);
   const bool fused = c->ws[0].fused;
   if (fused) {
      c->ring_has_filter[slot] = false;            // one launch: no event marks the end of its streaming phase
   }
   c->device_idle_hint = true;                  // (a synchronous scan: this wait is all the context is doing)
   rc = finish_pipeline(c, c->ws[0], c->stream, c->ev, g, pl, base_offset, max_candidates, sequential, oc);
   c->device_idle_hint = false;
   if (fused) {
      c->ring_filter_ms[slot] = c->ws[0].fused_filter_ms;
      if (!c->ring_timed[slot]) {
         // (no start event: the kernel's own clock, from its first workgroup's start to its header)
         c->ring_is_ms[slot] = true;
         c->ring_ms[slot][0] = c->ws[0].fused_filter_ms;
         c->ring_ms[slot][1] = c->ws[0].fused_total_ms;
      }
   }
   return rc;
}

int grow(uint64_t **buf, uint64_t *cap, uint64_t need)
{
   if (need > *cap) {
      if (*buf) {
         HIP_TRY(hipFree(*buf));
         *buf = nullptr;
         *cap = 0;
      }
      const uint64_t want = need + need / 4 + 1024;
      HIP_TRY(hipMalloc(buf, want * sizeof(uint64_t)));
      *cap = want;
   }
   return MMH_OK;
}

// n keys in device memory -> ascending in c->d_sort_out ("not a match" slots, ~0, end up behind the matches)
int sort_on_device(mmh_ctx *c, const uint64_t *keys, uint64_t n)
{
   int rc = grow(&c->d_sort_out, &c->sort_out_cap, n);
   if (rc != MMH_OK) {
      return rc;
   }
   rc = ensure_sort_temp(c, n);
   if (rc != MMH_OK) {
      return rc;
   }
   HIP_TRY(mm::sort_keys(c->stream, keys, c->d_sort_out, n, c->d_sort_tmp, c->sort_tmp_bytes));
   return MMH_OK;
}

// An ascending device list of n keys (holes behind the matches) to the caller: through a ring of two pinned pieces, the
// DMA of piece k + 1 under way while the CPU copies piece k to its place -- 8-10 GB/s, bound by that copy.  (Round 3 let
// hipMemcpyAsync write straight into a freshly value-initialised std::vector, pageable memory: 134 MB of offsets reached the
// caller at 1.6 GB/s, through three passes over them.)  dst may be null or too small: then the keys are only counted.
// *matches = keys in front of the first hole.
constexpr uint64_t kPiece = 1u << 20;                        // keys per piece of a long list's way to the host: 8 MiB

// the two pinned pieces long lists come to the host through (first use, or ahead of it: prepare_scans)
int ensure_fetch_ring(mmh_ctx *c)
{
   for (int k = 0; k < 2; k++) {
      if (!c->h_ring[k]) {
         HIP_TRY(hipHostMalloc(&c->h_ring[k], kPiece * sizeof(uint64_t), hipHostMallocDefault));
         HIP_TRY(hipEventCreateWithFlags(&c->ring_ev[k], hipEventDisableTiming));
      }
   }
   return MMH_OK;
}

int ensure_sort_temp(mmh_ctx *c, uint64_t n)
{
   const size_t tmp = mm::sort_temp_bytes(n);
   if (tmp > c->sort_tmp_bytes) {
      if (c->d_sort_tmp) {
         HIP_TRY(hipFree(c->d_sort_tmp));
         c->d_sort_tmp = nullptr;
         c->sort_tmp_bytes = 0;
      }
      HIP_TRY(hipMalloc(&c->d_sort_tmp, tmp + tmp / 4));
      c->sort_tmp_bytes = tmp + tmp / 4;
   }
   return MMH_OK;
}

int fetch_device_list(mmh_ctx *c, const uint64_t *d_list, uint64_t n, uint64_t *dst, uint64_t cap, uint64_t *matches)
{
   *matches = 0;
   if (n == 0) {
      return MMH_OK;
   }
   {
      const int rc = ensure_fetch_ring(c);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   const uint64_t pieces = (n + kPiece - 1) / kPiece;
   auto issue = [&](uint64_t k) -> hipError_t {
      const uint64_t len = std::min(kPiece, n - k * kPiece);
      hipError_t e = hipMemcpyAsync(c->h_ring[k & 1], d_list + k * kPiece, len * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
      return e != hipSuccess ? e : hipEventRecord(c->ring_ev[k & 1], c->stream);
   };
   HIP_TRY(issue(0));
   uint64_t kept = 0;
   bool holes = false;
   for (uint64_t k = 0; k < pieces; k++) {
      HIP_TRY(hipEventSynchronize(c->ring_ev[k & 1]));
      if (k + 1 < pieces && !holes) {
         HIP_TRY(issue(k + 1));
      }
      const uint64_t len = std::min(kPiece, n - k * kPiece);
      const uint64_t *src = static_cast<const uint64_t *>(c->h_ring[k & 1]);
      uint64_t valid = len;
      if (src[len - 1] == ~0ull) {
         valid = (uint64_t)(std::lower_bound(src, src + len, ~0ull) - src);
         holes = true;
      }
      if (dst && kept + valid <= cap) {
         std::memcpy(dst + kept, src, valid * sizeof(uint64_t));
      }
      kept += valid;
      if (holes) {
         break;                                               // (everything behind the first hole is holes)
      }
   }
   HIP_TRY(hipStreamSynchronize(c->stream));
   *matches = kept;
   return MMH_OK;
}

// n keys in device memory -> ascending in host memory, "not a match" slots (~0) dropped
int sort_to_host(mmh_ctx *c, const uint64_t *keys, uint64_t n, std::vector<uint64_t> *sorted)
{
   sorted->clear();
   if (n == 0) {
      return MMH_OK;
   }
   int rc = sort_on_device(c, keys, n);
   if (rc != MMH_OK) {
      return rc;
   }
   sorted->resize(n);
   uint64_t matches = 0;
   rc = fetch_device_list(c, c->d_sort_out, n, sorted->data(), n, &matches);
   if (rc != MMH_OK) {
      return rc;
   }
   sorted->resize(matches);
   return MMH_OK;
}

// The candidate-free forward engine (mm_forward.h).  Matches land in MM_CAND_LISTS device
// lists; they are fetched and ordered on the host (dense results are long lists anyway).
// found == nullptr: the list stays on the device, ordered, in c->d_sort_out; *device_n = its length incl. nothing but matches
int run_dense(mmh_ctx *c, const MmGeom &g, const mmh_plan_desc &pl, uint64_t base_offset, std::vector<uint64_t> *found,
              bool *grew, const uint32_t *dom_list = nullptr, uint64_t listed_domains = 0, uint64_t *device_n = nullptr)
{
   hipStream_t st = c->stream;
   *grew = false;
   if (found) {
      found->clear();
   }
   if (device_n) {
      *device_n = 0;
   }
   const mm::DenseGeom dg = mm::dense_geom(g, listed_domains);
   if (dg.tpd == 0) {
      return MMH_OK;                              // no alignment fits anywhere
   }
   auto round = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
   const size_t need = round(dg.maps_bytes);
   if (need > c->dense_bytes) {
      if (c->d_dense) {
         HIP_TRY(hipFree(c->d_dense));
         c->d_dense = nullptr;
         c->dense_bytes = 0;
      }
      HIP_TRY(hipMalloc(&c->d_dense, need));
      c->dense_bytes = need;
   }
   mm::DenseBuffers db;
   db.maps = c->d_dense;
   db.out = c->ws[0].d_out; db.out_cap = c->ws[0].out_cap; db.ctrl = c->ws[0].d_ctrl;

   {
      const int rc = grow(&c->d_sort_in, &c->sort_in_cap, c->ws[0].out_cap);     // (the lists together never hold more)
      if (rc != MMH_OK) {
         return rc;
      }
   }
   HIP_TRY(hipMemsetAsync(c->ws[0].d_ctrl, 0, mm::ctrl_bytes(), st));
   c->ws[0].ctrl_clean = false;
   begin_scan_events(c, false);
   c->ring_timed[(int)(c->scans_recorded % mmh_ctx::kRing)] = true;   // (this path records its start event whatever mmh_set_timing says)
   HIP_TRY(hipEventRecord(c->ev[0], st));
   HIP_TRY(hipEventRecord(c->ev[1], st));
   mm::launch_dense(st, g, pl, dg, db, base_offset, dom_list);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipEventRecord(c->ev[2], st));
   // the lists one behind the other, for the ordering: one launch right behind the engine (it was a copy per list after
   // the counters had come back: 0.15 ms of enqueueing for a thousand matches)
   const uint64_t list_cap = c->ws[0].out_cap / MM_CAND_LISTS;
   mm::launch_pack_lists(st, c->ws[0].d_out, list_cap, c->ws[0].d_ctrl + MM_CTRL_LISTS, c->d_sort_in);
   HIP_TRY(hipGetLastError());
   std::vector<unsigned long long> ctrl(mm::ctrl_bytes() / sizeof(unsigned long long));
   HIP_TRY(hipMemcpyAsync(ctrl.data(), c->ws[0].d_ctrl, mm::ctrl_bytes(), hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   c->scans_recorded++;

   uint64_t most = 0, total = 0;
   for (int l = 0; l < MM_CAND_LISTS; l++) {
      const uint64_t n = ctrl[MM_CTRL_LISTS + l * MM_LIST_STRIDE];
      most = std::max(most, n);
      total += n;
   }
   if (most > list_cap) {
      // some list overflowed: size every list for the fullest one and run again
      int rc = ensure_workspace(c, c->ws[0], (most + most / 8 + 1024) * MM_CAND_LISTS);
      if (rc != MMH_OK) {
         return rc;
      }
      *grew = true;
      return MMH_OK;
   }
   // order them the way search_engine.cpp:193-197 does -- on the device
   if (!found) {
      *device_n = total;
      return total ? sort_on_device(c, c->d_sort_in, total) : MMH_OK;
   }
   return sort_to_host(c, c->d_sort_in, total, found);
}

// run_dense until its output lists fit (each retry sizes them for the fullest list seen, so the
// second attempt fits); running out of attempts is an error, never a truncated list
int run_dense_settled(mmh_ctx *c, const MmGeom &g, const mmh_plan_desc &pl, uint64_t base_offset, std::vector<uint64_t> *found,
                      const uint32_t *dom_list, uint64_t listed_domains)
{
   for (int attempt = 0; attempt < 4; attempt++) {
      bool grew = false;
      int rc = run_dense(c, g, pl, base_offset, found, &grew, dom_list, listed_domains);
      if (rc != MMH_OK || !grew) {
         return rc;
      }
   }
   mmh_set_error("forward engine: the output lists still overflow after 4 attempts");
   return MMH_E_STATE;
}

} // namespace

namespace {

// A scan whose left-over lists overflowed (floods of candidates only the domain prefix can
// settle: matches right behind long constant runs, say).  Instead of sending the whole ROM to
// the forward engine: (1) flag pass -- mm_resolve again, setting the bit of every domain that
// holds an unsettled candidate; (2) forward engine over the flagged domains only; (3) the first
// pass's verdicts stand everywhere else.  Engine mode only (a whole-buffer scan is one domain).
// *handled = false: the flag pass did not see the candidates of the first pass (a candidate list overflowed --
// the two passes may spread the candidates over the lists differently: single-launch kernel first, plain
// streaming kernel here): the caller switches to the forward engine for everything
int run_flagged_domains(mmh_ctx *c, const MmGeom &g, const mmh_plan_desc &pl, uint64_t base_offset, uint32_t max_candidates,
                        uint64_t first_pass_slots, std::vector<uint64_t> *merged, uint64_t *domains_flagged, bool *handled)
{
   *handled = false;
   hipStream_t st = c->stream;
   MmWorkspace &w = c->ws[0];
   const uint64_t ndom = g.nblocks * g.S;
   const uint64_t words = (ndom + 31) / 32;
   int rc = grow(&c->d_domains, &c->domains_cap, (words + 1) / 2 + ndom / 2 + 2);   // bitmap, then the domain list, as u32
   if (rc != MMH_OK) {
      return rc;
   }
   uint32_t *d_bits = reinterpret_cast<uint32_t *>(c->d_domains);
   HIP_TRY(hipMemsetAsync(d_bits, 0, words * sizeof(uint32_t), st));
   HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, mm::ctrl_bytes(), st));
   w.ctrl_clean = false;
   // the candidate lists of the first pass are gone with the control block: run the filter again
   mm::FilterChoice fc;
   mm::choose_filter(pl, &fc);
   const mm::ResolveBuffers rb = resolve_buffers(w);
   mm::launch_filter(st, g, pl, fc, w.d_cand, w.d_ctrl, w.cand_cap);
   mm::launch_resolve(st, g, pl, rb, base_offset, max_candidates, d_bits);
   HIP_TRY(hipGetLastError());
   std::vector<uint32_t> bits(words);
   std::vector<uint64_t> slots(first_pass_slots);
   unsigned long long seen = 0;
   HIP_TRY(hipMemcpyAsync(bits.data(), d_bits, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
   // every candidate got its slot again (same candidates; their order may differ from the first pass)
   HIP_TRY(hipMemcpyAsync(slots.data(), w.d_out, first_pass_slots * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
   HIP_TRY(hipMemcpyAsync(&seen, w.d_ctrl + MM_CTRL_TOTAL, sizeof seen, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   if (seen != first_pass_slots) {
      // (~0: a list overflowed and mm_resolve did nothing -- found by a fuzz soak: 'bbbb' on a two-symbol
      // alphabet, 165 K candidates in 1 MiB, reported 527 of 43538 matches from stale slots)
      return MMH_OK;
   }

   std::vector<uint32_t> doms;
   for (uint64_t d = 0; d < ndom; d++) {
      if ((bits[d >> 5] >> (d & 31)) & 1u) {
         doms.push_back((uint32_t)d);
      }
   }
   *domains_flagged = doms.size();
   auto flagged = [&](uint64_t reported) {
      const uint64_t o = reported - base_offset;
      const uint64_t blk = o / g.block_bytes;
      const uint64_t d = blk * g.S + (o - blk * g.block_bytes) % g.S;
      return ((bits[d >> 5] >> (d & 31)) & 1u) != 0;
   };
   merged->clear();
   for (uint64_t v : slots) {
      if (v != ~0ull && !flagged(v)) {
         merged->push_back(v);
      }
   }
   if (!doms.empty()) {
      uint32_t *d_list = d_bits + words + (words & 1);
      HIP_TRY(hipMemcpyAsync(d_list, doms.data(), doms.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
      std::vector<uint64_t> dense;
      rc = run_dense_settled(c, g, pl, base_offset, &dense, d_list, doms.size());
      if (rc != MMH_OK) {
         return rc;
      }
      if (getenv("MMOORE_TRACE")) {
         fprintf(stderr, "run_flagged_domains: forward engine on %zu domains: %zu results\n", doms.size(), dense.size());
      }
      merged->insert(merged->end(), dense.begin(), dense.end());
   }
   static const bool trace = getenv("MMOORE_TRACE") != nullptr;
   if (trace) {
      uint64_t holes = 0, in_flagged = 0;
      for (uint64_t v : slots) {
         holes += v == ~0ull;
         in_flagged += v != ~0ull && flagged(v);
      }
      fprintf(stderr, "run_flagged_domains: %llu slots (%llu holes, %llu verdicts inside flagged domains), %zu of %llu domains flagged, %zu results after the merge\n",
              (unsigned long long)slots.size(), (unsigned long long)holes, (unsigned long long)in_flagged, doms.size(), (unsigned long long)ndom, merged->size());
   }
   std::sort(merged->begin(), merged->end());       // search_engine.cpp:193-197
   w.ctrl_clean = false;
   *handled = true;
   return MMH_OK;
}

// A scan with more candidates than the per-candidate path takes (a keyword that matches a
// whole padding run, say: 'abcde' on a +1 ramp, 'aaaa' on zeros).  Instead of sending the whole
// ROM to the forward engine: (1) count pass -- the filter again, counting candidates per domain;
// (2) the fullest domains are flagged until the rest fits; (3) the filter + resolver pipeline
// with the flagged domains masked out; (4) the forward engine over the flagged domains;
// (5) merge.  Engine mode only.  *handled = false: no use (floods everywhere) -> caller's fallback.
int run_candidate_floods(mmh_ctx *c, const MmGeom &g, const mmh_plan_desc &pl, const mm::FilterChoice &fc, uint64_t base_offset,
                         uint32_t max_candidates, std::vector<uint64_t> *merged, uint64_t *domains_flagged, bool *handled,
                         uint64_t *counted)
{
   hipStream_t st = c->stream;
   MmWorkspace &w = c->ws[0];
   *handled = false;
   const uint64_t ndom = g.nblocks * g.S;
   if (ndom < 2 || ndom > (1ull << 26)) {
      return MMH_OK;
   }
   const uint64_t words = (ndom + 31) / 32;
   // one scratch allocation, as u32: [ndom counts][words bitmap][ndom domain list]
   int rc = grow(&c->d_domains, &c->domains_cap, (2 * ndom + words) / 2 + 4);
   if (rc != MMH_OK) {
      return rc;
   }
   unsigned int *d_count = reinterpret_cast<unsigned int *>(c->d_domains);
   uint32_t *d_bits = reinterpret_cast<uint32_t *>(d_count + ndom);
   uint32_t *d_list = d_bits + words;
   HIP_TRY(hipMemsetAsync(d_count, 0, ndom * sizeof(unsigned int), st));
   HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, mm::ctrl_bytes(), st));
   w.ctrl_clean = false;
   mm::launch_filter(st, g, pl, fc, w.d_cand, w.d_ctrl, w.cand_cap, nullptr, nullptr, d_count, nullptr);
   HIP_TRY(hipGetLastError());
   std::vector<unsigned int> count(ndom);
   HIP_TRY(hipMemcpyAsync(count.data(), d_count, ndom * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));

   // flag the fullest domains until what is left fits the per-candidate path comfortably
   std::vector<uint32_t> order(ndom);
   unsigned long long total = 0;
   for (uint64_t d = 0; d < ndom; d++) {
      order[d] = (uint32_t)d;
      total += count[d];
   }
   *counted = total;                              // (what the count pass saw: the scan's candidate counter, mmh_last_counters)
   std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return count[x] > count[y]; });
   std::vector<uint32_t> bits(words, 0), doms;
   for (uint64_t k = 0; k < ndom && total > max_candidates / 2; k++) {
      const uint32_t d = order[k];
      bits[d >> 5] |= 1u << (d & 31);
      doms.push_back(d);
      total -= count[d];
   }
   if (doms.empty() || doms.size() > ndom / 2) {
      return MMH_OK;                              // candidates everywhere: the forward engine on everything it is
   }
   std::sort(doms.begin(), doms.end());
   HIP_TRY(hipMemcpyAsync(d_bits, bits.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(d_list, doms.data(), doms.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));

   Outcome oc;
   rc = run_pipeline(c, g, pl, fc, false, base_offset, max_candidates, &oc, d_bits);
   if (rc != MMH_OK) {
      return rc;
   }
   if (oc.candidates > w.out_cap || oc.candidates > max_candidates) {
      return MMH_OK;                              // still too much for the resolvers: caller's fallback
   }
   std::vector<uint64_t> sparse;
   if (oc.hard_overflow) {
      // On top of the flood, more undecidable candidates than the left-over lists take: flag
      // their domains as well (the flag pass of run_flagged_domains, flooded domains masked out).
         const mm::ResolveBuffers rb = resolve_buffers(w);
      // (the second resolver phase may have run and left the control block zeroed: filter again)
      HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, mm::ctrl_bytes(), st));
      mm::launch_filter(st, g, pl, fc, w.d_cand, w.d_ctrl, w.cand_cap, nullptr, nullptr, nullptr, d_bits);
      mm::launch_resolve(st, g, pl, rb, base_offset, max_candidates, d_bits);
      HIP_TRY(hipGetLastError());
      std::vector<uint64_t> slots(oc.candidates);
      HIP_TRY(hipMemcpyAsync(bits.data(), d_bits, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(slots.data(), w.d_out, slots.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      w.ctrl_clean = false;
      doms.clear();
      for (uint64_t d = 0; d < ndom; d++) {
         if ((bits[d >> 5] >> (d & 31)) & 1u) {
            doms.push_back((uint32_t)d);
         }
      }
      if (doms.size() > ndom / 2) {
         return MMH_OK;
      }
      HIP_TRY(hipMemcpyAsync(d_list, doms.data(), doms.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
      for (uint64_t v : slots) {
         if (v == ~0ull) {
            continue;
         }
         const uint64_t o = v - base_offset;
         const uint64_t blk = o / g.block_bytes;
         const uint64_t d = blk * g.S + (o - blk * g.block_bytes) % g.S;
         if (((bits[d >> 5] >> (d & 31)) & 1u) == 0) {
            sparse.push_back(v);
         }
      }
      std::sort(sparse.begin(), sparse.end());
   }
   else if (oc.sorted_on_device) {
      sparse.assign(w.h_result + kHeaderWords, w.h_result + kHeaderWords + oc.matches);
   }
   else {
      rc = sort_to_host(c, w.d_out, oc.listed, &sparse);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   std::vector<uint64_t> dense;
   rc = run_dense_settled(c, g, pl, base_offset, &dense, d_list, doms.size());
   if (rc != MMH_OK) {
      return rc;
   }
   merged->resize(sparse.size() + dense.size());
   std::merge(sparse.begin(), sparse.end(), dense.begin(), dense.end(), merged->begin());
   *domains_flagged = doms.size();
   *handled = true;
   return MMH_OK;
}

int check_scan_args(const mmh_ctx *c, const mmh_plan_desc *plan, const char *who)
{
   if (!c->rom) {
      mmh_set_error("%s: no ROM attached", who);
      return MMH_E_STATE;
   }
   if (plan->L < 2 || plan->L > MMH_MAX_KEYWORD || (plan->elem_bytes != 1 && plan->elem_bytes != 2) ||
       plan->match_jump < 1 || plan->n_skip > MMH_MAX_KEYWORD) {
      mmh_set_error("%s: malformed plan", who);
      return MMH_E_PLAN;
   }
   return MMH_OK;
}

MmGeom scan_geometry(const mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, const MmPending *view = nullptr)
{
   MmGeom g;
   g.rom = c->rom;
   g.nbytes = c->rom_bytes;
   if (view && view->view) {
      g.rom = c->rom + view->view_first;          // (block-aligned, hence 16-byte aligned: scan_split)
      g.nbytes = view->view_bytes;
   }
   g.block_bytes = block_bytes;
   g.S = plan->elem_bytes;
   g.L = plan->L;
   g.big_endian = (plan->elem_bytes == 2 && big_endian) ? 1u : 0u;
   g.whole = block_bytes == 0 ? 1u : 0u;
   g.nblocks = g.whole ? 1 : (g.nbytes + block_bytes - 1) / block_bytes;
   if (g.whole) {
      g.nbytes = (g.nbytes / g.S) * g.S;        // whole elements only, like search(const Ty*, len)
   }
   return g;
}

// More candidates than this and the forward engine (cost linear in the ROM: ~5 ms per GiB)
// is the better deal: the resolvers take ~2-10 ns per candidate (measured: 65 K candidates
// of a 3-symbol keyword on 4 GiB add 0.12 ms, against 21 ms for the forward engine).
// Lists beyond kMaxRankSort entries are ordered by the radix sort of mm_sort.hip.
uint32_t candidate_limit(const MmWorkspace &w)
{
   // (round 3: what the bucketed path takes -- one result slot per candidate; the list-based kernels notice by themselves
   // when one of their 64 lists overflows, which sends the scan on to the flood / forward paths)
   return (uint32_t)std::min<uint64_t>(mm::tuning().max_candidates, w.out_cap);
}

} // namespace

namespace {
// *kept: the list when it only exists in host memory (long lists, forward engine); *on_device: the
// list sits ordered in ws[0].d_result[result_turn]
// view (scan_split): the bytes [view_first, view_first + view_bytes) of the ROM, block-aligned, instead of all of it;
// start_dense: the per-candidate path is known to overflow on them (this scan's own parts found out) -- forward engine at once
int scan_impl(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, uint64_t *out,
              uint64_t cap, uint64_t *out_count, std::vector<uint64_t> *kept, bool *on_device, const MmPending *view = nullptr,
              bool start_dense = false)
{
   if (!c || !plan || !out_count || (!out && cap)) {
      mmh_set_error("mmh_scan: bad argument");
      return MMH_E_ARG;
   }
   *out_count = 0;
   int rc = check_scan_args(c, plan, "mmh_scan");
   if (rc != MMH_OK) {
      return rc;
   }
   HIP_TRY(hipSetDevice(c->device));
   rc = mm_ingest_drain(c);
   if (rc != MMH_OK) {
      return rc;
   }
   const MmGeom g = scan_geometry(c, plan, block_bytes, big_endian, view);

   rc = ensure_workspace(c, c->ws[0], std::max<uint64_t>(c->ws[0].out_cap, kInitialCap));
   if (rc != MMH_OK) {
      return rc;
   }
   std::memset(c->counters, 0, sizeof(c->counters));
   if (g.nbytes == 0) {
      return MMH_OK;                              // (an empty host list: the gather sends a count of 0)
   }

   mm::FilterChoice fc;
   const bool have_filter = mm::choose_filter(*plan, &fc);
   // engine 0: filter + per-candidate resolvers; inputs they do not suit (no SWAR key in the
   // pattern, candidate sets too dense, prefixes too long) go to the forward "dense" engine,
   // whose cost is linear in the ROM.  1 / 2 force the sequential / dense engine (tests).
   // (keywords beyond 32 symbols: the resolvers' phase sets do not hold their D > 31 phases)
   const bool narrow = plan->L <= MM_RESOLVER_MAX_KEYWORD;
   enum { FAST, SEQUENTIAL, DENSE } mode = c->engine == 1 ? SEQUENTIAL : (c->engine == 2 || !have_filter || !narrow) ? DENSE : FAST;
   // (No memo of earlier scans: until round 5 a search that had ended on the forward engine went there at once the next
   // time, and every first scan of such a search cost 1.2 to 6 times a later one.  What a scan learns it learns from its
   // own first part, scan_split.)
   if (start_dense && mode == FAST) {
      mode = DENSE;
   }
   const uint32_t scan_limit = candidate_limit(c->ws[0]);
   uint32_t max_candidates = scan_limit;

   Outcome oc;
   std::vector<uint64_t> long_list;
   bool host_list = false, flagged_domains = false, flooded_domains = false;
   bool device_list = false;                      // the result sits ordered in c->d_sort_out (forward engine, radix sort)
   uint64_t device_list_n = 0;
   bool settled = false;                          // the loop ended with a complete result (not by running out of attempts)
   for (int attempt = 0; attempt < 6 && !settled; attempt++) {
      if (mode == DENSE) {
         bool grew = false;
         rc = run_dense(c, g, *plan, base_offset, nullptr, &grew, nullptr, 0, &device_list_n);
         if (rc != MMH_OK) {
            return rc;
         }
         if (grew) {
            continue;
         }
         oc = Outcome();
         device_list = true;                      // (ordered in c->d_sort_out: fetched straight into the caller's buffer below)
         host_list = true;
         settled = true;
         break;
      }
      rc = run_pipeline(c, g, *plan, fc, mode == SEQUENTIAL, base_offset, scan_limit, &oc);
      if (rc != MMH_OK) {
         return rc;
      }
      max_candidates = oc.limit;                  // (the list-based kernels' limit if those ran, see enqueue_pipeline)
      if (mode == FAST && oc.hard_overflow && !g.whole && oc.candidates <= c->ws[0].out_cap && oc.candidates <= max_candidates) {
         // more left-overs than mm_resolve2 / mm_hard_resolve take: forward engine on their domains only
         bool handled = false;
         rc = run_flagged_domains(c, g, *plan, base_offset, max_candidates, oc.candidates, &long_list, &oc.tiles, &handled);
         if (rc != MMH_OK) {
            return rc;
         }
         if (handled) {
            oc.matches = long_list.size();
            host_list = true;
            flagged_domains = true;
            settled = true;
            break;
         }
         mode = DENSE;
         continue;
      }
      if (mode == FAST && !g.whole && (oc.candidates > c->ws[0].out_cap || oc.candidates > max_candidates)) {
         // candidate flood: forward engine on the flooded domains, the per-candidate path on the rest
         bool handled = false;
         uint64_t counted = 0;
         rc = run_candidate_floods(c, g, *plan, fc, base_offset, max_candidates, &long_list, &oc.tiles, &handled, &counted);
         if (rc != MMH_OK) {
            return rc;
         }
         if (handled) {
            oc.candidates = counted;                 // (the first pass only knew "a list overflowed": ~0)
            oc.matches = long_list.size();
            host_list = true;
            flooded_domains = true;
            settled = true;
            break;
         }
      }
      if (mode == FAST && (oc.candidates > c->ws[0].out_cap || oc.candidates > max_candidates || oc.hard_overflow)) {
         mode = DENSE;                            // too dense / too long for the per-candidate resolvers
         continue;
      }
      if (oc.listed > c->ws[0].out_cap) {
         rc = ensure_workspace(c, c->ws[0], oc.listed + oc.listed / 8 + 1024);
         if (rc != MMH_OK) {
            return rc;
         }
         continue;
      }
      settled = true;
   }
   if (!settled) {
      // every retry grows a buffer or switches to an engine that cannot overflow, so two attempts
      // are the most a scan needs today; running out means that invariant broke -- say so instead
      // of publishing a truncated list
      mmh_set_error("mmh_scan: no engine settled the scan in 6 attempts (candidates %llu, listed %llu, capacity %llu)",
                    (unsigned long long)oc.candidates, (unsigned long long)oc.listed, (unsigned long long)c->ws[0].out_cap);
      return MMH_E_STATE;
   }

   // lists too long for the rank kernels: radix sort on the device, "not a match" slots dropped
   // (search_engine.cpp:193-197 does a std::sort)
   if (!host_list && !oc.sorted_on_device) {
      rc = oc.listed ? sort_on_device(c, c->ws[0].d_out, oc.listed) : MMH_OK;
      if (rc != MMH_OK) {
         return rc;
      }
      device_list = true;
      device_list_n = oc.listed;
      host_list = true;
   }
   if (device_list) {
      // device -> the caller's buffer, no host list in between (a communicator may ask for the list later: then it is kept)
      rc = fetch_device_list(c, c->d_sort_out, device_list_n, out, cap, &oc.matches);
      if (rc != MMH_OK) {
         return rc;
      }
      if (c->mg.comm) {
         if (oc.matches <= cap) {
            long_list.assign(out, out + oc.matches);
         }
         else {
            // (the caller's buffer is too small and it will scan again -- but a gather started right now must still
            // find this scan's list)
            long_list.resize(oc.matches);
            uint64_t again = 0;
            rc = fetch_device_list(c, c->d_sort_out, device_list_n, long_list.data(), long_list.size(), &again);
            if (rc != MMH_OK) {
               return rc;
            }
         }
      }
   }
   c->counters[0] = oc.candidates == ~0ull ? 0 : oc.candidates;   // (~0: "a candidate list overflowed" of a pass that was abandoned)
   c->counters[1] = oc.matches;
   c->counters[2] = oc.tiles;
   c->counters[3] = mode == SEQUENTIAL ? 1 : (mode == DENSE ? 3 : (flooded_domains ? 5 : (flagged_domains ? 4 : (oc.hard ? 2 : 0))));
   *out_count = oc.matches;
   *on_device = !host_list;
   rc = MMH_OK;
   if (oc.matches > cap) {
      mmh_set_error("mmh_scan: %llu matches do not fit the caller's buffer of %llu",
                    (unsigned long long)oc.matches, (unsigned long long)cap);
      rc = MMH_E_CAPACITY;
   }
   else if (oc.matches != 0 && !device_list) {
      std::memcpy(out, host_list ? long_list.data() : c->ws[0].h_result + kHeaderWords, oc.matches * sizeof(uint64_t));
   }
   if (host_list) {
      kept->swap(long_list);
   }
   return rc;
}
} // namespace

namespace {
bool split_applies(const mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian);
int scan_split(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, uint64_t *out,
               uint64_t cap, uint64_t *out_count);
void scan_timings(mmh_ctx *c, uint64_t k, float *ms4);
}

// mmh_scan proper + what the multi-GPU gather needs to know about its list (mm_multi.hip)
extern "C" int mmh_scan(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                        uint64_t base_offset, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   if (c) {
      c->mg.last_src = nullptr;
      c->mg.last_count = 0;
      c->mg.last_slots = 0;
      c->mg.last_list.clear();
      c->mg.last_end = nullptr;
   }
   // (the memo keys hash plan->L entries of the plan's tables: a malformed plan is turned away by scan_impl, not read here)
   if (c && plan && out_count && (out || !cap) && c->rom && check_scan_args(c, plan, "mmh_scan") == MMH_OK &&
       split_applies(c, plan, block_bytes, big_endian)) {
      return scan_split(c, plan, block_bytes, big_endian, base_offset, out, cap, out_count);
   }
   std::vector<uint64_t> host_list;
   bool on_device = false;
   g_sync_trace.mark(0);
   g_sync_trace.t[1] = g_sync_trace.t[2] = g_sync_trace.t[3] = g_sync_trace.t[0];
   int rc = scan_impl(c, plan, block_bytes, big_endian, base_offset, out, cap, out_count, &host_list, &on_device);
   g_sync_trace.mark(4);
   g_sync_trace.done();
   if (c && (rc == MMH_OK || rc == MMH_E_CAPACITY)) {
      c->mg.last_count = *out_count;
      c->mg.last_slots = on_device ? c->counters[0] : 0;      // (one slot per candidate)
      c->mg.last_src = on_device ? c->ws[0].d_result[c->ws[0].result_turn] : nullptr;
      c->mg.last_end = on_device ? c->ev[2] : nullptr;   // the gather's stream waits for the scan's last kernel to retire
      if (!on_device && c->mg.comm) {
         c->mg.last_list.swap(host_list);            // only kept when a communicator may ask for it
      }
   }
   return rc;
}

// ---- scans in flight ----------------------------------------------------------------------
//
// mmh_scan_submit enqueues the streaming kernel + tail kernel of a scan on one of three lanes (own
// stream, own workspace, own pinned result block) and returns; mmh_scan_collect waits for it.
// The host's share of a scan (launches, the wait, copying the offsets out) and the tail kernel
// then overlap the NEXT scan's streaming kernel; two scans are at work on the device at a time,
// the third lane holds the one the host has enqueued ahead (see mmh_scan_submit).
// Anything the lanes do not run themselves -- forced engines, patterns without a SWAR key,
// candidate floods, lists beyond the rank kernels -- is rescanned synchronously by collect.

namespace {
// fills in the timings a collected lane scan still owes (see mmh_ctx::lane_timing_owed); its events have
// completed or are about to
void settle_lane_timing(mmh_ctx *c, int lane)
{
   const int64_t k = c->lane_timing_owed[lane];
   if (k < 0) {
      return;
   }
   c->lane_timing_owed[lane] = -1;
   if (c->scans_recorded - (uint64_t)k > mmh_ctx::kRing) {
      return;                                  // its ring slot belongs to a later scan by now
   }
   hipEvent_t *e = c->lane_ev[lane];
   float filter_ms = 0, total_ms = 0;
   if (hipEventSynchronize(e[2]) == hipSuccess && hipEventElapsedTime(&filter_ms, e[0], e[1]) == hipSuccess &&
       hipEventElapsedTime(&total_ms, e[0], e[2]) == hipSuccess) {
      const int slot = (int)((uint64_t)k % mmh_ctx::kRing);
      c->ring_ms[slot][0] = filter_ms;
      c->ring_ms[slot][1] = total_ms;
   }
}
} // namespace

namespace {
int submit_impl(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, int *ticket,
                bool view, uint64_t view_first, uint64_t view_bytes);
}

extern "C" int mmh_scan_submit(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                               uint64_t base_offset, int *ticket)
{
   return submit_impl(c, plan, block_bytes, big_endian, base_offset, ticket, false, 0, 0);
}

namespace {
// a lane's workspace and events, the lanes' two streams and the fence event (first use, or ahead of it: prepare_scans)
int ensure_lane(mmh_ctx *c, int lane)
{
   MmWorkspace &w = c->ws[1 + lane];
   const int rc = ensure_workspace(c, w, std::max<uint64_t>(w.out_cap, kInitialCap));
   if (rc != MMH_OK) {
      return rc;
   }
   for (int k = 0; k < 2; k++) {
      if (!c->lane_stream[k]) {
         HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
      }
   }
   if (!c->lane_fence) {
      HIP_TRY(hipEventCreateWithFlags(&c->lane_fence, hipEventDisableTiming));
   }
   for (auto &e : c->lane_ev[lane]) {
      if (!e) {
         HIP_TRY(hipEventCreate(&e));
      }
   }
   return MMH_OK;
}

int submit_impl(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, int *ticket,
                bool view, uint64_t view_first, uint64_t view_bytes)
{
   if (!c || !plan || !ticket) {
      mmh_set_error("mmh_scan_submit: bad argument");
      return MMH_E_ARG;
   }
   int rc = check_scan_args(c, plan, "mmh_scan_submit");
   if (rc != MMH_OK) {
      return rc;
   }
   const int lane = c->next_ticket % mmh_ctx::kLanes;
   MmPending &p = c->pending[lane];
   if (p.active) {
      mmh_set_error("mmh_scan_submit: %d scans are already outstanding, collect ticket %d first", mmh_ctx::kLanes, p.ticket);
      return MMH_E_STATE;
   }
   HIP_TRY(hipSetDevice(c->device));
   rc = mm_ingest_drain(c);
   if (rc != MMH_OK) {
      return rc;
   }
   MmWorkspace &w = c->ws[1 + lane];
   rc = ensure_lane(c, lane);
   if (rc != MMH_OK) {
      return rc;
   }
   // Two streams: scan t's streaming kernel AND tail kernel on stream t % 2.  (Measured and dropped, r03: every streaming
   // kernel on one stream with the tails on a second; streaming kernels alternating with the tails on a third.)
   const hipStream_t lane_st = c->lane_stream[c->next_ticket % 2];
   const hipStream_t tail_st = lane_st;
   c->pending_tail_stream[lane] = tail_st;
   p = MmPending();
   p.ticket = c->next_ticket;
   p.plan = *plan;
   p.block_bytes = block_bytes;
   p.big_endian = big_endian;
   p.base_offset = base_offset;
   p.max_candidates = candidate_limit(w);
   p.view = view;
   p.view_first = view_first;
   p.view_bytes = view_bytes;

   const MmGeom g = scan_geometry(c, plan, block_bytes, big_endian, &p);
   mm::FilterChoice fc;
   const bool have_filter = mm::choose_filter(*plan, &fc);
   if (c->engine != 0 || !have_filter || g.nbytes == 0 || plan->L > MM_RESOLVER_MAX_KEYWORD) {
      p.needs_rescan = true;                    // collect runs mmh_scan
   }
   else {
      // The ROM may still be in the making on the context's stream (upload, synth, poke; a stream the caller gave us: whatever
      // it put there): the lane waits for that -- when there is something to wait for.  An idle stream (one query, no packet)
      // spares the scan an event record, a marker on that stream and a barrier packet in front of its streaming kernel:
      // ~6 us of a synchronous scan's first part (round 5).
      const hipError_t q = hipStreamQuery(c->stream);
      const bool idle = q == hipSuccess;
      if (q != hipSuccess && q != hipErrorNotReady) {
         (void)hip_ok(q, "hipStreamQuery (the context's stream)");
         return MMH_E_DEVICE;
      }
      if (!idle) {
         (void)hipGetLastError();                   // (hipErrorNotReady is sticky for hipGetLastError)
         HIP_TRY(hipEventRecord(c->lane_fence, c->stream));
         HIP_TRY(hipStreamWaitEvent(lane_st, c->lane_fence, 0));
      }
      // How scans in flight share the device (MMOORE_LANE_TRACE=1; rocprofv3 kernel trace, tools/lane_trace.sh).  Scan t
      // starts behind scan t-2 (same stream) and runs beside scan t-1: its streaming kernel begins on the wave slots
      // the streaming kernel of t-1 leaves free (6 of 7 per SIMD) and takes over as that one's workgroups finish;
      // the tail kernel of t-2 in front of it waits for a slot beside the two streaming kernels (~0.45 ms from
      // dispatch to end for 27 us of work) -- which is what staggers the scans.  Net: 0.69 ms per 4 GiB scan in
      // the steady state, below the duration of ONE streaming kernel run alone -- tail kernel, result hand-over and
      // the gaps between kernels cost nothing.  Three lanes (workspaces, result blocks) on the two streams, although
      // only two scans are ever at work on the device: the third is the one the host has ALREADY enqueued -- with
      // two, scan t could only be submitted once t-2 had been collected, ~0.12 ms before its kernel was due, and a
      // host that was late (a busy box: 0.80 ms per scan measured) left the device waiting.
      // Measured and dropped (profiles/r03_lane_stream_arrangements.log, r02 notes): every streaming kernel on ONE
      // stream and the tail kernels on a second one behind their end events (0.75-0.79 ms per scan: back-to-back
      // kernels of one stream leave ~17 us between them, and nothing overlaps a kernel's drain); streaming kernels
      // alternating on two streams with the tails on a third (0.73-0.75); a scan's filter waiting for the previous
      // scan's "filter done" event (0.77-0.91); holding scan t back until the streaming kernel of t-1 is 60 .. 95 %
      // through its rounds (0.725-0.76); a gate in front of the second scan of a burst (no measurable difference).
      // What did help in round 3: a tail kernel that fits beside a streaming kernel (mm_scan_tail2, 61 VGPRs) on a
      // small grid (512 workgroups) -- 0.71-0.735 ms per scan over the first 20 scans from an empty pipeline over
      // boxes and runs (round 2: 0.745-0.76).  Round 4: the grouped tail (80 / 96 VGPRs) on 1024 workgroups, 0.698-0.707.
      settle_lane_timing(c, lane);             // (before the lane's events are recorded again)
      std::copy(c->lane_ev[lane], c->lane_ev[lane] + 3, p.ev);
      // (filter + tail kernel, the end polled in the lane's own pinned block; never the single-launch kernel:
      // its grid barrier wants the device to itself)
      rc = enqueue_pipeline(c, w, lane_st, p.ev, g, *plan, fc, false, base_offset, p.max_candidates, nullptr, true, false, tail_st,
                            mm::tuning().lane_tail_blocks);
      if (rc != MMH_OK) {
         return rc;
      }
   }
   p.active = true;
   *ticket = c->next_ticket++;
   return MMH_OK;
}

int collect_impl(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count, bool *unsettled, bool *overflow = nullptr);
} // namespace

extern "C" int mmh_scan_collect(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   return collect_impl(c, ticket, out, cap, out_count, nullptr);
}

namespace {
// unsettled (tickets of scan_split only): set when the lane could not settle its part -- nothing is rescanned here then;
// *overflow: ... because the part's bucketed store overflowed (narrower buckets = smaller parts may still do)
int collect_impl(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count, bool *unsettled, bool *overflow)
{
   if (!c || !out_count || (!out && cap)) {
      mmh_set_error("mmh_scan_collect: bad argument");
      return MMH_E_ARG;
   }
   *out_count = 0;
   const int lane = ((ticket % mmh_ctx::kLanes) + mmh_ctx::kLanes) % mmh_ctx::kLanes;
   MmPending &p = c->pending[lane];
   if (!p.active || p.ticket != ticket) {
      mmh_set_error("mmh_scan_collect: ticket %d is not outstanding", ticket);
      return MMH_E_STATE;
   }
   HIP_TRY(hipSetDevice(c->device));
   bool rescan = p.needs_rescan;
   Outcome oc;
   MmWorkspace &w = c->ws[1 + lane];
   if (!rescan) {
      const MmGeom g = scan_geometry(c, &p.plan, p.block_bytes, p.big_endian, &p);
      // (a second phase, if any, goes behind whatever later scans have been enqueued on the ticket's stream: it
      // works on this ticket's own workspace)
      const hipStream_t lane_st = c->pending_tail_stream[lane];
      int rc = finish_pipeline(c, w, lane_st, p.ev, g, p.plan, p.base_offset, p.max_candidates, false, &oc, p.view);
      if (rc != MMH_OK) {
         p.active = false;
         return rc;
      }
      if (overflow) {
         // (more candidates than the lane takes: narrower parts hold fewer, like narrower buckets)
         *overflow = oc.bucket_overflow || oc.candidates > w.out_cap || oc.candidates > oc.limit;
      }
      rescan = oc.candidates > w.out_cap || oc.candidates > oc.limit || oc.hard_overflow || !oc.sorted_on_device;
      static const bool lane_trace = getenv("MMOORE_LANE_TRACE") != nullptr;     // development: where the lanes' kernels lie in time
      if (lane_trace && c->timing) {
         static hipEvent_t base = nullptr;
         if (!base && hipEventCreate(&base) == hipSuccess) {
            (void)hipEventRecord(base, lane_st);
            (void)hipEventSynchronize(base);
         }
         float t0 = 0, t1 = 0, t2 = 0, own = 0;
         (void)hipEventSynchronize(p.ev[2]);
         (void)hipEventElapsedTime(&t0, base, p.ev[0]);
         (void)hipEventElapsedTime(&t1, base, p.ev[1]);
         (void)hipEventElapsedTime(&t2, base, p.ev[2]);
         (void)hipEventElapsedTime(&own, p.ev[0], p.ev[1]);
         fprintf(stderr, "lane %d ticket %d: streaming kernel dispatched %.1f us, ended %.1f us, tail kernel ended %.1f us after the first collect; streaming kernel ran %.1f us\n",
                 lane, ticket, t0 * 1e3, t1 * 1e3, t2 * 1e3, own * 1e3);
      }
      // the lane's timings enter the history: now when its last event has completed, else a little later
      const int slot = (int)(c->scans_recorded % mmh_ctx::kRing);
      c->ring_is_ms[slot] = true;
      c->ring_parts[slot] = 0;
      c->ring_ms[slot][0] = c->ring_ms[slot][1] = 0;
      c->lane_timing_owed[lane] = c->timing ? (int64_t)c->scans_recorded : -1;      // (no start event: the entry stays at 0)
      c->scans_recorded++;
      if (hipEventQuery(p.ev[2]) == hipSuccess) {
         settle_lane_timing(c, lane);
      }
   }
   if (rescan && p.view) {
      p.active = false;                         // (a part of scan_split: the caller falls back to one scan of the whole ROM)
      if (unsettled) {
         *unsettled = true;
      }
      return MMH_OK;
   }
   if (rescan) {
      // (the ticket stays outstanding when the caller's buffer turns out too small: collect again)
      int rc = mmh_scan(c, &p.plan, p.block_bytes, p.big_endian, p.base_offset, out, cap, out_count);
      if (rc != MMH_E_CAPACITY) {
         p.active = false;
      }
      return rc;
   }
   c->counters[0] = oc.candidates;
   c->counters[1] = oc.matches;
   c->counters[2] = oc.tiles;
   c->counters[3] = oc.hard ? 2 : 0;
   *out_count = oc.matches;
   if (oc.matches > cap) {
      mmh_set_error("mmh_scan_collect: %llu matches do not fit the caller's buffer of %llu (collect again)",
                    (unsigned long long)oc.matches, (unsigned long long)cap);
      return MMH_E_CAPACITY;                    // results stay in the lane's pinned block
   }
   std::memcpy(out, w.h_result + kHeaderWords, oc.matches * sizeof(uint64_t));
   p.active = false;
   // (mmh_gather_start(NULL, 0) sends this ticket's list from the lane's device-side copy)
   c->mg.last_src = w.d_result[w.result_turn];
   c->mg.last_end = p.ev[2];
   c->mg.last_count = oc.matches;
   c->mg.last_slots = oc.candidates;
   c->mg.last_list.clear();
   return MMH_OK;
}
} // namespace

namespace {

// ---- big ROMs: one synchronous scan as a pipeline of parts ------------------------------------------------------------
//
// A synchronous scan of a big ROM spends its last tens of microseconds -- hundreds with tens of thousands of candidates --
// behind the streaming kernel: the tail kernel (20 us at 4 K candidates, 0.16 ms at 250 K) and the host's share (reading
// freshly written pinned memory, validating, copying out), all of it while the device streams nothing.  So mmh_scan
// cuts a ROM of >= 1 GiB in HBM (engine semantics) into block-aligned parts -- the multi-GPU partition rule: whole blocks
// plus (L - 1) S bytes of overlap, so the concatenated lists ARE the whole ROM's list -- and sends them through the
// submit lanes: part k's tail kernel and host work run while part k + 1 streams, consecutive streaming kernels overlap
// on the lanes' two streams, and what is left in the open is the last part's tail.
//
// Every scan decides by itself, from its own parts -- nothing is remembered from one scan to the next (rounds 4 and 5
// kept memos keyed on plan + ROM: "sparse", "floods the usual parts", "floods"; a ROM hacker scans a keyword once, and a
// first scan cost 1.2 to 15 times a later one):
//   1. the first two parts are an eighth and three eighths of the ROM; by the time the first one is collected its
//      candidate count tells what the search is like, and the rest goes as ONE part (sparse: C2's 4223 candidates -- every
//      part costs ~10 us of launches and ramp), as two (tens of thousands of candidates) or in eighths (hundreds of
//      thousands: th*s, 251 K candidates per 4 GiB, 1.33 -> 1.06 ms);
//   2. a part whose bucketed store overflows (a flood: more than 4096 candidates in one bucket -- 1 MiB of a 4 GiB part)
//      ends stage 1 THERE: what the parts in front of it delivered stays, and the ROM from that part on goes in parts of a
//      sixteenth (narrower buckets: 'the' on a text-like ROM, 0.76 M matches, through the candidate path);
//   3. where those overflow as well -- or a part needs what the lanes do not run -- the REST of the ROM is one synchronous
//      scan (scan_impl on a view): the forward engine at once after an overflow, its usual route otherwise.
// Parts are collected in ROM order, so a stage's good parts are a prefix of the list.
bool split_applies(const mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian)
{
   (void)big_endian;
   if ((routes_off(c) & MMH_ROUTE_NO_SPLIT) || c->engine != 0 || block_bytes == 0 || !c->rom || c->rom == c->rom_host ||
       c->rom_bytes < kSplitMinBytes || (block_bytes & 15) != 0 || block_bytes > kSplitUnitMin || plan->L > MM_RESOLVER_MAX_KEYWORD) {
      return false;
   }
   for (const MmPending &q : c->pending) {
      if (q.active) {
         return false;                              // the caller has tickets of its own outstanding
      }
   }
   mm::FilterChoice fc;
   return mm::choose_filter(*plan, &fc);            // (no SWAR key: the forward engine's)
}

// what a split scan has so far: the blocks [0, done_blocks) are settled, their `total` offsets at the head of the caller's buffer
struct SplitProgress {
   uint64_t done_blocks = 0, total = 0, candidates = 0, tiles = 0;
   uint32_t parts = 0;
   bool hard = false;
};

// One stage of the pipeline: parts of `unit` blocks from pg->done_blocks on (adaptive: see 1. above) until the ROM's end
// or the first part that does not settle; *overflowed: that part's bucketed store overflowed.  Returns an error only for
// failures of the device / arguments; a part that does not settle just ends the stage (pg says how far it got).
// probe: the stage's first part goes alone -- behind a part that flooded the next one may well flood too, and parts in
// flight behind a part that does not settle are scanned for nothing.
int split_stage(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, uint64_t *out,
                uint64_t cap, SplitProgress *pg, uint64_t unit, bool adaptive, bool *overflowed, bool probe = false, uint64_t stop_block = ~0ull)
{
   *overflowed = false;
   const uint64_t N = c->rom_bytes, S = plan->elem_bytes;
   const uint64_t nblocks = std::min<uint64_t>((N + block_bytes - 1) / block_bytes, stop_block);    // (stop_block: the stage ends there)
   const uint64_t overlap = (uint64_t)(plan->L - 1) * S;
   int tickets[mmh_ctx::kLanes];
   uint64_t ends[mmh_ctx::kLanes];                  // the block behind ticket k's part
   int outstanding = 0;
   uint64_t next_block = pg->done_blocks, step = unit, collected = 0, submitted = 0, good = 0;
   bool failed = false;
   int error = MMH_OK;
   auto collect_oldest = [&]() {
      uint64_t n = 0;
      bool unsettled = false, overflow = false;
      // (behind a part that did not settle the later ones are only taken off their lanes: the next stage scans them again)
      const uint64_t room = failed ? 0 : (pg->total <= cap ? cap - pg->total : 0);
      uint64_t nowhere = 0;                         // (no room left: the part is only counted)
      c->device_idle_hint = next_block >= nblocks;       // (everything is submitted: what is still collected lies in the open)
      int rc = collect_impl(c, tickets[0], room ? out + pg->total : &nowhere, room, &n, &unsettled, &overflow);
      c->device_idle_hint = false;
      if (rc == MMH_E_CAPACITY) {
         // (the part's list is in its lane's block; only the count matters now: the caller comes back with more room)
         c->pending[((tickets[0] % mmh_ctx::kLanes) + mmh_ctx::kLanes) % mmh_ctx::kLanes].active = false;
         rc = MMH_OK;
      }
      const uint64_t end = ends[0];
      for (int k = 1; k < outstanding; k++) {
         tickets[k - 1] = tickets[k];
         ends[k - 1] = ends[k];
      }
      outstanding--;
      if (rc != MMH_OK) {
         error = error == MMH_OK ? rc : error;
         failed = true;
         return;
      }
      if (failed) {
         return;
      }
      if (unsettled) {
         *overflowed = overflow;
         failed = true;
         return;
      }
      good++;
      pg->total += n;
      pg->done_blocks = end;
      pg->candidates += c->counters[0];
      pg->tiles += c->counters[2];
      pg->hard = pg->hard || c->counters[3] == 2;
      if (adaptive && collected++ == 0) {
         // What the search is like, from the first eighth: the second half of the ROM in one part, in two, or in eighths.
         // (thresholds in candidates per unit; 4 GiB: < 2 K = 16 K per ROM: one; < 25 K = 200 K per ROM: two)
         const uint64_t per_unit = c->counters[0];
         const uint64_t left = nblocks > next_block ? nblocks - next_block : 0;
         step = per_unit < 2048 ? left : per_unit < 25600 ? (left + 1) / 2 : unit;
         step = std::max<uint64_t>(step, 1);
      }
   };
   while (next_block < nblocks && !failed) {
      if (outstanding == mmh_ctx::kLanes || (adaptive && collected == 0 && outstanding == 2) || (probe && good == 0 && outstanding == 1)) {
         collect_oldest();                          // (adaptive: the third part waits for the first one's verdict)
         if (failed) {
            break;
         }
      }
      // adaptive: an eighth first (its verdict comes early: the second part has six of a CU's seven wave slots only once
      // the first has ended), three eighths beside it (the device is busy while the host reads the verdict and decides)
      const uint64_t width = !adaptive ? step : submitted == 0 ? unit : submitted == 1 ? 3 * unit : step;
      submitted++;
      const uint64_t b0 = next_block, b1 = std::min(nblocks, next_block + width);
      const uint64_t first = b0 * block_bytes;
      const uint64_t bytes = std::min((b1 - b0) * block_bytes + overlap, N - first);
      int t = 0;
      const int rc = submit_impl(c, plan, block_bytes, big_endian, base_offset + first, &t, true, first, bytes);
      if (rc != MMH_OK) {
         error = rc;
         failed = true;
         break;
      }
      tickets[outstanding] = t;
      ends[outstanding++] = b1;
      next_block = b1;
      pg->parts++;
   }
   while (outstanding) {
      collect_oldest();                             // (also behind a failure: no ticket stays outstanding)
   }
   for (int lane = 0; lane < mmh_ctx::kLanes; lane++) {
      settle_lane_timing(c, lane);
   }
   return error;
}

int scan_split(mmh_ctx *c, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian, uint64_t base_offset, uint64_t *out,
               uint64_t cap, uint64_t *out_count)
{
   *out_count = 0;
   const uint64_t N = c->rom_bytes;
   const uint64_t nblocks = (N + block_bytes - 1) / block_bytes;
   const uint64_t overlap = (uint64_t)(plan->L - 1) * plan->elem_bytes;
   // an eighth of the ROM, but no less than 256 MiB (a 1 GiB ROM: quarters)
   const uint64_t unit = std::max<uint64_t>(std::max<uint64_t>((nblocks + 7) / 8, (kSplitUnitMin + block_bytes - 1) / block_bytes), 1);
   // ... and half of that for the ROM behind a flood (a sixteenth, at least 64 MiB: a part's buckets are a 4096th of it wide)
   const uint64_t fine = std::max<uint64_t>(unit / 2, ((64ull << 20) + block_bytes - 1) / block_bytes);
   const uint64_t first_recorded = c->scans_recorded;
   const auto t_start = std::chrono::steady_clock::now();
   SplitProgress pg;
   bool overflowed = false;
   int rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, unit, true, &overflowed);
   uint64_t path = 0;
   std::vector<uint64_t> rest_list;                 // (what scan_impl keeps of a list that only exists on the host: not needed here)
   // [first block, behind the last) in one synchronous scan, its list behind what the parts delivered
   auto scan_rest = [&](uint64_t b0, uint64_t b1, bool dense) {
      MmPending view;
      view.view = true;
      view.view_first = b0 * block_bytes;
      view.view_bytes = std::min((b1 - b0) * block_bytes + overlap, N - view.view_first);
      const uint64_t room = pg.total <= cap ? cap - pg.total : 0;
      uint64_t nowhere = 0, n = 0;
      bool on_device = false;
      int r = scan_impl(c, plan, block_bytes, big_endian, base_offset + view.view_first, room ? out + pg.total : &nowhere, room, &n, &rest_list,
                        &on_device, &view, dense);
      if (r == MMH_E_CAPACITY) {
         r = MMH_OK;                                // (counted below against the caller's whole buffer)
      }
      pg.total += n;
      pg.candidates += c->counters[0];
      pg.tiles += c->counters[2];
      path = std::max<uint64_t>(path, c->counters[3]);
      pg.parts++;
      pg.done_blocks = b1;
      return r;
   };
   // (stage 1 above; from here on: behind every flood the coarse parts again)
   int floods = 0;                                  // parts that went to the forward engine on their own
   while (rc == MMH_OK && pg.done_blocks < nblocks) {
      if (!overflowed) {
         // a part needs what the lanes do not run (left-overs beyond the resolvers, ...): the rest of the ROM the usual way
         rc = scan_rest(pg.done_blocks, nblocks, false);
         break;
      }
      // The coarse part at done_blocks flooded: ITS extent in parts half as wide (narrower buckets), the first one alone.
      const uint64_t coarse_end = std::min(nblocks, pg.done_blocks + unit);
      bool other = false;                           // a fine part failed for another reason than a flood
      while (rc == MMH_OK && pg.done_blocks < coarse_end && !other) {
         if (fine < unit) {
            rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, fine, false, &overflowed, true, coarse_end);
            if (rc != MMH_OK || pg.done_blocks >= coarse_end) {
               break;
            }
            if (!overflowed) {
               other = true;
               break;
            }
         }
         // The part at done_blocks floods the narrow buckets as well (padding that matches the keyword wholesale: a few MiB
         // of a ROM): the forward engine on THAT part -- 256 MiB of a 4 GiB ROM, not all of it -- and the candidate path
         // again behind it.  A ROM that floods everywhere (a two-symbol keyword) stops being asked after two such parts.
         floods++;
         rc = scan_rest(pg.done_blocks, floods > 2 ? nblocks : std::min(nblocks, pg.done_blocks + std::min(fine, unit)), true);
      }
      if (rc != MMH_OK || pg.done_blocks >= nblocks) {
         break;
      }
      if (other) {
         overflowed = false;
         continue;                                  // (-> the rest of the ROM the usual way)
      }
      // behind the flooded part: coarse parts again, the adaptive way (the first one alone: it may flood as well)
      rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, unit, true, &overflowed, true);
   }
   path = path ? path : (pg.hard ? 2 : 0);
   // The parts' timings as ONE entry of the history: [streaming kernels of all parts, summed -- they overlap, so the sum
   // exceeds their share of the wall time --, the scan's wall time on the host].
   if (c->scans_recorded > first_recorded && c->scans_recorded - first_recorded <= mmh_ctx::kRing) {
      float filter_sum = 0;
      for (uint64_t k = first_recorded; k < c->scans_recorded; k++) {
         float t[4] = {0, 0, 0, 0};
         scan_timings(c, k, t);
         filter_sum += t[0];
      }
      const int slot = (int)(first_recorded % mmh_ctx::kRing);
      c->ring_is_ms[slot] = true;
      c->ring_ms[slot][0] = filter_sum;
      c->ring_ms[slot][1] = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() * 1e3);
      c->ring_parts[slot] = pg.parts;
      c->scans_recorded = first_recorded + 1;
   }
   if (rc != MMH_OK) {
      return rc;
   }
   *out_count = pg.total;
   c->counters[0] = pg.candidates;
   c->counters[1] = pg.total;
   c->counters[2] = pg.tiles;
   c->counters[3] = path;
   // the list exists in the caller's buffer only (a gather that wants it: from the host)
   c->mg.last_src = nullptr;
   c->mg.last_end = nullptr;
   c->mg.last_slots = 0;
   c->mg.last_count = pg.total;
   c->mg.last_list.clear();
   if (pg.total > cap) {
      mmh_set_error("mmh_scan: %llu matches do not fit the caller's buffer of %llu", (unsigned long long)pg.total, (unsigned long long)cap);
      return MMH_E_CAPACITY;
   }
   if (c->mg.comm) {
      c->mg.last_list.assign(out, out + pg.total);
   }
   return MMH_OK;
}

} // namespace

namespace {
// timings of scan number k (it must still be in the ring)
void scan_timings(mmh_ctx *c, uint64_t k, float *ms4)
{
   const int slot = (int)(k % mmh_ctx::kRing);
   hipEvent_t *e = c->ring[slot];
   ms4[0] = ms4[1] = ms4[2] = ms4[3] = 0;
   if (c->ring_is_ms[slot]) {
      ms4[0] = c->ring_ms[slot][0];
      ms4[3] = c->ring_ms[slot][1];
   }
   else if (!c->ring_timed[slot]) {
      return;                                     // (mmh_set_timing(0): no start event, nothing to measure from)
   }
   else {
      // (a scan's end shows in pinned memory a few microseconds before its last event completes: wait for it;
      // an event that was never recorded leaves its figure at 0)
      hipError_t err = hipEventElapsedTime(&ms4[3], e[0], e[2]);
      if (err == hipErrorNotReady && hipEventSynchronize(e[2]) == hipSuccess) {
         err = hipEventElapsedTime(&ms4[3], e[0], e[2]);
      }
      if (err != hipSuccess) {
         (void)hipGetLastError();
         ms4[3] = 0;
      }
      if (c->ring_has_filter[slot] && hipEventElapsedTime(&ms4[0], e[0], e[1]) != hipSuccess) {
         ms4[0] = 0;
      }
      if (!c->ring_has_filter[slot]) {
         ms4[0] = c->ring_filter_ms[slot];        // fused scan: the kernel's own stamps
      }
   }
   ms4[1] = ms4[3] - ms4[0];                    // everything behind the streaming kernel
   if (c->ring_parts[slot]) {
      ms4[1] = 0;                               // (a split scan: the parts' kernels overlap, see include/mmoore_hip.h)
      ms4[2] = (float)c->ring_parts[slot];
   }
}
} // namespace

extern "C" int mmh_last_timings(mmh_ctx *c, float *ms4)
{
   if (!c || !ms4) {
      mmh_set_error("mmh_last_timings: bad argument");
      return MMH_E_ARG;
   }
   if (c->scans_recorded == 0) {
      std::memset(ms4, 0, 4 * sizeof(float));
      return MMH_OK;
   }
   for (int lane = 0; lane < mmh_ctx::kLanes; lane++) {
      settle_lane_timing(c, lane);
   }
   scan_timings(c, c->scans_recorded - 1, ms4);
   return MMH_OK;
}

extern "C" int mmh_timing_history(mmh_ctx *c, float *filter_ms, float *total_ms, int cap, int *count)
{
   if (!c || !filter_ms || !total_ms || !count || cap < 0) {
      mmh_set_error("mmh_timing_history: bad argument");
      return MMH_E_ARG;
   }
   for (int lane = 0; lane < mmh_ctx::kLanes; lane++) {
      settle_lane_timing(c, lane);
   }
   const uint64_t have = std::min<uint64_t>(c->scans_recorded, mmh_ctx::kRing);
   const int n = (int)std::min<uint64_t>(have, (uint64_t)cap);
   for (int i = 0; i < n; i++) {
      float t[4];
      scan_timings(c, c->scans_recorded - n + i, t);
      filter_ms[i] = t[0];
      total_ms[i] = t[3];
   }
   *count = n;
   return MMH_OK;
}

extern "C" int mmh_filter_shape(const mmh_plan_desc *plan, uint32_t *info12)
{
   if (!plan || !info12 || plan->L < 2 || plan->L > MMH_MAX_KEYWORD || (plan->elem_bytes != 1 && plan->elem_bytes != 2)) {
      mmh_set_error("mmh_filter_shape: bad argument");
      return MMH_E_ARG;
   }
   std::memset(info12, 0, 12 * sizeof(uint32_t));
   mm::FilterChoice fc;
   if (mm::choose_filter(*plan, &fc)) {
      if (!mm::shape_known(plan->elem_bytes, fc.shape)) {
         mmh_set_error("mmh_filter_shape: no streaming kernel of the chosen shape");
         return MMH_E_STATE;
      }
      info12[0] = fc.ncond; info12[1] = fc.iA; info12[2] = fc.shape;
      info12[3] = mm::filter_verifies(*plan, fc) ? 1u : 0u;
      for (uint32_t k = 0; k < fc.ncond; k++) {
         info12[4 + 2 * k] = fc.pos[k];
         info12[5 + 2 * k] = fc.gap[k];
      }
   }
   return MMH_OK;
}

extern "C" int mmh_last_counters(mmh_ctx *c, uint64_t *c4)
{
   if (!c || !c4) {
      mmh_set_error("mmh_last_counters: bad argument");
      return MMH_E_ARG;
   }
   std::memcpy(c4, c->counters, sizeof(c->counters));
   return MMH_OK;
}

// ---- route health: run-time switches, what the validation saw, the first-use self-test ------------------------------

extern "C" int mmh_set_route(mmh_ctx *c, uint32_t mask)
{
   if (!c || (mask & ~31u)) {
      mmh_set_error("mmh_set_route: bad argument");
      return MMH_E_ARG;
   }
   c->route_off = mask;
   return MMH_OK;
}

extern "C" int mmh_debug_inject(mmh_ctx *c, uint32_t kind)
{
   if (!c || kind > 5) {
      mmh_set_error("mmh_debug_inject: bad argument");
      return MMH_E_ARG;
   }
   c->health.inject = kind;
   return MMH_OK;
}

extern "C" int mmh_health(mmh_ctx *c, uint64_t *h16)
{
   if (!c || !h16) {
      mmh_set_error("mmh_health: bad argument");
      return MMH_E_ARG;
   }
   std::memset(h16, 0, 16 * sizeof(uint64_t));
   h16[0] = c->health.fallback_reason;
   h16[1] = c->health.fallbacks;
   h16[2] = c->health.late_slots;
   h16[3] = c->health.last_reason;
   h16[4] = routes_off(c);
   h16[5] = g_routes_off.load();
   h16[6] = c->device < kSelftestDevices ? (uint64_t)g_selftest_state[c->device].load() : 0;
   h16[7] = c->health.validated;
   if (c->ws[0].h_result) {
      std::memcpy(h16 + 8, c->ws[0].h_result, 8 * sizeof(uint64_t));
   }
   return MMH_OK;
}

namespace {

// The known answer: 4133 bytes of splitmix64 noise with the keyword's shape ("abcde": four deltas of +1) planted at
// the start, across block boundaries (1020, 2044, 3068, 4092), behind a constant run, in the last bytes, twice back to
// back, off the reference's skip chain and once modulo 256 (which the reference's signed compare does not report);
// blocks of 1024 bytes.  What the reference reports is the constant below (tests/test_oracle.py holds it against the
// oracle and the compiled reference).
constexpr uint64_t kKatBytes = 4133, kKatBlock = 1024;
// (15 plants, 12 reported: 1505 and 2050 are not on the reference's skip chain, 3000 wraps around 0xFF)
const uint64_t kKatExpected[] = {0, 16, 600, 1020, 1028, 1500, 2044, 2596, 3068, 3500, 4092, 4128};

void kat_rom(uint8_t *rom)
{
   uint64_t x = 0x6d6d6f6f72653432ull;
   for (uint64_t i = 0; i < kKatBytes; i += 8) {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      for (uint64_t k = 0; k < 8 && i + k < kKatBytes; k++) {
         rom[i + k] = (uint8_t)(z >> (8 * k));
      }
   }
   std::memset(rom + 2500, 0x41, 96);                        // a constant run: the chain crosses it in default skips
   const struct { uint32_t at; uint8_t base; } plants[] = {
      {0, 0x30}, {16, 0x61}, {600, 0x11}, {1020, 0x10}, {1028, 0xF0}, {1500, 0x00}, {1505, 0x20}, {2044, 0x77}, {2050, 0x22},
      {2596, 0x41}, {3000, 0xFD}, {3068, 0x05}, {3500, 0x90}, {4092, 0x80}, {4128, 0x33}};
   for (const auto &p : plants) {
      for (uint32_t k = 0; k < 5; k++) {
         rom[p.at + k] = (uint8_t)(p.base + k);             // (0xFD: wraps -- matches modulo 256 only)
      }
   }
}

mmh_plan_desc kat_plan()
{
   const uint32_t kw[5] = {'a', 'b', 'c', 'd', 'e'};
   mmh_plan_desc plan;
   std::memset(&plan, 0, sizeof plan);
   (void)mmh_plan_relative(1, kw, 5, 0, nullptr, 0, &plan);
   return plan;
}

// one scan of the KAT on context t with `mask` routes off; through mmh_scan or through the submit lanes
bool kat_scan(mmh_ctx *t, const uint8_t *rom, const mmh_plan_desc &plan, uint32_t mask, int engine, bool lanes, std::string *why)
{
   constexpr uint64_t n_expected = sizeof(kKatExpected) / sizeof(kKatExpected[0]);
   uint64_t got[64] = {0}, n = 0;
   t->route_off = mask;
   t->engine = engine;
   const uint64_t fallbacks = t->health.fallbacks;
   int rc = mmh_rom_upload(t, rom, kKatBytes);
   if (rc == MMH_OK) {
      if (lanes) {
         int ticket = 0;
         rc = mmh_scan_submit(t, &plan, kKatBlock, 0, 0, &ticket);
         if (rc == MMH_OK) {
            rc = mmh_scan_collect(t, ticket, got, 64, &n);
         }
      }
      else {
         rc = mmh_scan(t, &plan, kKatBlock, 0, 0, got, 64, &n);
      }
   }
   t->engine = 0;
   char buf[256];
   if (rc != MMH_OK) {
      snprintf(buf, sizeof buf, "routes off 0x%x, engine %d%s: error %d (%s)", mask, engine, lanes ? ", lanes" : "", rc, mmh_last_error());
      *why = buf;
      return false;
   }
   if (t->health.fallbacks != fallbacks) {
      snprintf(buf, sizeof buf, "routes off 0x%x%s: the published block failed validation (reason %llu)", mask, lanes ? ", lanes" : "",
               (unsigned long long)t->health.last_reason);
      *why = buf;
      return false;
   }
   if (n != n_expected || std::memcmp(got, kKatExpected, n * sizeof(uint64_t)) != 0) {
      uint64_t first = 0;
      while (first < n && first < n_expected && got[first] == kKatExpected[first]) {
         first++;
      }
      snprintf(buf, sizeof buf, "routes off 0x%x, engine %d%s: %llu offsets instead of %llu, first difference at entry %llu (%llu)", mask, engine,
               lanes ? ", lanes" : "", (unsigned long long)n, (unsigned long long)n_expected, (unsigned long long)first,
               (unsigned long long)(first < n ? got[first] : 0));
      *why = buf;
      return false;
   }
   return true;
}

} // namespace

extern "C" int mmh_selftest_kat(uint8_t *rom, uint64_t rom_cap, uint64_t *rom_bytes, uint64_t *expected, uint64_t expected_cap,
                                uint64_t *expected_count)
{
   constexpr uint64_t n_expected = sizeof(kKatExpected) / sizeof(kKatExpected[0]);
   if (!rom || !rom_bytes || !expected_count || (!expected && expected_cap) || rom_cap < kKatBytes) {
      mmh_set_error("mmh_selftest_kat: bad argument (the ROM takes %llu bytes)", (unsigned long long)kKatBytes);
      return MMH_E_ARG;
   }
   kat_rom(rom);
   *rom_bytes = kKatBytes;
   *expected_count = n_expected;
   if (expected_cap < n_expected) {
      return MMH_E_CAPACITY;
   }
   std::memcpy(expected, kKatExpected, sizeof(kKatExpected));
   return MMH_OK;
}

extern "C" int mmh_selftest_run(int device, uint32_t *routes_off_out)
{
   if (!routes_off_out) {
      mmh_set_error("mmh_selftest_run: null argument");
      return MMH_E_ARG;
   }
   *routes_off_out = 0;
   mmh_ctx *t = nullptr;
   int rc = create_context(device, &t);
   if (rc != MMH_OK) {
      return rc;
   }
   std::vector<uint8_t> rom(kKatBytes);
   kat_rom(rom.data());
   const mmh_plan_desc plan = kat_plan();
   std::string why;
   // the plain kernels and the sequential chain kernel must know the answer: everything else is measured against them
   if (!kat_scan(t, rom.data(), plan, 15, 0, false, &why) || !kat_scan(t, rom.data(), plan, 15, 1, false, &why)) {
      mmh_destroy(t);
      mmh_set_error("self-test: %s", why.c_str());
      return MMH_E_DEVICE;
   }
   // the fast routes, all on first; then with more and more of them switched off until the answer is right
   static const bool trace = getenv("MMOORE_SELFTEST_TRACE") != nullptr;
   const uint32_t masks[] = {0, MMH_ROUTE_NO_ZERO_COPY, MMH_ROUTE_NO_SINGLE_LAUNCH, MMH_ROUTE_NO_ZERO_COPY | MMH_ROUTE_NO_SINGLE_LAUNCH,
                             MMH_ROUTE_NO_ZERO_COPY | MMH_ROUTE_NO_SINGLE_LAUNCH | MMH_ROUTE_NO_BUCKETS, 15};
   uint32_t settled = 15;
   for (uint32_t mask : masks) {
      std::string w1;
      if (kat_scan(t, rom.data(), plan, mask, 0, false, &w1) && kat_scan(t, rom.data(), plan, mask, 0, true, &w1)) {
         settled = mask;
         break;
      }
      fprintf(stderr, "libmmoore_hip: self-test on device %d: %s\n", device, w1.c_str());
   }
   if (trace) {
      fprintf(stderr, "libmmoore_hip: self-test on device %d: routes off 0x%x\n", device, settled);
   }
   mmh_destroy(t);
   *routes_off_out = settled;
   return MMH_OK;
}
