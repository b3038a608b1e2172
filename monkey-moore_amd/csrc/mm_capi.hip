// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi.hip -- the C ABI of include/mmoore_hip.h: device context, ROM
// ownership, and the orchestration of one scan on one HIP stream.
//
// There is no CPU compute path here: every entry point that touches data needs
// a HIP device and fails with MMH_E_DEVICE otherwise.  Host work: the pattern
// plan (mm_plan.cpp), sizing buffers, choosing the engine from the counters a
// scan publishes (second resolver phase, forward engine on flagged domains or
// on everything), and merging lists of different engines in the rare mixed case.
//
// One translation unit in sections (round 6; the file was 2 900 lines):
//   mm_capi.hip           errors, context create / destroy, the device warm-up, ROM entry points, mmh_scan (scan_impl),
//                         timings and the small queries
//   mm_capi_workspace.h   a scan's buffers, bucket stores, prepare_scans (what is set up when the context gets its ROM)
//   mm_capi_validate.h    the checks on a published block
//   mm_capi_pipeline.h    enqueue / wait / second phase of one scan on one workspace
//   mm_capi_engines.h     long lists, forward engine, flood paths
//   mm_capi_lanes.h       mmh_scan_submit / mmh_scan_collect
//   mm_capi_split.h       the pipeline of parts of a big ROM
//   mm_capi_selftest.h    the first-use self-test
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

// MMOORE_TRACE=sync (development): where a synchronous scan's host time goes -- marks taken along mmh_scan, averages
// printed every 256 scans: [0] entry -> launches done, [1] -> flag seen (device time + launch latency), [2] -> block
// validated, [3] -> back at the caller
namespace {
struct SyncTrace {
   bool on = mm_trace("sync");
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

#include "mm_capi_workspace.h"


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

uint64_t zero_copy_limit() { return 512ull << 10; }

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
      // (keywords of 3 / 2 symbols on the candidate path: their streaming kernels are NOT the ones BASELINE's keywords select
      // -- a profile of a caller's run holds no launch of this warm-up under the kernels it is about; the code objects are
      // per translation unit and every kernel has been looked up, preload_kernels)
      for (int engine : {0, 2}) {
         (void)mmh_set_engine(t, engine);
         for (uint32_t elem : {1u, 2u}) {
            for (uint32_t len : {engine == 0 ? 4u - elem : 12u, engine == 0 ? 4u - elem : 65u}) {
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
      if (mmh_plan_relative(1, kw, 3, 0, nullptr, 0, &plan) == MMH_OK) {
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
      // (MMOORE_WARMUP=0: a process that wants its context at once and pays at its first scans instead)
      static const bool warm = [] { const char *e = getenv("MMOORE_WARMUP"); return !(e && *e == '0'); }();
      if (warm) {
         std::call_once(g_warm_once[device], [device] { warm_device(device); });
      }
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

#include "mm_capi_validate.h"
#include "mm_capi_pipeline.h"
#include "mm_capi_engines.h"


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
   const bool narrow = plan->L <= MM_CANDIDATE_MAX_KEYWORD;
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
      static const bool tracing = mm_trace("split");
      const auto t_fetch = std::chrono::steady_clock::now();
      rc = fetch_device_list(c, c->d_sort_out, device_list_n, out, cap, &oc.matches);
      if (rc != MMH_OK) {
         return rc;
      }
      if (tracing) {
         fprintf(stderr, "      %llu ordered slots -> %llu offsets in the caller's buffer: %.1f us (waiting for the sort included)\n", (unsigned long long)device_list_n,
                 (unsigned long long)oc.matches, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fetch).count() * 1e6);
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

#include "mm_capi_lanes.h"


#include "mm_capi_split.h"


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

#include "mm_capi_selftest.h"
