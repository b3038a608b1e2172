// mm_context.h -- the device context behind the C ABI's opaque mmh_ctx (private to csrc/)
#ifndef MM_CONTEXT_H
#define MM_CONTEXT_H

#include <hip/hip_runtime.h>

#include <stdint.h>

#include <vector>

#include "mmoore_hip.h"

// pinned staging of mmh_rom_load_file (mm_ingest.hip): two pieces per reader thread
struct MmIngest {
   static constexpr size_t kPiece = 4u << 20;
   std::vector<void *> staging;
   std::vector<hipEvent_t> events;
   std::vector<hipStream_t> streams;
   double last_seconds = 0;
   uint64_t last_bytes = 0;
   int last_threads = 0;
};

// Device buffers of ONE scan in flight (see mm::ResolveBuffers) plus the pinned host block its
// results are published in.
struct MmWorkspace {
   uint64_t *d_cand = nullptr;      // candidate byte offsets
   uint64_t cand_cap = 0;
   uint64_t *d_out = nullptr;       // unordered matches
   uint64_t out_cap = 0;
   unsigned long long *d_ctrl = nullptr;   // counters + arrival tickets, zeroed per scan (mm::ResolveBuffers)
   uint64_t *d_mid_off = nullptr;   // hand-over list mm_resolve -> mm_resolve2
   uint64_t *d_mid_hi = nullptr;
   uint32_t *d_mid_set = nullptr;
   uint32_t *d_mid_slot = nullptr;
   uint64_t *d_hard_off = nullptr;
   uint64_t *d_hard_hi = nullptr;
   uint32_t *d_hard_set = nullptr;
   uint32_t *d_hard_slot = nullptr;
   uint8_t *d_scratch = nullptr;    // tile maps of hard candidates
   uint32_t *d_partials = nullptr;  // rank sort partial counts
   uint64_t *h_result = nullptr;    // pinned: [kHeaderWords counters][kMaxRankSort ordered matches], written by the device
   bool ctrl_clean = false;         // the previous scan's last kernel left d_ctrl zeroed
};

// a scan submitted with mmh_scan_submit and not collected yet
struct MmPending {
   bool active = false;
   bool needs_rescan = false;       // plan / engine choice the lanes do not run: collect scans synchronously
   int ticket = 0;
   mmh_plan_desc plan{};
   uint64_t block_bytes = 0;
   int big_endian = 0;
   uint64_t base_offset = 0;
   uint32_t max_candidates = 0;
   hipEvent_t *ev = nullptr;        // its event triple in the ring
};

struct mmh_ctx {
   int device = 0;
   hipStream_t own_stream = nullptr;
   hipStream_t stream = nullptr;

   uint8_t *rom = nullptr;
   uint64_t rom_bytes = 0;
   uint64_t rom_alloc = 0;          // > 0 when the library owns the buffer

   MmWorkspace ws[3];               // [0] mmh_scan; [1], [2] the two lanes of mmh_scan_submit / _collect
   uint8_t *d_dense = nullptr;      // dense engine: tile maps, super-tile maps, entry phases (one allocation)
   size_t dense_bytes = 0;
   uint64_t *d_sort_in = nullptr;   // long lists: contiguous keys (dense engine), ordered keys, rocPRIM scratch
   uint64_t sort_in_cap = 0;
   uint64_t *d_sort_out = nullptr;
   uint64_t sort_out_cap = 0;
   void *d_sort_tmp = nullptr;
   size_t sort_tmp_bytes = 0;

   // Ring of event triples {scan start, behind the streaming kernel, scan end}: elapsed
   // times are only computed when somebody asks (mmh_last_timings / mmh_timing_history),
   // never on the scan's own critical path.
   static constexpr int kRing = 64;
   hipEvent_t ring[kRing][3] = {};
   bool ring_has_filter[kRing] = {};
   uint64_t scans_recorded = 0;     // slot of scan k is k % kRing
   hipEvent_t *ev = nullptr;        // the current scan's triple
   hipStream_t lane_stream[2] = {nullptr, nullptr};   // streams of the submit lanes
   hipEvent_t lane_fence = nullptr;                     // orders a lane behind earlier work on `stream`
   MmPending pending[2];
   int next_ticket = 0;
   int engine = 0;
   uint64_t counters[4] = {0, 0, 0, 0};
   MmIngest ingest;
};

// defined in mm_capi.hip
int mmh_workspace(mmh_ctx *c);

#endif
