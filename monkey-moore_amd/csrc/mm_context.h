// SPDX-License-Identifier: GPL-3.0-or-later
// mm_context.h -- the device context behind the C ABI's opaque mmh_ctx (private to csrc/)
#ifndef MM_CONTEXT_H
#define MM_CONTEXT_H

#include <hip/hip_runtime.h>

#include <stdint.h>

#include <atomic>
#include <memory>
#include <thread>
#include <cstdlib>
#include <string>
#include <vector>

#include "mmoore_hip.h"

// pinned staging of mmh_rom_load_file (mm_ingest.hip): two pieces per reader thread
struct MmIngest {
   static constexpr size_t kPiece = 4u << 20;
   std::vector<void *> staging;
   std::vector<hipEvent_t> events;
   std::vector<hipStream_t> streams;
   bool undrained = false;           // an aborted load left copies in flight on `streams` (mm_ingest_drain)
   std::vector<std::thread> stragglers;   // ... and possibly readers inside a blocking call (joined by mm_ingest_drain)
   std::shared_ptr<std::atomic<int>> straggling;   // how many of them are still at it
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
   uint32_t limit = 0;              // candidate limit of the scan enqueued last (enqueue_pipeline)
   uint64_t mid_listed = ~0ull;     // the scan's first phase handed this many undecided candidates on, all of them in d_mid_off
                                    // (~0: not known / more than the list holds / a keyword whose left-overs are only counted)
   unsigned long long *d_ctrl = nullptr;   // counters + arrival tickets, zeroed per scan (mm::ResolveBuffers)
   uint64_t *d_mid_off = nullptr;   // hand-over list mm_resolve -> mm_resolve2
   uint64_t *d_mid_hi = nullptr;
   uint64_t *d_mid_set = nullptr;
   uint32_t *d_mid_slot = nullptr;
   uint64_t *d_hard_off = nullptr;
   uint64_t *d_hard_hi = nullptr;
   uint64_t *d_hard_set = nullptr;
   uint32_t *d_hard_slot = nullptr;
   uint8_t *d_scratch = nullptr;    // tile maps of hard candidates
   uint32_t *d_partials = nullptr;  // rank sort partial counts
   uint64_t *d_bcand = nullptr;     // bucketed candidate store (big ROMs; mm_internal.h MM_BUCKET_*), allocated on first use
   unsigned int *d_bcount = nullptr;   // its bucket and super-bucket counters, left zeroed by mm_scan_tail2
   bool buckets_clean = false;      // ... unless a scan did not get that far
   bool bucketed = false;           // the scan under way went through the bucketed store
   uint64_t *h_result = nullptr;    // pinned: [kHeaderWords counters][kMaxRankSort ordered matches], written by the device
   // Device-side copies of that block, written by the same kernel: what the multi-GPU offset
   // gather (mm_multi.hip) sends -- the collective never waits for a host round trip.  Two of
   // them, alternating from scan to scan: the gather of scan k may still be reading its copy
   // while scan k+1 publishes (the gather overlaps the next scan).
   uint64_t *d_result[2] = {nullptr, nullptr};
   int result_turn = 0;             // d_result[result_turn] belongs to the most recent scan
   uint64_t seq = 0;                // fused scans: the number the kernel raises in h_result[MM_HDR_FLAG_WORD]
   uint64_t pub_seq = 0;            // ... and mm_publish_list in the word behind it (a long list fetched by a kernel, finish_pipeline)
   bool fused = false;              // the scan under way is one mm_scan_fused launch (and holds the process-wide fused lock)
   bool polled = false;             // the scan under way announces its end in h_result[MM_HDR_FLAG_WORD] (fused or filter + tail)
   float fused_filter_ms = 0;       // its streaming phase, from the kernel's own wall-clock stamps
   float fused_total_ms = 0;        // ... and the whole launch up to its header, from the same stamps
   bool ctrl_clean = false;         // the previous scan's last kernel left d_ctrl zeroed
   uint64_t bcand_buckets = 0;      // buckets d_bcand has room for (sized from the ROM, grown when a larger one arrives)
   // Result slots of h_result that may hold something else than MM_SLOT_POISON (slots a scan published straight into
   // pinned memory, the rank kernels' lists): re-poisoned before the next launch, so that a slot whose PCIe write has
   // not landed when the flag word shows is recognised instead of read as an offset (validate_published, mm_capi.hip)
   uint64_t dirty_slots = 0;
   uint32_t max_rank = 0;           // slots the polled scan under way may publish (MM_MAX_RANK_SORT / MM_MAX_PUBLISH)
};

// What the library knows about the health of its fast routes (mmh_health): every polled scan's published block is
// validated before it is trusted; a violation sends the scan through the plain, event-synchronised kernels and is
// remembered here for good.
struct MmHealth {
   uint64_t fallback_reason = 0;    // MMH_FB_* of the FIRST violation on this context (sticky; 0: none)
   uint64_t fallbacks = 0;          // scans that were rerun through the plain kernels because of a violation
   uint64_t late_slots = 0;         // result slots that showed after the flag word (waited for, not a violation)
   uint64_t last_reason = 0;
   uint64_t validated = 0;          // polled scans whose block went through the validation
   uint32_t inject = 0;             // tests: the next polled scan's block is damaged on the host (mmh_debug_inject)
};

// One offset gather in flight (mm_multi.hip): the all-gather's receive table, the pinned block the
// merged list is packed into, and the events that bracket the collective on the comm stream.
struct MmGatherSlot {
   uint64_t *d_table = nullptr;     // [nranks][kGatherRecordWords]
   uint64_t *h_merged = nullptr;    // pinned: [kGatherHeaderWords + nranks counts][merged offsets]
   uint64_t merged_cap = 0;         // offsets h_merged has room for
   hipEvent_t begin = nullptr, end = nullptr;
   bool busy = false;
   uint32_t limit = 0;              // offsets (or slots) a record of THIS gather holds: 1016 (8 KiB per rank) or 16384 (128 KiB)
   uint64_t last_longest = ~0ull;   // the longest extent over all ranks the slot's previous gather saw (~0: none yet)
   bool from_host = false;          // the local list came from host memory (long lists, forward engine)
   const uint64_t *src = nullptr;   // else: the device-side result copy (of the scan's workspace) it sends
   uint64_t local_count = 0;        // this rank's list length
   // A device-resident list too long for a record (or with more slots than one holds): its published block is copied
   // HERE when the gather starts.  The second phase runs at mmh_gather_finish, by when later scans -- a retry, two scans
   // on, a second outstanding gather -- may have published into the result copy the first phase sent from (they only
   // wait for the first phase's end event).
   uint64_t *d_keep = nullptr;
   uint64_t keep_cap = 0;           // words
   bool kept = false;               // this gather's list lives in d_keep
   std::vector<uint64_t> host_list; // a host-resident list too long for a record: kept HERE for the second phase, so that no
                                    // later scan can take it away (all ranks must enter that phase or none)
   double start_wall_s = 0;         // host time spent in mmh_gather_start
};

// Multi-GPU state of a context: its RCCL communicator and the gather buffers.
struct MmComm {
   void *comm = nullptr;            // ncclComm_t
   int rank = 0, nranks = 1;
   hipStream_t stream = nullptr;    // the collective runs here, behind the scan it belongs to
   MmGatherSlot slot[2];
   int turn = 0;                    // slot of the next mmh_gather_start
   int oldest = 0;                  // slot of the next mmh_gather_finish
   uint64_t *d_send = nullptr;      // record built from a host list: [header][offsets]
   uint64_t *d_long = nullptr;      // second phase (some list longer than a record): [packing header][this rank's list, padded]
   uint64_t long_cap = 0;
   uint64_t *d_long_table = nullptr;
   uint64_t long_table_cap = 0;
   std::vector<uint64_t> last_list; // host copy of the most recent scan's list when it is not device resident
   const uint64_t *last_src = nullptr; // the most recent scan's (or collected ticket's) list sits ordered in this device-side
                                    // result copy of its workspace; null: it only exists on the host (last_list)
   uint64_t last_count = 0;
   uint64_t last_slots = 0;         // result slots of that block (>= last_count: one per candidate when the list has holes)
   hipEvent_t last_end = nullptr;   // end event of the scan that left last_src: a polled scan returns when its flag word shows in
                                    // pinned memory, which may be before its last kernel has retired and its plain stores to
                                    // last_src are visible device-wide -- the gather's stream waits for this event first
   float last_device_ms = 0;        // collective + packing of the last finished gather (HIP events)
   double last_wall_ms = 0;         // host time inside mmh_gather_start + mmh_gather_finish of it
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
   hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // the lane's own event triple {start, behind the filter, end}
   // a ticket of mmh_scan's split pipeline (dense searches, mm_capi.hip: scan_split) scans a VIEW of the ROM: bytes
   // [view_first, view_first + view_bytes); what such a ticket cannot settle on its lane is not rescanned by collect
   bool view = false;
   uint64_t view_first = 0, view_bytes = 0;
};

struct mmh_ctx {
   int device = 0;
   hipStream_t own_stream = nullptr;
   hipStream_t stream = nullptr;

   uint8_t *rom = nullptr;          // the ROM the scans read: rom_own, rom_host, or a borrowed device pointer
   uint64_t rom_bytes = 0;
   uint8_t *rom_own = nullptr;      // library-owned device buffer
   uint64_t rom_alloc = 0;          //   ... and its size
   uint8_t *rom_host = nullptr;     // pinned host buffer small uploads are scanned from in place (zero copy)

   // finish_pipeline: nothing else of this context is at work on the device (a synchronous scan's own wait, the last collects of
   // mmh_scan's pipeline of parts): a long list is fetched by mm_publish_list + a polled word instead of hipMemcpy
   bool device_idle_hint = false;
   bool timing = true;              // mmh_set_timing: scans record their start events (mmh_last_timings works for every scan)
   static constexpr int kLanes = 3; // scans mmh_scan_submit keeps in flight
   MmWorkspace ws[1 + kLanes];      // [0] mmh_scan; [1 ...] the lanes of mmh_scan_submit / _collect
   uint8_t *d_dense = nullptr;      // dense engine: tile maps, super-tile maps, entry phases (one allocation)
   size_t dense_bytes = 0;
   uint64_t *d_sort_in = nullptr;   // long lists: contiguous keys (dense engine), ordered keys, rocPRIM scratch
   uint64_t sort_in_cap = 0;
   // domain bitmap / list / counts of run_flagged_domains and run_candidate_floods.  Its own buffer: the forward
   // engine they call packs its finds into d_sort_in and may reallocate it (a fuzz soak found the domain list
   // read from the freed buffer on the engine's second attempt: 527 of 43538 matches)
   uint64_t *d_domains = nullptr;
   uint64_t domains_cap = 0;
   uint64_t *d_sort_out = nullptr;
   uint64_t sort_out_cap = 0;
   void *d_sort_tmp = nullptr;
   size_t sort_tmp_bytes = 0;
   void *h_ring[2] = {nullptr, nullptr};   // pinned pieces long device lists travel through (fetch_device_list, mm_capi.hip)
   hipEvent_t ring_ev[2] = {nullptr, nullptr};

   // Ring of event triples {scan start, behind the streaming kernel, scan end}: elapsed
   // times are only computed when somebody asks (mmh_last_timings / mmh_timing_history),
   // never on the scan's own critical path.
   static constexpr int kRing = 64;
   hipEvent_t ring[kRing][3] = {};
   bool ring_has_filter[kRing] = {};
   // a slot may instead hold finished numbers: scans of the submit lanes use their own events (a
   // ring slot could be re-recorded by 64 later scans before the lane is collected) and copy
   // their timings in here when they are collected
   float ring_filter_ms[kRing] = {};   // streaming phase of a fused scan (no event marks it inside the one launch)
   bool ring_is_ms[kRing] = {};
   bool ring_timed[kRing] = {};     // the scan recorded its start event (mmh_set_timing): else only what kernels stamped themselves is known
   float ring_ms[kRing][2] = {};    // {streaming kernel, whole scan}
   uint32_t ring_parts[kRing] = {}; // parts a scan ran as when it went through the split pipeline (0: one launch)
   uint64_t scans_recorded = 0;     // slot of scan k is k % kRing
   hipEvent_t *ev = nullptr;        // the current scan's triple
   // The lanes' kernels go to TWO streams, scan t to stream t % 2: scan t then runs behind scan t-2 by stream
   // order, and the context stays within the 4 hardware queues a process gets by default (GPU_MAX_HW_QUEUES):
   // its own stream, these two and the gather's.  Streams that share a hardware queue serialize -- a third lane
   // stream cost 0.706 -> 0.728 ms per scan as soon as a fifth stream existed in the process.
   static constexpr int kLaneStreams = 3;               // (the third only in the experimental lane mode 2)
   hipStream_t lane_stream[kLaneStreams] = {};
   hipStream_t pending_tail_stream[kLanes] = {};        // the stream a lane's tail kernel (and any follow-up work) went to
   hipEvent_t lane_fence = nullptr;                     // orders a lane behind earlier work on `stream`
   hipEvent_t lane_ev[kLanes][3] = {};                  // event triples of the lanes
   int64_t lane_timing_owed[kLanes] = {-1, -1, -1};
   MmPending pending[kLanes];
   int next_ticket = 0;
   int engine = 0;
   bool fused_ok = true;            // cleared for good when a fused scan's grid barrier ever timed out on this context
   uint32_t route_off = 0;          // MMH_ROUTE_* bits switched off on this context (mmh_set_route); the process-wide ones come on top
   MmHealth health;
   uint64_t counters[4] = {0, 0, 0, 0};
   MmIngest ingest;
   MmComm mg;
};

// MMOORE_TRACE: diagnostics on stderr, by topic -- a comma-separated list of sync, split, lanes, fused, floods, ingest,
// selftest, or 1 / all for every one of them.  (One switch: rounds 2-5 had one variable per topic.)
inline bool mm_trace(const char *topic)
{
   static const std::string want = [] { const char *e = getenv("MMOORE_TRACE"); return std::string(e ? e : ""); }();
   if (want.empty() || want == "0") {
      return false;
   }
   if (want == "1" || want == "all") {
      return true;
   }
   return ("," + want + ",").find(std::string(",") + topic + ",") != std::string::npos;
}

// defined in mm_capi.hip
int mmh_workspace(mmh_ctx *c);
// defined in mm_ingest.hip: waits for the copies an aborted mmh_rom_load_file_watched left in flight (no-op otherwise)
int mm_ingest_drain(mmh_ctx *c);

#endif
