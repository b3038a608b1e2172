// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_lanes.h -- scans in flight: mmh_scan_submit / mmh_scan_collect on three lanes.
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

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
   if (c->engine != 0 || !have_filter || g.nbytes == 0 || plan->L > MM_CANDIDATE_MAX_KEYWORD) {
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
      // How scans in flight share the device (MMOORE_TRACE=lanes; rocprofv3 kernel trace).  Scan t
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

int collect_impl(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count, bool *unsettled, bool *overflow = nullptr,
                 uint64_t *flood2 = nullptr);
} // namespace

extern "C" int mmh_scan_collect(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   return collect_impl(c, ticket, out, cap, out_count, nullptr);
}

namespace {
// unsettled (tickets of scan_split only): set when the lane could not settle its part -- nothing is rescanned here then;
// *overflow: ... because the part's bucketed store overflowed (narrower buckets = smaller parts may still do)
// flood2 (with overflow): [0] first byte, [1] bytes of the flooded buckets' extent in the ROM (0 bytes: not known)
int collect_impl(mmh_ctx *c, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count, bool *unsettled, bool *overflow, uint64_t *flood2)
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
      if (flood2) {
         flood2[0] = (p.view ? p.view_first : 0) + oc.flood_first;
         flood2[1] = oc.bucket_overflow ? oc.flood_bytes : 0;
      }
      rescan = oc.candidates > w.out_cap || oc.candidates > oc.limit || oc.hard_overflow || !oc.sorted_on_device;
      static const bool lane_trace = mm_trace("lanes");     // development: where the lanes' kernels lie in time
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
