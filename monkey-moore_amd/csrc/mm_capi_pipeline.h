// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_pipeline.h -- one scan on one workspace: enqueue (streaming kernel + tail, or the single-launch kernel, or the event-synchronised chain), wait, second phase.
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.


struct Outcome {
   uint64_t candidates = 0;         // filter survivors (0 on the sequential path)
   uint64_t listed = 0;             // keys in d_out: result slots (fast path) or appended matches (sequential)
   uint64_t matches = 0;            // valid when sorted_on_device
   uint64_t tiles = 0;
   uint32_t hard = 0;
   bool hard_overflow = false;
   bool sorted_on_device = false;
   bool bucket_overflow = false;    // (tickets of scan_split only) the bucketed store overflowed and nothing was run again
   uint64_t flood_first = 0, flood_bytes = 0;   // ... and where: the flooded buckets' extent in the scanned bytes (0 bytes: not known)
   uint32_t limit = 0;              // the candidate limit of the kernels that ran (bucketed store: 2^20, list-based kernels: 2^18)
};

// enqueue [zero counters] -> engine kernels -> ordering into pinned host memory; `ev` = the
// scan's event triple {start, behind the streaming kernel, end}
// One fused scan kernel at a time per process: its grid barrier needs all of its workgroups
// resident, and two such grids in flight could keep each other's stragglers out (mm_fused.h).
// A scan that finds the lock taken simply runs the plain kernels.
std::mutex g_fused_lock;

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
   const bool polled = allow_polled && !sequential && !skip_bits && !(off & MMH_ROUTE_NO_POLLED);
   // (keywords beyond 64 symbols: the single-launch kernel is not instantiated for their two-phases-per-lane resolver)
   const bool single_launch = polled && allow_single_launch && c->fused_ok && !(off & MMH_ROUTE_NO_SINGLE_LAUNCH) && mm::fused_applies(g) &&
                              pl.L <= MM_RESOLVER_MAX_KEYWORD;
   const bool bucketed = polled && !single_launch && !(off & MMH_ROUTE_NO_BUCKETS);
   // Only the bucketed store takes the full limit: the list-based kernels keep round 2's (their lists share d_cand, and the
   // callers read "more candidates than this" as "a flood: take it apart domain by domain").
   if (!bucketed) {
      max_candidates = list_candidate_limit(w, max_candidates);
   }
   w.limit = max_candidates;
   w.mid_listed = ~0ull;
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
      static const bool trace = mm_trace("fused");
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
            static const bool fetch_trace = mm_trace("split");
            const auto t_fetch = std::chrono::steady_clock::now();
            HIP_TRY(hipStreamWaitEvent(c->own_stream, ev[2], 0));
            // Round 5: up to a megabyte comes over by a KERNEL that stores it into the pinned block and raises a word behind
            // the scan's flag, which this thread polls -- hipMemcpyAsync + hipStreamSynchronize took 82-120 us for the 130-200 KiB
            // of a part with 16-24 K matches (the runtime's time, not the DMA engine's).
            bool fetched = false;
            // (only with the device to itself: beside another part's streaming kernel the copy kernel waits for wave slots --
            // 153 us measured -- where the copy engine's 100 us at least overlap the device's work)
            if (c->device_idle_hint && oc->candidates <= 131072) {
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
            static const bool slots_trace = mm_trace("split");
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
         if (w.h_result[1] >> 32 & 1) {
            // (mm_scan_tail2 says where the overflowing buckets are, in units of 16 buckets of this scan's bucket width)
            const uint64_t unit = 16ull << mm::bucket_geom(g.nbytes).shift;
            const uint64_t lo = w.h_result[1] & 0xFFFF, hi = (w.h_result[1] >> 16) & 0xFFFF;
            oc->flood_first = lo * unit;
            oc->flood_bytes = std::min<uint64_t>((hi + 1) * unit, g.nbytes) - oc->flood_first;
         }
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
   if (leftovers <= mm::mid_cap() && pl.L <= MM_RESOLVER_MAX_KEYWORD) {
      w.mid_listed = leftovers;                 // (run_flagged_domains: the domains concerned without another pass over the ROM)
   }
   // (keywords beyond 64 symbols: the first resolver counts what it cannot settle but hands nothing on -- the second
   // phase's stored maps are 64 bytes --: any left-over is "more than the second phase takes", csrc/mm_tiles.h)
   if (pl.L > MM_RESOLVER_MAX_KEYWORD) {
      oc->hard_overflow = true;
   }
   if (oc->hard_overflow || leftovers > mm::mid_cap() || oc->candidates > w.out_cap || oc->candidates > max_candidates) {
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
