// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_split.h -- big ROMs: one synchronous scan as a pipeline of parts, routed from its own parts.
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

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
       c->rom_bytes < kSplitMinBytes || (block_bytes & 15) != 0 || block_bytes > kSplitUnitMin || plan->L > MM_CANDIDATE_MAX_KEYWORD) {
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
                uint64_t cap, SplitProgress *pg, uint64_t unit, bool adaptive, bool *overflowed, bool probe = false, uint64_t stop_block = ~0ull,
                uint64_t *flood2 = nullptr)
{
   *overflowed = false;
   if (flood2) {
      flood2[0] = flood2[1] = 0;                    // where the part that ended the stage floods (bytes of the ROM; [1] = 0: not known)
   }
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
      uint64_t where[2] = {0, 0};
      int rc = collect_impl(c, tickets[0], room ? out + pg->total : &nowhere, room, &n, &unsettled, &overflow, where);
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
         if (flood2 && overflow) {
            flood2[0] = where[0];
            flood2[1] = where[1];
         }
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
   uint64_t flood[2] = {0, 0};
   // MMOORE_TRACE=split: what every stage of this scan did and when (host clock, us from the scan's start)
   static const bool tracing = mm_trace("split");
   auto note = [&](const char *what, uint64_t b0, uint64_t b1) {
      if (tracing) {
         fprintf(stderr, "   split %8.1f us  %-26s blocks [%llu, %llu)  done %llu of %llu, %llu offsets, %u parts%s\n",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() * 1e6, what, (unsigned long long)b0,
                 (unsigned long long)b1, (unsigned long long)pg.done_blocks, (unsigned long long)nblocks, (unsigned long long)pg.total, pg.parts,
                 overflowed ? (flood[1] ? "  [flood, extent known]" : "  [flood]") : "");
      }
   };
   // A keyword whose deltas all lie on one line -- `aaaa`, `abcd`, `ab*de`: every literal is its left neighbour plus the
   // same step per place -- matches padding and ramps wholesale, and ROMs are full of those: its first part goes alone
   // (what is in flight behind a part that floods is scanned for nothing: three eighths of the ROM).  Read off the plan,
   // not remembered from an earlier scan.
   bool on_a_line = true;
   bool have_step = false;
   int64_t step_num = 0;
   for (uint32_t i = 0; i < plan->L && on_a_line; i++) {
      if (plan->cmp_mask[i] == 0 || plan->bridge[i] >= 0) {
         continue;
      }
      const int64_t places = -(int64_t)plan->bridge[i];
      if (plan->expected[i] % places != 0) {
         on_a_line = false;
      }
      else if (!have_step) {
         have_step = true;
         step_num = plan->expected[i] / places;
      }
      else {
         on_a_line = plan->expected[i] / places == step_num;
      }
   }
   int rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, unit, true, &overflowed, on_a_line && have_step, ~0ull, flood);
   note("stage 1 (adaptive)", 0, nblocks);
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
      note(dense ? "forward engine on" : "one synchronous scan of", b0, b1);
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
      if (flood[1] == 0 && floods == 0) {
         // The first part to flood has more candidates than a scan keeps, and no bucket of it overflowed: they lie all over
         // the part -- a two-symbol keyword, `q*v`: a candidate every 256 bytes of any ROM --, not in a run of padding.
         // Narrower parts would hold as many per byte (and the per-candidate path costs 2.4 ms per million of them, the
         // forward engine 0.7 - 1.9 ms per GiB whatever the data): the rest of the ROM on the forward engine, now.  (Until
         // round 6: parts half as wide, the forward engine on two of them, then the rest -- 4 ms of a two-symbol scan.)
         floods = 3;
         rc = scan_rest(pg.done_blocks, nblocks, true);
         break;
      }
      if (flood[1] != 0 && flood[1] <= fine * block_bytes / 4) {
         // The tail kernel said WHERE the part floods (the overflowing buckets' extent: padding that matches the keyword
         // wholesale is a MiB or two of a ROM): the blocks in front of it on the candidate path, the forward engine on the
         // flooded blocks alone, the coarse parts again behind them.  (A keyword's length of slack on either side: buckets
         // are keyed on a candidate's anchor.)
         const uint64_t f0 = std::max<uint64_t>(pg.done_blocks, (flood[0] > 256 ? flood[0] - 256 : 0) / block_bytes);
         const uint64_t f1 = std::min<uint64_t>(nblocks, (flood[0] + flood[1] + 256 + block_bytes - 1) / block_bytes);
         flood[1] = 0;
         if (f0 > pg.done_blocks) {
            bool again = false;
            rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, std::max<uint64_t>(fine, 1), false, &again, false, f0);
            note("parts up to the flood", pg.done_blocks, f0);
            if (rc != MMH_OK) {
               break;
            }
            if (pg.done_blocks < f0) {
               overflowed = again;                  // (it did not get there: the general way from where it stands)
               if (again) {
                  goto general;
               }
               continue;
            }
         }
         floods++;
         rc = scan_rest(pg.done_blocks, floods > 2 ? nblocks : std::max(f1, pg.done_blocks + 1), true);
         if (rc != MMH_OK || pg.done_blocks >= nblocks) {
            break;
         }
         rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, unit, true, &overflowed, true, ~0ull, flood);
         note("coarse parts again", f1, nblocks);
         continue;
      }
   general:
      // The coarse part at done_blocks flooded: ITS extent in parts half as wide (narrower buckets), the first one alone.
      const uint64_t coarse_end = std::min(nblocks, pg.done_blocks + unit);
      bool other = false;                           // a fine part failed for another reason than a flood
      while (rc == MMH_OK && pg.done_blocks < coarse_end && !other) {
         if (fine < unit) {
            rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, fine, false, &overflowed, true, coarse_end);
            note("finer parts", pg.done_blocks, coarse_end);
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
      rc = split_stage(c, plan, block_bytes, big_endian, base_offset, out, cap, &pg, unit, true, &overflowed, true, ~0ull, flood);
      note("coarse parts again", pg.done_blocks, nblocks);
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
