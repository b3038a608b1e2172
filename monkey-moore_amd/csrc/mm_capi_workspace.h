// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_workspace.h -- workspaces: the buffers of a scan, bucket stores, and what a context sets up when it gets its ROM (prepare_scans).
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

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
   if (rc == MMH_OK) {
      // the flag pass's domain bitmap + list (run_flagged_domains, run_candidate_floods): blocks of >= 64 KiB, both alignments
      rc = grow(&c->d_domains, &c->domains_cap, 3 * (c->rom_bytes >> 16) + 64);
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
