// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_engines.h -- long lists (radix sort, the way to the host), the forward engine, the flood paths, scan geometry.
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

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
         // (one thread copies out of pinned memory at 8-9 GB/s: a two-symbol keyword's 16 M matches spent 15 ms here.  Whole
         // pieces go in four slices side by side)
         if (valid >= kPiece / 2) {
            constexpr int kSlices = 4;
            std::thread helpers[kSlices - 1];
            const uint64_t each = (valid + kSlices - 1) / kSlices;
            for (int t = 1; t < kSlices; t++) {
               const uint64_t a = std::min<uint64_t>(valid, each * t), b = std::min<uint64_t>(valid, each * (t + 1));
               helpers[t - 1] = std::thread([=] { std::memcpy(dst + kept + a, src + a, (b - a) * sizeof(uint64_t)); });
            }
            std::memcpy(dst + kept, src, std::min<uint64_t>(valid, each) * sizeof(uint64_t));
            for (auto &h : helpers) {
               h.join();
            }
         }
         else {
            std::memcpy(dst + kept, src, valid * sizeof(uint64_t));
         }
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
   static const bool tracing = mm_trace("split");           // (with the stages of scan_split: where a forward-engine stage's time goes)
   const auto t_start = std::chrono::steady_clock::now();
   auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() * 1e6; };
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
   const double t_enqueued = tracing ? since() : 0;
   HIP_TRY(hipStreamSynchronize(st));
   c->scans_recorded++;
   if (tracing) {
      fprintf(stderr, "      forward engine: %llu domains of %u tiles (batches of %u) enqueued in %.1f us, engine + packed lists + counters back after %.1f us\n",
              (unsigned long long)dg.ndom, dg.tpd, dg.batch, t_enqueued, since());
   }

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
      const int rc = total ? sort_on_device(c, c->d_sort_in, total) : MMH_OK;
      if (tracing) {
         fprintf(stderr, "      forward engine: %llu offsets, their sort enqueued after %.1f us\n", (unsigned long long)total, since());
      }
      return rc;
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
   std::vector<uint32_t> bits(words);
   std::vector<uint64_t> slots(first_pass_slots);
   if (w.mid_listed != ~0ull && w.mid_listed <= mm::mid_cap()) {
      // Round 6: the first pass itself listed every candidate its windows left open (d_mid_off: the hand-over to the second
      // phase, complete while it did not overflow) and left every candidate's verdict in its slot: the domains concerned
      // are the domains of that list -- a superset of what the second phase could not settle either --, and the flag pass
      // below (the filter over the whole ROM once more: 0.7 ms per 4 GiB) is not needed.
      std::vector<uint64_t> open(w.mid_listed);
      if (!open.empty()) {
         HIP_TRY(hipMemcpyAsync(open.data(), w.d_mid_off, open.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      }
      HIP_TRY(hipMemcpyAsync(slots.data(), w.d_out, first_pass_slots * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (uint64_t o : open) {
         const uint64_t blk = o / g.block_bytes;
         const uint64_t d = blk * g.S + (o - blk * g.block_bytes) % g.S;
         if (d < ndom) {
            bits[d >> 5] |= 1u << (d & 31);
         }
      }
      // (the second phase has written verdicts into some of the listed candidates' slots: they lie in flagged domains
      // and are dropped below like every other verdict there)
   }
   else {
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
      if (mm_trace("floods")) {
         fprintf(stderr, "run_flagged_domains: forward engine on %zu domains: %zu results\n", doms.size(), dense.size());
      }
      merged->insert(merged->end(), dense.begin(), dense.end());
   }
   static const bool trace = mm_trace("floods");
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
