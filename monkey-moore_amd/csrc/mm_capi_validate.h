// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_validate.h -- what a polled scan published, checked before it is trusted (include/mmoore_hip.h, "route health").
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

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
