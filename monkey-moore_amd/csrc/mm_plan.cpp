// SPDX-License-Identifier: GPL-3.0-or-later
// mm_plan.cpp -- host-side pattern-plan builder (no device code).
//
// Turns a keyword (or value-scan vector) into the flattened mmh_plan_desc the
// HIP kernels consume.  It reproduces the tables of the reference's
// preprocessing (src/core/monkey_moore.cpp:54-304) but emits ONE uniform
// "compare list" for both scan loops instead of two table families:
//
//   position i (visited L-1 .. 0):  d = x[h+i] - x[h+i+bridge[i]]
//                                   mismatch iff ((d ^ expected[i]) & cmp_mask[i]) != 0
//
//   simple / value-scan (monkey_moore.cpp:106-142, 316-410)
//        bridge[i] = -1 (i >= 1), bridge[0] = L-1 (the wrap compare, :367-371)
//        cmp_mask  = 0xFFFFFFFF  -> signed, un-wrapped equality
//        wst[i]    = 255         -> no wildcard cap on the jump
//   wildcard / mixed case (monkey_moore.cpp:144-304, 425-546)
//        bridge[i] = distance to the previous literal (first wraps to last)
//        cmp_mask  = element mask on literals (modular equality), 0 on wildcards
//        wst[i]    = wildcard_skip_table
//
// The bad-character table is kept sparse (<= L distinct diffs + default).

#include "mmoore_hip.h"
#include "mm_internal.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

bool ascii_upper(uint32_t c) { return c >= 'A' && c <= 'Z'; }
bool ascii_lower(uint32_t c) { return c >= 'a' && c <= 'z'; }

struct Builder {
   uint32_t elem_bytes;
   int L;
   int max_val;
   std::vector<uint32_t> keyword;
   std::map<uint32_t, int> seq_index;   // like the reference's std::map: missing keys read as 0
   bool has_seq = false;

   int value_of(uint32_t c) const
   {
      if (!has_seq) {
         return 0;
      }
      auto it = seq_index.find(c);
      return it == seq_index.end() ? 0 : it->second;
   }

   // difference of two pattern symbols, monkey_moore.cpp:551-585 and :234-241
   int rel(uint32_t a, uint32_t b) const
   {
      if (has_seq) {
         return value_of(a) - value_of(b);
      }
      return static_cast<int>(static_cast<uint32_t>(a - b));   // CharType arithmetic is modulo 2^32
   }

   bool index_ok(int diff) const
   {
      long idx = static_cast<long>(diff) + max_val;
      return idx >= 0 && idx < 2L * (max_val + 1);              // :130, :261
   }
};

void put_skip(std::vector<std::pair<int, int>> &table, int diff, int value, bool only_if_unset)
{
   for (auto &e : table) {
      if (e.first == diff) {
         if (!only_if_unset) {
            e.second = value;
         }
         return;
      }
   }
   table.emplace_back(diff, value);
}

int finish(const Builder &b, mmh_plan_desc *out, const std::vector<std::pair<int, int>> &skips)
{
   if (out->match_jump < 1) {
      // the reference would never advance (monkey_moore.cpp:398, :526-527)
      mmh_set_error("keyword rejected: a match would advance the search head by %d", (int)out->match_jump);
      return MMH_E_PLAN;
   }
   out->n_skip = static_cast<uint32_t>(skips.size());
   for (size_t k = 0; k < skips.size(); k++) {
      out->skip_diff[k] = skips[k].first;
      out->skip_val[k] = skips[k].second;
   }
   (void)b;
   return MMH_OK;
}

int build_simple(const Builder &b, uint32_t mode, mmh_plan_desc *out)
{
   const int L = b.L;
   out->mode = mode;
   out->match_jump = static_cast<uint32_t>(L - 1);
   out->lead_wildcards = 0;
   out->first_literal = 0;
   out->default_skip = L - 1;

   std::vector<int> e(L);
   e[0] = b.rel(b.keyword[0], b.keyword[L - 1]);
   for (int i = 1; i < L; i++) {
      e[i] = b.rel(b.keyword[i], b.keyword[i - 1]);
   }
   std::vector<std::pair<int, int>> skips;
   for (int i = L - 1; i >= 0; i--) {            // rightmost occurrence wins, :127-141
      if (!b.index_ok(e[i])) {
         mmh_set_error("Skip table index out of bounds");
         return MMH_E_PLAN;
      }
      // the reference tests "still holds the default"; a stored value equal to the
      // default (only i == 0 produces it) is indistinguishable from unset
      bool unset = true;
      for (auto &s : skips) {
         if (s.first == e[i] && s.second != L - 1) {
            unset = false;
         }
      }
      if (unset) {
         put_skip(skips, e[i], L - 1 - i, false);
      }
   }
   for (int i = 0; i < L; i++) {
      out->expected[i] = e[i];
      out->cmp_mask[i] = 0xFFFFFFFFu;
      out->bridge[i] = static_cast<int8_t>(i == 0 ? L - 1 : -1);
      out->wst[i] = 255;
   }
   return finish(b, out, skips);
}

int build_wildcard(const Builder &b, uint32_t wildcard, mmh_plan_desc *out)
{
   const int L = b.L;
   std::vector<uint32_t> norm = b.keyword;

   if (!b.has_seq) {                              // :150-181
      int upper = 0, lower = 0;
      for (uint32_t c : b.keyword) {
         upper += ascii_upper(c);
         lower += ascii_lower(c);
      }
      if (upper > 0 && lower > 0) {
         for (auto &c : norm) {
            bool minority = upper > lower ? ascii_lower(c) : ascii_upper(c);
            if (minority) {
               c = wildcard;
            }
         }
      }
   }

   std::vector<int> literals;
   for (int i = 0; i < L; i++) {
      if (norm[i] != wildcard) {
         literals.push_back(i);
      }
   }
   int lead = 0;
   while (lead < L && norm[lead] == wildcard) {
      lead++;
   }

   out->mode = MMH_MODE_WILDCARD;
   out->lead_wildcards = static_cast<uint32_t>(lead);
   out->first_literal = static_cast<uint32_t>(literals.empty() ? 0 : literals.front());
   out->match_jump = static_cast<uint32_t>(std::max(L - 1 - lead, 0));
   out->default_skip = static_cast<int>(static_cast<signed char>(L - 1));   // :250-253

   std::vector<int> e(L, 0);
   for (int i = 0; i < L; i++) {
      out->expected[i] = 0;
      out->cmp_mask[i] = 0;
      out->bridge[i] = 0;
   }
   for (size_t k = 0; k < literals.size(); k++) {  // :222-247
      int cur = literals[k];
      int prev = k == 0 ? literals.back() : literals[k - 1];
      e[cur] = b.rel(norm[cur], norm[prev]);
      out->expected[cur] = e[cur];
      out->cmp_mask[cur] = static_cast<uint32_t>(b.max_val);
      out->bridge[cur] = static_cast<int8_t>(prev - cur);
   }

   std::vector<std::pair<int, int>> skips;
   for (int i = L - 1; i > 0; --i) {              // unconditional overwrite, index 0 excluded, :257-276
      if (!b.index_ok(e[i])) {
         mmh_set_error("Skip table index out of bounds");
         return MMH_E_PLAN;
      }
      int remaining = 0;
      for (int j = i + 1; j < L; j++) {
         remaining += norm[j] == wildcard;
      }
      put_skip(skips, e[i], static_cast<int>(static_cast<signed char>(L - remaining - i - 1)), false);
   }

   for (int i = 0; i < L; i++) {                  // :280-303
      if (norm[i] == wildcard) {
         out->wst[i] = 1;
      }
      else {
         int last = 0;
         for (int j = 0; j < i; j++) {
            if (norm[j] == wildcard) {
               last = j;
            }
         }
         out->wst[i] = static_cast<uint8_t>(std::max(i - last - 1, 1));
      }
   }
   return finish(b, out, skips);
}

} // namespace

extern "C" int mmh_plan_relative(uint32_t elem_bytes, const uint32_t *keyword, uint32_t keyword_len,
                                 uint32_t wildcard, const uint32_t *char_seq, uint32_t char_seq_len,
                                 mmh_plan_desc *out)
{
   if (!out || !keyword || (elem_bytes != 1 && elem_bytes != 2) || keyword_len == 0) {
      mmh_set_error("mmh_plan_relative: bad argument");
      return MMH_E_ARG;
   }
   if (keyword_len < 2) {
      mmh_set_error("keyword rejected: a 1-symbol keyword never advances the reference search head");
      return MMH_E_PLAN;
   }
   if (keyword_len > MMH_MAX_KEYWORD) {
      mmh_set_error("keyword longer than %d symbols is not supported by the GPU plan", MMH_MAX_KEYWORD);
      return MMH_E_PLAN;
   }
   std::memset(out, 0, sizeof(*out));
   Builder b;
   b.elem_bytes = elem_bytes;
   b.L = static_cast<int>(keyword_len);
   b.max_val = elem_bytes == 1 ? 0xFF : 0xFFFF;
   b.keyword.assign(keyword, keyword + keyword_len);
   b.has_seq = char_seq_len > 0;
   for (uint32_t i = 0; i < char_seq_len; i++) {
      b.seq_index[char_seq[i]] = static_cast<int>(i);           // later duplicates overwrite, :86-89
   }
   out->elem_bytes = elem_bytes;
   out->L = keyword_len;

   bool has_wildcards = std::count(b.keyword.begin(), b.keyword.end(), wildcard) > 0;
   bool case_change = false;
   if (!b.has_seq) {                                            // :68-73
      int upper = 0, lower = 0;
      for (uint32_t c : b.keyword) {
         upper += ascii_upper(c);
         lower += ascii_lower(c);
      }
      case_change = upper > 0 && lower > 0;
   }
   if (has_wildcards || case_change) {                          // :75-77
      return build_wildcard(b, wildcard, out);
   }
   return build_simple(b, MMH_MODE_SIMPLE, out);
}

extern "C" int mmh_plan_value_scan(uint32_t elem_bytes, const int16_t *values, uint32_t n, mmh_plan_desc *out)
{
   if (!out || !values || (elem_bytes != 1 && elem_bytes != 2) || n == 0) {
      mmh_set_error("mmh_plan_value_scan: bad argument");
      return MMH_E_ARG;
   }
   if (n < 2) {
      mmh_set_error("value scan rejected: a 1-value pattern never advances the reference search head");
      return MMH_E_PLAN;
   }
   if (n > MMH_MAX_KEYWORD) {
      mmh_set_error("value scan longer than %d values is not supported by the GPU plan", MMH_MAX_KEYWORD);
      return MMH_E_PLAN;
   }
   std::memset(out, 0, sizeof(*out));
   Builder b;
   b.elem_bytes = elem_bytes;
   b.L = static_cast<int>(n);
   b.max_val = elem_bytes == 1 ? 0xFF : 0xFFFF;
   for (uint32_t i = 0; i < n; i++) {
      b.keyword.push_back(static_cast<uint32_t>(static_cast<int32_t>(values[i])));   // :33-35
   }
   out->elem_bytes = elem_bytes;
   out->L = n;
   return build_simple(b, MMH_MODE_VALUE_SCAN, out);
}
