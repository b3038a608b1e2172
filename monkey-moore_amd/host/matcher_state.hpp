// SPDX-License-Identifier: GPL-3.0-or-later
// matcher_state.hpp -- host-side state shared by the MonkeyMoore<Ty> / SearchEngine<T>
// facade (not installed; the public headers only forward-declare MatcherState).
#ifndef MMOORE_AMD_MATCHER_STATE_HPP
#define MMOORE_AMD_MATCHER_STATE_HPP

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "mmoore/monkey_moore.hpp"
#include "mmoore_hip.h"

namespace mmoore_amd {

struct MatcherState {
   mmh_plan_desc plan{};                 // what the GPU consumes
   std::vector<CharType> keyword;        // as given (value scan: the numbers as code points)
   std::vector<CharType> char_seq;
   std::map<CharType, int> seq_index;    // symbol -> index in char_seq (later duplicates win)
   bool value_scan = false;
   bool has_case_change = false;         // ASCII keyword mixing upper and lower case
   bool mostly_lowercase = false;
   int anchor = 0;                       // keyword position the alphabet offset is read from
   int opposing = -1;                    // first position of the minority case (mixed case only)
};

// throws std::runtime_error with the C ABI's message
[[noreturn]] void throw_last_error(const char *what);

// one device context per calling thread (search() is re-entrant across threads)
mmh_ctx *thread_context();

// equivalency map of one match; elem(k) returns element k of the match
template <class Ty, class ElemAt>
std::map<CharType, Ty> build_values_map(const MatcherState &st, ElemAt elem)
{
   std::map<CharType, Ty> out;
   if (st.value_scan) {
      return out;                                          // offsets only (reference monkey_moore.cpp:377)
   }
   auto index_of = [&](CharType c) {
      auto it = st.seq_index.find(c);
      return it == st.seq_index.end() ? 0 : it->second;
   };
   if (!st.char_seq.empty()) {
      // every symbol of the custom sequence, shifted by where the anchor symbol landed (:386-391, :515-520)
      const int shift = static_cast<int>(elem(st.anchor)) - index_of(st.keyword[st.anchor]);
      for (CharType c : st.char_seq) {
         out[c] = static_cast<Ty>(index_of(c) + shift);
      }
      return out;
   }
   const int shift = static_cast<int>(elem(st.anchor)) - static_cast<int>(st.keyword[st.anchor]);
   int upper_shift = shift, lower_shift = shift;
   if (st.has_case_change) {
      // the minority case has its own base, read from its first occurrence (:490-511)
      const int other = static_cast<int>(elem(st.opposing)) - static_cast<int>(st.keyword[st.opposing]);
      (st.mostly_lowercase ? upper_shift : lower_shift) = other;
   }
   out[U'A'] = static_cast<Ty>('A' + upper_shift);
   out[U'a'] = static_cast<Ty>('a' + lower_shift);
   return out;
}

} // namespace mmoore_amd

#endif
