// SPDX-License-Identifier: GPL-3.0-or-later
// text_utils.hpp -- small helpers of the mmoore API (MI355X build).
//
// The names and meanings match what the reference's header of the same name offers
// (find_last_index, count_prefix_length, is_ascii_upper / lower / digit) so that harness
// code written against it keeps compiling.  The pattern-plan builder of this build
// (monkey-moore_amd/csrc/mm_plan.cpp) has its own copies and does not depend on this file.
#ifndef MMOORE_AMD_TEXT_UTILS_HPP
#define MMOORE_AMD_TEXT_UTILS_HPP

#include <algorithm>
#include <cstdint>
#include <iterator>

// Position (counted from `first`) of the LAST element of [first, last) equal to `value`;
// -1 when no element is.
template <class FwdIt, class T>
inline int find_last_index(FwdIt first, const FwdIt last, const T &value)
{
   int answer = -1;
   for (int position = 0; first != last; ++first, ++position) {
      answer = (*first == value) ? position : answer;
   }
   return answer;
}

// How many elements at the FRONT of [first, last) equal `value` before the first one that
// does not.
template <class FwdIt, class T>
inline int count_prefix_length(FwdIt first, const FwdIt last, const T &value)
{
   const FwdIt stop = std::find_if(first, last, [&value](const auto &element) { return !(element == value); });
   return static_cast<int>(std::distance(first, stop));
}

// ASCII classification in the "C" locale.  Code points outside 7-bit ASCII are never
// letters or digits, whatever the process locale says.
inline bool is_ascii_upper(const char32_t &c) { return c >= U'A' && c <= U'Z'; }
inline bool is_ascii_lower(const char32_t &c) { return c >= U'a' && c <= U'z'; }
inline bool is_ascii_digit(const char32_t &c) { return c >= U'0' && c <= U'9'; }

#endif
