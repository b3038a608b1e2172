// text_utils.hpp -- small helpers of the mmoore API (MI355X build).
//
// Same names and meaning as the reference's include/mmoore/text_utils.hpp:13-56 so that
// harness code written against it keeps compiling; the implementations are ours.
#ifndef MMOORE_AMD_TEXT_UTILS_HPP
#define MMOORE_AMD_TEXT_UTILS_HPP

#include <cstdint>
#include <iterator>

// index of the last element equal to `value` in [first, last), -1 when there is none
template <class FwdIt, class T>
inline int find_last_index(FwdIt first, const FwdIt last, const T &value)
{
   int found = -1;
   int index = 0;
   for (; first != last; ++first, ++index) {
      if (*first == value) {
         found = index;
      }
   }
   return found;
}

// length of the run of `value` at the front of [first, last)
template <class FwdIt, class T>
inline int count_prefix_length(FwdIt first, const FwdIt last, const T &value)
{
   int run = 0;
   while (first != last && *first == value) {
      ++first;
      ++run;
   }
   return run;
}

// ASCII classification in the "C" locale, code points >= 128 are never letters/digits
inline bool is_ascii_upper(const char32_t &c) { return c >= U'A' && c <= U'Z'; }
inline bool is_ascii_lower(const char32_t &c) { return c >= U'a' && c <= U'z'; }
inline bool is_ascii_digit(const char32_t &c) { return c >= U'0' && c <= U'9'; }

#endif
