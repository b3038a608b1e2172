// SPDX-License-Identifier: GPL-3.0-or-later
// monkey_moore.hpp -- MonkeyMoore<Ty>, the relative-search matcher of the mmoore API,
// backed by the MI355X engine (libmmoore_hip.so, include/mmoore_hip.h).
//
// Public surface = the reference's include/mmoore/monkey_moore.hpp:18-51 (same template,
// constructors, search() signature, result types), so its harnesses compile unchanged.
// Behind it there is no CPU matcher: the constructor flattens the keyword into an
// mmh_plan_desc, search() uploads the borrowed buffer to HBM, runs one whole-buffer scan on
// the GPU and rebuilds each match's equivalency map on the host from one or two data
// elements (reference src/core/monkey_moore.cpp:374-393, 472-521).
//
// Errors: an empty keyword asserts; keywords the reference rejects ("Skip table index out of
// bounds") or cannot terminate on, and any device failure, throw std::runtime_error.
// search() may be called concurrently on one instance (every thread uses its own device
// context), exactly as SearchEngine's workers do in the reference.
#ifndef MMOORE_AMD_MONKEY_MOORE_HPP
#define MMOORE_AMD_MONKEY_MOORE_HPP

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

using CharType = char32_t;

namespace mmoore_amd {
struct MatcherState;   // plan + what is needed to rebuild equivalency maps (host/monkey_moore.cpp)
}

template <class Ty> class MonkeyMoore {
public:
   using equivalency_map = std::map<CharType, Ty>;
   using result_type = std::pair<uint64_t, equivalency_map>;

   // relative search for `keyword`; `wildcard` matches any element; with a non-empty
   // `char_seq` symbols are valued by their index in it instead of their code point
   MonkeyMoore(const std::vector<CharType> &keyword, CharType wildcard = 0,
               const std::vector<CharType> &char_seq = {});

   // value-scan: the relative pattern of a list of numbers
   MonkeyMoore(const std::vector<short> &reference_values);

   // element indices (ascending) of the matches in data[0 .. data_len), each with its
   // equivalency map; `data` stays owned by the caller
   std::vector<result_type> search(const Ty *data, uint64_t data_len);

   // MI355X build only: the flattened plan, for callers that drive the C ABI themselves
   const mmoore_amd::MatcherState &state() const { return *st; }

private:
   std::shared_ptr<const mmoore_amd::MatcherState> st;
};

#endif
