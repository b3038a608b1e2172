// SPDX-License-Identifier: GPL-3.0-or-later
// Neutral types between the two sides of tests/cpp/facade_diff_main.cpp: the SAME source,
// facade_diff_side.cpp, is compiled twice -- against the reference's headers and core (class and
// namespace renamed on the command line: -DMonkeyMoore=MonkeyMooreRef -Dmmoore=mmoore_ref, so that
// both cores fit one binary) and against this repository's include/mmoore + libmonkey-core.so.
#ifndef FACADE_DIFF_HPP
#define FACADE_DIFF_HPP
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

struct DiffCase {
   int elem_bytes = 1;
   bool use_values = false;
   std::vector<char32_t> keyword;
   char32_t wildcard = 0;
   std::vector<char32_t> char_seq;
   std::vector<short> values;
   // engine runs
   std::string path;
   bool big_endian = false;
   int block_size = 524288;
   int threads = 1;
   int preview_width = 50;
   bool previews = false;
};

struct DiffMatch {
   uint64_t where = 0;                                      // element index (search) / byte offset (engine)
   std::vector<std::pair<uint32_t, uint32_t>> map;          // equivalency map, in key order
   std::string preview;
   bool operator==(const DiffMatch &o) const { return where == o.where && map == o.map && preview == o.preview; }
};

struct DiffOutcome {
   bool threw = false;
   std::string what;
   std::vector<DiffMatch> matches;
   int callbacks = 0;
};

DiffOutcome diff_search_ref(const DiffCase &c, const void *data, uint64_t count);
DiffOutcome diff_search_gpu(const DiffCase &c, const void *data, uint64_t count);
DiffOutcome diff_engine_ref(const DiffCase &c);
DiffOutcome diff_engine_gpu(const DiffCase &c);
#endif
