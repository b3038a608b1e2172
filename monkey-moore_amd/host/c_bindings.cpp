// SPDX-License-Identifier: GPL-3.0-or-later
// c_bindings.cpp -- the include/mmoore C++ API (MonkeyMoore<T>::search, SearchEngine<T>::run) as plain C entry points of
// libmonkey-core.so, for hosts that bind through an FFI instead of linking C++ (ctypes, cgo, JNI ...).
// bench.py's `end_to_end` leg drives the facade through these: the very call the reference's CPU baseline is timed on
// (SearchEngine<uint8_t>::run on a file, src/core/search_engine.cpp:23-216), results and equivalency maps compared.
//
// The results of the most recent call stay with the calling thread until its next call: offsets are copied out by the
// call itself, the equivalency map / preview of result i are fetched afterwards.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "mmoore/monkey_moore.hpp"
#include "mmoore/search_engine.hpp"

namespace {

struct LastCall {
   std::string error;
   std::vector<std::vector<std::pair<uint32_t, uint32_t>>> maps;
   std::vector<std::string> previews;
   int progress_calls = 0;
};
thread_local LastCall g_last;

std::vector<CharType> code_points(const uint32_t *p, int n)
{
   return std::vector<CharType>(p, p + (n > 0 ? n : 0));
}

template <class Map>
void keep_map(const Map &m)
{
   std::vector<std::pair<uint32_t, uint32_t>> flat;
   flat.reserve(m.size());
   for (const auto &kv : m) {
      flat.emplace_back(static_cast<uint32_t>(kv.first), static_cast<uint32_t>(kv.second));
   }
   g_last.maps.push_back(std::move(flat));
}

template <class T>
int64_t search_buffer(MonkeyMoore<T> &matcher, const void *data, uint64_t len, uint64_t *out, uint64_t cap)
{
   const auto found = matcher.search(static_cast<const T *>(data), len);
   uint64_t n = 0;
   for (const auto &r : found) {
      if (out && n < cap) {
         out[n] = r.first;
      }
      keep_map(r.second);
      n++;
   }
   return static_cast<int64_t>(n);
}

template <class T>
int64_t run_file(const mmoore::SearchConfig &cfg, bool previews, uint64_t *out, uint64_t cap)
{
   mmoore::SearchEngine<T> engine(cfg);
   std::atomic<bool> abort{false};
   const auto found = engine.run([](int, const mmoore::SearchStep) { g_last.progress_calls++; }, abort, previews);
   uint64_t n = 0;
   for (const auto &r : found) {
      if (out && n < cap) {
         out[n] = r.offset;
      }
      keep_map(r.values_map);
      g_last.previews.push_back(r.preview);
      n++;
   }
   return static_cast<int64_t>(n);
}

} // namespace

extern "C" {

const char *mmoore_c_last_error(void) { return g_last.error.c_str(); }

// MonkeyMoore<Ty>(keyword, wildcard, char_seq).search(data, len) -- include/mmoore/monkey_moore.hpp.  len in ELEMENTS;
// element indices to out (at most cap); returns their number, -1 on error (mmoore_c_last_error).
int64_t mmoore_c_search(int elem_bytes, const uint32_t *keyword, int keyword_len, uint32_t wildcard, const uint32_t *char_seq,
                        int char_seq_len, const void *data, uint64_t len, uint64_t *out, uint64_t cap)
{
   g_last = LastCall();
   try {
      if (elem_bytes == 1) {
         MonkeyMoore<uint8_t> m(code_points(keyword, keyword_len), wildcard, code_points(char_seq, char_seq_len));
         return search_buffer(m, data, len, out, cap);
      }
      MonkeyMoore<uint16_t> m(code_points(keyword, keyword_len), wildcard, code_points(char_seq, char_seq_len));
      return search_buffer(m, data, len, out, cap);
   }
   catch (const std::exception &e) {
      g_last.error = e.what();
      return -1;
   }
}

// SearchEngine<T>(config).run(progress, abort, generate_previews) -- include/mmoore/search_engine.hpp.  keyword_len == 0:
// a value scan of `values`.  Byte offsets to out (at most cap); returns their number, -1 on error.
int64_t mmoore_c_engine_run(int elem_bytes, const char *path, const uint32_t *keyword, int keyword_len, uint32_t wildcard,
                            const uint32_t *char_seq, int char_seq_len, const int16_t *values, int n_values, int big_endian,
                            int block_size, int preview_width, int generate_previews, uint64_t *out, uint64_t cap)
{
   g_last = LastCall();
   try {
      mmoore::SearchConfig cfg;
      cfg.file_path = path;
      cfg.is_relative_search = keyword_len > 0;
      cfg.endianness = big_endian ? mmoore::Endianness::Big : mmoore::Endianness::Little;
      cfg.keyword = code_points(keyword, keyword_len);
      cfg.custom_char_seq = code_points(char_seq, char_seq_len);
      cfg.wildcard = wildcard;
      if (values && n_values > 0) {
         cfg.reference_values.assign(values, values + n_values);
      }
      cfg.preferred_search_block_size = block_size;
      cfg.preferred_preview_width = preview_width;
      return elem_bytes == 1 ? run_file<uint8_t>(cfg, generate_previews != 0, out, cap)
                             : run_file<uint16_t>(cfg, generate_previews != 0, out, cap);
   }
   catch (const std::exception &e) {
      g_last.error = e.what();
      return -1;
   }
}

// equivalency map of result i of the last call on this thread: pairs (symbol, element value); returns their number
int mmoore_c_result_map(int64_t i, uint32_t *symbols, uint32_t *values, int cap)
{
   if (i < 0 || static_cast<size_t>(i) >= g_last.maps.size()) {
      return -1;
   }
   int n = 0;
   for (const auto &kv : g_last.maps[static_cast<size_t>(i)]) {
      if (n < cap) {
         symbols[n] = kv.first;
         values[n] = kv.second;
      }
      n++;
   }
   return n;
}

// All equivalency maps of the last call at once, for maps of the same length (ASCII keywords: 'A' and 'a'): result i's
// pairs at [i * pairs_each, (i + 1) * pairs_each).  Returns the number of results copied, -1 when a map has another length.
int64_t mmoore_c_result_maps(uint32_t *symbols, uint32_t *values, uint64_t cap_results, int pairs_each)
{
   uint64_t i = 0;
   for (; i < g_last.maps.size() && i < cap_results; i++) {
      const auto &m = g_last.maps[i];
      if (static_cast<int>(m.size()) != pairs_each) {
         return -1;
      }
      for (int k = 0; k < pairs_each; k++) {
         symbols[i * pairs_each + k] = m[k].first;
         values[i * pairs_each + k] = m[k].second;
      }
   }
   return static_cast<int64_t>(i);
}

// preview text (UTF-8) of result i of the last mmoore_c_engine_run; returns its length in bytes
int mmoore_c_result_preview(int64_t i, char *buf, int cap)
{
   if (i < 0 || static_cast<size_t>(i) >= g_last.previews.size()) {
      return -1;
   }
   const std::string &s = g_last.previews[static_cast<size_t>(i)];
   if (buf && cap > 0) {
      const size_t n = std::min(s.size(), static_cast<size_t>(cap) - 1);
      std::memcpy(buf, s.data(), n);
      buf[n] = 0;
   }
   return static_cast<int>(s.size());
}

int mmoore_c_progress_calls(void) { return g_last.progress_calls; }

} // extern "C"
