// SPDX-License-Identifier: GPL-3.0-or-later
// ref_shim.cpp -- extern "C" entry points over the UNMODIFIED reference core.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
// /root/reference/src/core/{monkey_moore,search_engine}.cpp (from where they lie,
// never copied) by oracle/Makefile into oracle/_ref/libmmref.so.  It exists so
// that Python tests, the golden-vector generator and bench.py's cpu_baseline leg
// can drive the real MonkeyMoore<T>::search / SearchEngine<T>::run.
//
// Interfaces bound: include/mmoore/monkey_moore.hpp:24-47,
// include/mmoore/search_engine.hpp:23-59.

#include "mmoore/monkey_moore.hpp"
#include "mmoore/search_engine.hpp"

#include <atomic>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;

struct Payload {
   std::vector<std::vector<std::pair<uint32_t, uint32_t>>> maps;
   std::vector<std::string> previews;
   int progress_calls = 0;
   int last_progress = -1;
   bool progress_monotone = true;
};
thread_local Payload g_payload;

template <class Ty>
int64_t do_search(MonkeyMoore<Ty> &searcher, const void *data, uint64_t len,
                  uint64_t *out, uint64_t cap)
{
   auto results = searcher.search(static_cast<const Ty *>(data), len);
   g_payload.maps.clear();
   uint64_t n = 0;
   for (auto &r : results) {
      if (n < cap && out) {
         out[n] = r.first;
      }
      std::vector<std::pair<uint32_t, uint32_t>> m;
      for (auto &kv : r.second) {
         m.emplace_back(static_cast<uint32_t>(kv.first), static_cast<uint32_t>(kv.second));
      }
      g_payload.maps.push_back(std::move(m));
      n++;
   }
   return static_cast<int64_t>(n);
}

template <class Ty>
int64_t do_engine(const mmoore::SearchConfig &cfg, int gen_previews, int abort_after,
                  uint64_t *out, uint64_t cap)
{
   mmoore::SearchEngine<Ty> engine(cfg);
   std::atomic<bool> abort{false};
   g_payload = Payload{};
   auto cb = [&](int pct, const mmoore::SearchStep) {
      g_payload.progress_calls++;
      if (pct < g_payload.last_progress) {
         g_payload.progress_monotone = false;
      }
      g_payload.last_progress = pct;
      if (abort_after > 0 && g_payload.progress_calls >= abort_after) {
         abort = true;
      }
   };
   auto results = engine.run(cb, abort, gen_previews != 0);
   uint64_t n = 0;
   for (auto &r : results) {
      if (n < cap && out) {
         out[n] = r.offset;
      }
      std::vector<std::pair<uint32_t, uint32_t>> m;
      for (auto &kv : r.values_map) {
         m.emplace_back(static_cast<uint32_t>(kv.first), static_cast<uint32_t>(kv.second));
      }
      g_payload.maps.push_back(std::move(m));
      g_payload.previews.push_back(r.preview);
      n++;
   }
   return static_cast<int64_t>(n);
}

std::vector<CharType> to_vec(const uint32_t *p, int n)
{
   std::vector<CharType> v;
   v.reserve(n > 0 ? n : 0);
   for (int i = 0; i < n; i++) {
      v.push_back(static_cast<CharType>(p[i]));
   }
   return v;
}

} // namespace

extern "C" {

const char *mmref_last_error() { return g_error.c_str(); }

// MonkeyMoore<Ty>(keyword, wildcard, char_seq).search(data, len)
int64_t mmref_search(int elem_bytes, const uint32_t *kw, int kw_len, uint32_t wildcard,
                     const uint32_t *seq, int seq_len, const void *data, uint64_t len,
                     uint64_t *out, uint64_t cap)
{
   try {
      if (elem_bytes == 1) {
         MonkeyMoore<uint8_t> s(to_vec(kw, kw_len), wildcard, to_vec(seq, seq_len));
         return do_search(s, data, len, out, cap);
      }
      MonkeyMoore<uint16_t> s(to_vec(kw, kw_len), wildcard, to_vec(seq, seq_len));
      return do_search(s, data, len, out, cap);
   }
   catch (const std::exception &e) {
      g_error = e.what();
      return -1;
   }
}

// MonkeyMoore<Ty>(reference_values).search(data, len)
int64_t mmref_value_scan(int elem_bytes, const int16_t *vals, int n, const void *data,
                         uint64_t len, uint64_t *out, uint64_t cap)
{
   try {
      std::vector<short> v(vals, vals + n);
      if (elem_bytes == 1) {
         MonkeyMoore<uint8_t> s(v);
         return do_search(s, data, len, out, cap);
      }
      MonkeyMoore<uint16_t> s(v);
      return do_search(s, data, len, out, cap);
   }
   catch (const std::exception &e) {
      g_error = e.what();
      return -1;
   }
}

// SearchEngine<T>(cfg).run(cb, abort, previews)
int64_t mmref_engine(int elem_bytes, const char *path, int is_relative,
                     const uint32_t *kw, int kw_len, uint32_t wildcard,
                     const uint32_t *seq, int seq_len,
                     const int16_t *ref_vals, int n_ref_vals,
                     int big_endian, int threads, int block_size, int preview_width,
                     int gen_previews, int abort_after,
                     uint64_t *out, uint64_t cap)
{
   try {
      mmoore::SearchConfig cfg;
      cfg.file_path = path;
      cfg.is_relative_search = is_relative != 0;
      cfg.endianness = big_endian ? mmoore::Endianness::Big : mmoore::Endianness::Little;
      cfg.keyword = to_vec(kw, kw_len);
      cfg.custom_char_seq = to_vec(seq, seq_len);
      cfg.wildcard = wildcard;
      cfg.reference_values.assign(ref_vals, ref_vals + n_ref_vals);
      cfg.preferred_num_threads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
      cfg.preferred_search_block_size = block_size;
      cfg.preferred_preview_width = preview_width;
      return elem_bytes == 1 ? do_engine<uint8_t>(cfg, gen_previews, abort_after, out, cap)
                             : do_engine<uint16_t>(cfg, gen_previews, abort_after, out, cap);
   }
   catch (const std::exception &e) {
      g_error = e.what();
      return -1;
   }
}

// payload of the most recent call on this thread
int mmref_result_map(int64_t i, uint32_t *keys, uint32_t *vals, int cap)
{
   if (i < 0 || (size_t)i >= g_payload.maps.size()) {
      return -1;
   }
   int n = 0;
   for (auto &kv : g_payload.maps[(size_t)i]) {
      if (n < cap) {
         keys[n] = kv.first;
         vals[n] = kv.second;
      }
      n++;
   }
   return n;
}

int mmref_result_preview(int64_t i, char *buf, int cap)
{
   if (i < 0 || (size_t)i >= g_payload.previews.size()) {
      return -1;
   }
   const std::string &s = g_payload.previews[(size_t)i];
   if (cap > 0) {
      size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
      std::memcpy(buf, s.data(), n);
      buf[n] = 0;
   }
   return (int)s.size();
}

void mmref_progress_info(int *calls, int *last, int *monotone)
{
   *calls = g_payload.progress_calls;
   *last = g_payload.last_progress;
   *monotone = g_payload.progress_monotone ? 1 : 0;
}

int mmref_hardware_concurrency() { return (int)std::thread::hardware_concurrency(); }

// The input of the reference's own benchmark (benchmarks/bench_search.cpp:11-22 describes it:
// std::mt19937 seeded with 42, one uniform_int_distribution<unsigned>(0, max(DataType)) draw
// per element).  Re-stated here so that tests can run MonkeyMoore<T>::search of the compiled
// reference and the GPU engine on exactly that buffer.
void mmref_bench_data(int elem_bytes, uint64_t nbytes, void *out)
{
   std::mt19937 rng(42);
   if (elem_bytes == 1) {
      std::uniform_int_distribution<unsigned int> dist(0, std::numeric_limits<uint8_t>::max());
      uint8_t *p = static_cast<uint8_t *>(out);
      for (uint64_t i = 0; i < nbytes; i++) {
         p[i] = static_cast<uint8_t>(dist(rng));
      }
   }
   else {
      std::uniform_int_distribution<unsigned int> dist(0, std::numeric_limits<uint16_t>::max());
      uint16_t *p = static_cast<uint16_t *>(out);
      for (uint64_t i = 0; i < nbytes / 2; i++) {
         p[i] = static_cast<uint16_t>(dist(rng));
      }
   }
}

} // extern "C"
