// SPDX-License-Identifier: GPL-3.0-or-later
// One side of the facade differential test (see facade_diff.hpp): written against the public mmoore
// API both trees share, compiled once per tree.  DIFF_SIDE is `ref` or `gpu`.
#include <atomic>
#include <exception>

#include "mmoore/monkey_moore.hpp"
#include "mmoore/search_engine.hpp"

#include "facade_diff.hpp"

#define DIFF_CAT2(a, b) a##b
#define DIFF_CAT(a, b) DIFF_CAT2(a, b)

namespace {
template <class Ty> DiffOutcome search_one(const DiffCase &c, const void *data, uint64_t count)
{
   DiffOutcome out;
   try {
      MonkeyMoore<Ty> m = c.use_values ? MonkeyMoore<Ty>(c.values) : MonkeyMoore<Ty>(c.keyword, c.wildcard, c.char_seq);
      for (const auto &r : m.search(static_cast<const Ty *>(data), count)) {
         DiffMatch d;
         d.where = r.first;
         for (const auto &kv : r.second) {
            d.map.emplace_back((uint32_t)kv.first, (uint32_t)kv.second);
         }
         out.matches.push_back(std::move(d));
      }
   }
   catch (const std::exception &e) {
      out.threw = true;
      out.what = e.what();
   }
   return out;
}

template <class Ty> DiffOutcome engine_one(const DiffCase &c)
{
   DiffOutcome out;
   try {
      mmoore::SearchConfig cfg;
      cfg.file_path = c.path;
      cfg.is_relative_search = !c.use_values;
      cfg.endianness = c.big_endian ? mmoore::Endianness::Big : mmoore::Endianness::Little;
      cfg.keyword = c.keyword;
      cfg.custom_char_seq = c.char_seq;
      cfg.wildcard = c.wildcard;
      cfg.reference_values = c.values;
      cfg.preferred_num_threads = c.threads;
      cfg.preferred_search_block_size = c.block_size;
      cfg.preferred_preview_width = c.preview_width;
      mmoore::SearchEngine<Ty> eng(cfg);
      std::atomic<bool> abort_flag{false};
      std::atomic<int> calls{0};
      auto results = eng.run([&](int, const mmoore::SearchStep) { calls++; }, abort_flag, c.previews);
      out.callbacks = calls.load();
      for (const auto &r : results) {
         DiffMatch d;
         d.where = r.offset;
         for (const auto &kv : r.values_map) {
            d.map.emplace_back((uint32_t)kv.first, (uint32_t)kv.second);
         }
         d.preview = r.preview;
         out.matches.push_back(std::move(d));
      }
   }
   catch (const std::exception &e) {
      out.threw = true;
      out.what = e.what();
   }
   return out;
}
} // namespace

DiffOutcome DIFF_CAT(diff_search_, DIFF_SIDE)(const DiffCase &c, const void *data, uint64_t count)
{
   return c.elem_bytes == 1 ? search_one<uint8_t>(c, data, count) : search_one<uint16_t>(c, data, count);
}

DiffOutcome DIFF_CAT(diff_engine_, DIFF_SIDE)(const DiffCase &c)
{
   return c.elem_bytes == 1 ? engine_one<uint8_t>(c) : engine_one<uint16_t>(c);
}
