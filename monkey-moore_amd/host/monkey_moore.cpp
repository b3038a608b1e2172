// SPDX-License-Identifier: GPL-3.0-or-later
// monkey_moore.cpp -- MonkeyMoore<Ty> on the MI355X engine (C++17 facade over the C ABI).
//
// Mirrors the behaviour of the reference's src/core/monkey_moore.cpp behind the same
// class; the matching itself happens in the HIP kernels.
#include "mmoore/monkey_moore.hpp"

#include <cstdlib>
#include <stdexcept>

#include "matcher_state.hpp"
#include "mmoore/text_utils.hpp"

namespace mmoore_amd {

void throw_last_error(const char *what)
{
   throw std::runtime_error(std::string(what) + ": " + mmh_last_error());
}

namespace {
struct ContextHolder {
   mmh_ctx *ctx = nullptr;
   ~ContextHolder()
   {
      if (ctx) {
         mmh_destroy(ctx);
      }
   }
};
} // namespace

mmh_ctx *thread_context()
{
   thread_local ContextHolder holder;
   if (!holder.ctx) {
      const char *env = std::getenv("MMOORE_HIP_DEVICE");
      const int device = env ? std::atoi(env) : 0;
      if (mmh_create(device, &holder.ctx) != MMH_OK) {
         throw_last_error("MI355X engine unavailable (there is no CPU fallback)");
      }
      (void)mmh_set_timing(holder.ctx, 0);     // (nobody behind this API asks for device timings: ~4.5 us less per search)
   }
   return holder.ctx;
}

static void plan_or_throw(int rc)
{
   if (rc != MMH_OK) {
      // same exception type as the reference's table builder (monkey_moore.cpp:139, :274)
      throw std::runtime_error(mmh_last_error());
   }
}

static std::shared_ptr<const MatcherState> make_relative(uint32_t elem_bytes, const std::vector<CharType> &keyword,
                                                         CharType wildcard, const std::vector<CharType> &char_seq)
{
   assert(!keyword.empty());
   auto st = std::make_shared<MatcherState>();
   st->keyword = keyword;
   st->char_seq = char_seq;
   for (size_t i = 0; i < char_seq.size(); i++) {
      st->seq_index[char_seq[i]] = static_cast<int>(i);
   }
   std::vector<uint32_t> kw(keyword.begin(), keyword.end()), seq(char_seq.begin(), char_seq.end());
   plan_or_throw(mmh_plan_relative(elem_bytes, kw.data(), static_cast<uint32_t>(kw.size()), wildcard,
                                   seq.empty() ? nullptr : seq.data(), static_cast<uint32_t>(seq.size()), &st->plan));

   if (char_seq.empty()) {
      const auto uppers = std::count_if(keyword.begin(), keyword.end(), is_ascii_upper);
      const auto lowers = std::count_if(keyword.begin(), keyword.end(), is_ascii_lower);
      st->has_case_change = uppers > 0 && lowers > 0;
      st->mostly_lowercase = lowers > uppers;
      if (st->has_case_change) {
         auto minority = [&](CharType c) { return st->mostly_lowercase ? is_ascii_upper(c) : is_ascii_lower(c); };
         st->opposing = static_cast<int>(std::find_if(keyword.begin(), keyword.end(), minority) - keyword.begin());
      }
   }
   // simple path: the first symbol; wildcard path: the first literal of the normalised keyword
   st->anchor = st->plan.mode == MMH_MODE_WILDCARD ? static_cast<int>(st->plan.first_literal) : 0;
   return st;
}

static std::shared_ptr<const MatcherState> make_value_scan(uint32_t elem_bytes, const std::vector<short> &values)
{
   assert(!values.empty());
   auto st = std::make_shared<MatcherState>();
   st->value_scan = true;
   for (short v : values) {
      st->keyword.push_back(static_cast<CharType>(v));
   }
   plan_or_throw(mmh_plan_value_scan(elem_bytes, values.data(), static_cast<uint32_t>(values.size()), &st->plan));
   return st;
}

} // namespace mmoore_amd

template <class Ty>
MonkeyMoore<Ty>::MonkeyMoore(const std::vector<CharType> &keyword, CharType wildcard, const std::vector<CharType> &char_seq)
   : st(mmoore_amd::make_relative(sizeof(Ty), keyword, wildcard, char_seq))
{
}

template <class Ty>
MonkeyMoore<Ty>::MonkeyMoore(const std::vector<short> &reference_values)
   : st(mmoore_amd::make_value_scan(sizeof(Ty), reference_values))
{
}

template <class Ty>
std::vector<typename MonkeyMoore<Ty>::result_type> MonkeyMoore<Ty>::search(const Ty *data, uint64_t data_len)
{
   using namespace mmoore_amd;
   std::vector<result_type> results;
   if (data_len == 0) {
      return results;
   }
   mmh_ctx *ctx = thread_context();
   if (mmh_rom_upload(ctx, data, data_len * sizeof(Ty)) != MMH_OK) {
      throw_last_error("uploading the search buffer failed");
   }
   std::vector<uint64_t> positions(4096);
   uint64_t count = 0;
   for (;;) {
      // block_bytes 0: one chain over the whole buffer, results are element indices
      int rc = mmh_scan(ctx, &st->plan, 0, 0, 0, positions.data(), positions.size(), &count);
      if (rc == MMH_E_CAPACITY) {
         positions.resize(count + 16);
         continue;
      }
      if (rc != MMH_OK) {
         throw_last_error("GPU scan failed");
      }
      break;
   }
   results.reserve(count);
   for (uint64_t i = 0; i < count; i++) {
      const Ty *at = data + positions[i];
      results.emplace_back(positions[i], build_values_map<Ty>(*st, [at](int k) { return at[k]; }));
   }
   return results;
}

template class MonkeyMoore<uint8_t>;
template class MonkeyMoore<uint16_t>;
