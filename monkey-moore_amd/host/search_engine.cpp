// search_engine.cpp -- SearchEngine<T>::run on the MI355X engine.
//
// Replaces the reference's block dispatcher (src/core/search_engine.cpp:23-216): the file
// is streamed to HBM in block-aligned partitions, each partition is scanned on the GPU
// with the reference's block semantics, results are byte offsets in the file.  Host work
// that remains here: file I/O, the progress / abort protocol, the per-match equivalency
// map and the optional preview text (search_engine.cpp:256-348 restated).
#include "mmoore/search_engine.hpp"

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "matcher_state.hpp"

namespace {

using mmoore_amd::MatcherState;

// HBM upload granularity (rounded to whole blocks).  The parallel ingest needs a few hundred MiB to
// get up to speed (measured: 45 GB/s for 1 GiB pieces, 55.6 GB/s for 4 GiB); abort is polled per partition.
constexpr uint64_t kPartitionBytes = 4ull << 30;

std::string utf8_of(char32_t cp)
{
   std::string s;
   if (cp < 0x80) {
      s += static_cast<char>(cp);
   }
   else if (cp < 0x800) {
      s += static_cast<char>(0xC0 | (cp >> 6));
      s += static_cast<char>(0x80 | (cp & 0x3F));
   }
   else if (cp < 0x10000) {
      s += static_cast<char>(0xE0 | (cp >> 12));
      s += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      s += static_cast<char>(0x80 | (cp & 0x3F));
   }
   else {
      s += static_cast<char>(0xF0 | (cp >> 18));
      s += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
      s += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      s += static_cast<char>(0x80 | (cp & 0x3F));
   }
   return s;
}

template <typename T>
T element_at(const uint8_t *bytes, mmoore::Endianness order)
{
   if (sizeof(T) == 1) {
      return static_cast<T>(bytes[0]);
   }
   return order == mmoore::Endianness::Big ? static_cast<T>(bytes[0] << 8 | bytes[1])
                                           : static_cast<T>(bytes[1] << 8 | bytes[0]);
}

// text shown for one match: `width` elements around it, decoded through the match's own
// alphabet; unknown values print as '#'; value scans print hex (search_engine.cpp:256-348)
template <typename T>
std::string make_preview(const mmoore::SearchConfig &cfg, std::ifstream &file, uint64_t file_size, uint64_t match_offset,
                         const std::map<CharType, T> &values_map)
{
   const int width = cfg.preferred_preview_width;
   const int64_t elem = sizeof(T);
   // centre the keyword in the window
   const int64_t lead_positions = width / 2 - static_cast<int64_t>(cfg.keyword.size() / 2);
   int64_t lead_bytes = lead_positions * elem;
   lead_bytes = (lead_bytes + (elem - 1)) & ~(elem - 1);
   int64_t first = static_cast<int64_t>(match_offset) - lead_bytes;
   const int64_t last = first + static_cast<int64_t>(width) * elem;
   if (static_cast<uint64_t>(last) > file_size) {          // the reference compares as unsigned, too
      first -= last - static_cast<int64_t>(file_size);
   }
   file.clear();
   file.seekg(std::max<int64_t>(0, first), std::ios::beg);
   std::vector<uint8_t> raw(static_cast<size_t>(width) * elem);
   file.read(reinterpret_cast<char *>(raw.data()), static_cast<std::streamsize>(raw.size()));
   const size_t items = static_cast<size_t>(file.gcount()) / elem;

   std::ostringstream text;
   if (!cfg.is_relative_search) {
      text << std::hex << std::uppercase;
      for (size_t i = 0; i < items; i++) {
         text.width(elem * 2);
         text.fill('0');
         text << static_cast<uint64_t>(element_at<T>(&raw[i * elem], cfg.endianness));
         if (i + 1 < items) {
            text << " ";
         }
      }
      return text.str();
   }
   const bool ascii = cfg.custom_char_seq.empty();
   std::unordered_map<T, std::string> glyph;
   for (const auto &[symbol, value] : values_map) {
      if (ascii && (symbol == U'a' || symbol == U'A')) {
         for (int k = 0; k < 26; k++) {
            glyph[static_cast<T>(value + k)] = utf8_of(symbol + k);
         }
      }
      else {
         glyph[value] = utf8_of(symbol);
      }
   }
   for (size_t i = 0; i < items; i++) {
      auto it = glyph.find(element_at<T>(&raw[i * elem], cfg.endianness));
      text << (it == glyph.end() ? std::string("#") : it->second);
   }
   return text.str();
}

} // namespace

template <typename DataType>
std::vector<mmoore::SearchResult<DataType>> mmoore::SearchEngine<DataType>::run(ProgressCallback on_progress,
                                                                                std::atomic<bool> &abort_flag,
                                                                                bool generate_previews)
{
   using namespace mmoore_amd;
   std::vector<SearchResult<DataType>> results;

   if (!std::filesystem::exists(config.file_path)) {
      throw std::runtime_error("File not found");
   }
   on_progress(0, SearchStep::Initializing);

   const uint64_t file_size = std::filesystem::file_size(config.file_path);
   MonkeyMoore<DataType> matcher = config.is_relative_search
                                      ? MonkeyMoore<DataType>(config.keyword, config.wildcard, config.custom_char_seq)
                                      : MonkeyMoore<DataType>(config.reference_values);
   const MatcherState &st = matcher.state();

   const uint64_t block = static_cast<uint32_t>(config.preferred_search_block_size);
   if (block == 0) {
      throw std::runtime_error("preferred_search_block_size must be positive");
   }
   const uint64_t overlap = static_cast<uint64_t>(st.plan.L - 1) * sizeof(DataType);
   const uint64_t num_blocks = (file_size + block - 1) / block;
   const uint64_t blocks_per_partition = std::max<uint64_t>(1, kPartitionBytes / block);
   const float progress_step = num_blocks ? 100.0f / static_cast<float>(num_blocks) : 0.0f;
   float progress = 0.0f;

   on_progress(0, SearchStep::Searching);

   std::ifstream file(config.file_path, std::ios::binary);
   if (!file.is_open() && file_size) {
      throw std::runtime_error("Failed to open file: " + config.file_path.string());
   }
   mmh_ctx *ctx = thread_context();
   std::vector<uint64_t> offsets(4096), rom_offsets;
   std::vector<uint8_t> under;                              // the elements under every match of a partition
   const int big_endian = config.endianness == Endianness::Big ? 1 : 0;
   const uint32_t match_bytes = st.plan.L * static_cast<uint32_t>(sizeof(DataType));
   const std::string path = config.file_path.string();
   file.close();

   for (uint64_t first_block = 0; first_block < num_blocks; first_block += blocks_per_partition) {
      const uint64_t nblk = std::min(blocks_per_partition, num_blocks - first_block);
      const uint64_t base = first_block * block;                                 // 64-bit, unlike the reference
      const uint64_t want = std::min(nblk * block + overlap, file_size - base);  // pattern-length overlap
      // file -> HBM through parallel readers and overlapped copies (mm_ingest.hip); no host copy is kept
      if (mmh_rom_load_file(ctx, path.c_str(), base, want, 0) != MMH_OK) {
         const std::string why = mmh_last_error();
         if (why.find("short read") != std::string::npos) {
            throw std::runtime_error("Short read from " + path);
         }
         throw_last_error("streaming a file partition to the GPU failed");
      }
      uint64_t count = 0;
      for (;;) {
         int rc = mmh_scan(ctx, &st.plan, block, big_endian, base, offsets.data(), offsets.size(), &count);
         if (rc == MMH_E_CAPACITY) {
            offsets.resize(count + 16);
            continue;
         }
         if (rc != MMH_OK) {
            throw_last_error("GPU scan failed");
         }
         break;
      }
      if (count && !st.value_scan) {
         rom_offsets.resize(count);
         for (uint64_t i = 0; i < count; i++) {
            rom_offsets[i] = offsets[i] - base;
         }
         under.resize(count * match_bytes);
         if (mmh_rom_gather(ctx, rom_offsets.data(), count, match_bytes, under.data()) != MMH_OK) {
            throw_last_error("fetching the matched elements failed");
         }
      }
      for (uint64_t i = 0; i < count; i++) {
         const uint8_t *at = st.value_scan ? nullptr : under.data() + i * match_bytes;
         const Endianness order = config.endianness;
         auto elem = [at, order](int k) { return element_at<DataType>(at + static_cast<size_t>(k) * sizeof(DataType), order); };
         results.push_back({offsets[i], build_values_map<DataType>(st, elem), std::string()});
      }
      // one progress tick per reference block, then the abort poll (search_engine.cpp:161-187)
      for (uint64_t b = 0; b < nblk; b++) {
         progress += progress_step;
         on_progress(static_cast<int>(progress), SearchStep::Searching);
         if (abort_flag) {
            return {};
         }
      }
      if (abort_flag) {
         return {};
      }
   }
   if (abort_flag) {
      return {};
   }

   on_progress(100, SearchStep::GeneratingPreviews);
   // partitions are scanned in file order and each returns ascending offsets, so the list
   // is already ordered the way the reference's std::sort leaves it (:193-197)

   if (generate_previews && !results.empty()) {
      std::ifstream preview_file(config.file_path, std::ios::binary);
      if (!preview_file.is_open()) {
         throw std::runtime_error("Failed to open file to generate previews: " + config.file_path.string());
      }
      for (auto &r : results) {
         r.preview = make_preview<DataType>(config, preview_file, file_size, r.offset, r.values_map);
      }
   }
   return results;
}

template class mmoore::SearchEngine<uint8_t>;
template class mmoore::SearchEngine<uint16_t>;
