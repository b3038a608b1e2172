// SPDX-License-Identifier: GPL-3.0-or-later
// search_engine.cpp -- SearchEngine<T>::run on the MI355X engine.
//
// Replaces the reference's block dispatcher (src/core/search_engine.cpp:23-216): the file
// is streamed to HBM in block-aligned partitions, each partition is scanned on the GPU
// with the reference's block semantics, results are byte offsets in the file.  Host work
// that remains here: file I/O, the progress / abort protocol, the per-match equivalency
// map and the optional preview text (search_engine.cpp:256-348 restated).
#include "mmoore/search_engine.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
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

namespace {

// ---- the GPUs a run() fans its partitions over ---------------------------------------------
//
// The reference keeps `preferred_num_threads` workers busy with blocks (search_engine.cpp:66-188);
// here the unit of work is a block-aligned partition of the file and the workers are GPUs: every
// device streams its own partition over its own PCIe link, scans it, and the ascending offset
// lists are gathered over RCCL (mmh_scan_multi).  Small files stay on one device -- bringing up
// eight contexts and a communicator for a 16 MiB ROM would cost more than the search.
//   MMOORE_HIP_DEVICES=n   use at most n devices (default: all visible)
//   MMOORE_HIP_DEVICE_LIST=a,b,...  the devices to use, in rank order (default: MMOORE_HIP_DEVICE, MMOORE_HIP_DEVICE + 1, ...)
//   MMOORE_HIP_MULTI=1     take the multi-device path even with one device (tests: communicator of one); with a device
//                          list: over ALL listed devices whatever the file's size (tests: two contexts on one GPU through
//                          the RCCL stand-in of tests/shim -- RCCL itself refuses a device that is listed twice)
constexpr uint64_t kMinBytesPerDevice = 1ull << 30;

std::vector<int> device_list()
{
   std::vector<int> ids;
   const char *env = std::getenv("MMOORE_HIP_DEVICE_LIST");
   for (const char *p = env; p && *p;) {
      char *end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) {
         break;
      }
      ids.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
   }
   return ids;
}

struct DeviceSet {
   std::mutex lock;                         // one run() at a time drives the set
   std::vector<mmh_ctx *> ctx;
};

// The set lives for the rest of the process and is deliberately never destroyed: a static destructor would call
// ncclCommDestroy / hipFree at exit, when HIP or RCCL (other shared objects, unspecified destruction order) may be
// gone already -- a known source of exit-time hangs.  The driver reclaims everything when the process ends.
DeviceSet &device_set()
{
   static DeviceSet *set = new DeviceSet();
   return *set;
}

int devices_wanted(uint64_t file_size, bool *forced)
{
   const char *multi = std::getenv("MMOORE_HIP_MULTI");
   *forced = multi && *multi && *multi != '0';
   int visible = 0;
   if (mmh_device_count(&visible) != MMH_OK || visible < 1) {
      return 1;                             // thread_context() reports the missing device properly
   }
   // (the set starts at device MMOORE_HIP_DEVICE: only the devices from there on count -- or it is the explicit list)
   const std::vector<int> listed = device_list();
   const char *first_env = std::getenv("MMOORE_HIP_DEVICE");
   const int first = first_env ? std::max(0, std::atoi(first_env)) : 0;
   visible = listed.empty() ? std::max(1, visible - first) : (int)listed.size();
   const char *cap = std::getenv("MMOORE_HIP_DEVICES");
   int most = cap && std::atoi(cap) > 0 ? std::min(visible, std::atoi(cap)) : visible;
   if (*forced && !listed.empty()) {
      return most;
   }
   const uint64_t by_size = std::max<uint64_t>(1, file_size / kMinBytesPerDevice);
   return (int)std::min<uint64_t>((uint64_t)most, by_size);
}

// contexts 0 .. n-1 of the set, rank i of one communicator of n (rebuilt when n changes)
void ensure_devices(DeviceSet &set, int n)
{
   using mmoore_amd::throw_last_error;
   int rank = 0, nranks = 0;
   if ((int)set.ctx.size() == n && mmh_comm_info(set.ctx[0], &rank, &nranks) == MMH_OK && nranks == n) {
      return;
   }
   for (mmh_ctx *c : set.ctx) {
      mmh_destroy(c);
   }
   set.ctx.clear();
   const char *env = std::getenv("MMOORE_HIP_DEVICE");
   const int first = env ? std::atoi(env) : 0;
   const std::vector<int> listed = device_list();
   int visible = 0;
   if (mmh_device_count(&visible) != MMH_OK || first < 0 || (listed.empty() ? first + n > visible : n > (int)listed.size())) {
      throw std::runtime_error("MMOORE_HIP_DEVICE=" + std::to_string(first) + " with " + std::to_string(n) + " devices wanted, but " +
                               std::to_string(listed.empty() ? visible : (int)listed.size()) + " are visible / listed");
   }
   for (int i = 0; i < n; i++) {
      mmh_ctx *c = nullptr;
      if (mmh_create(listed.empty() ? first + i : listed[i], &c) != MMH_OK) {
         throw_last_error("MI355X engine unavailable (there is no CPU fallback)");
      }
      (void)mmh_set_timing(c, 0);              // (nobody behind this API asks for device timings)
      set.ctx.push_back(c);
   }
   if (mmh_comm_init_all(set.ctx.data(), n) != MMH_OK) {
      throw_last_error("RCCL communicator over the visible GPUs failed");
   }
}

struct Partition {
   uint64_t base = 0, bytes = 0, blocks = 0;
};

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

   bool forced_multi = false;
   const int ndev = devices_wanted(file_size, &forced_multi);
   const bool multi = ndev > 1 || forced_multi;
   DeviceSet &set = device_set();
   std::unique_lock<std::mutex> hold(set.lock, std::defer_lock);
   std::vector<mmh_ctx *> ctxs;
   if (multi) {
      hold.lock();
      ensure_devices(set, ndev);
      ctxs = set.ctx;
   }
   else {
      ctxs.push_back(thread_context());
   }
   std::vector<uint64_t> offsets(4096), rom_offsets;
   std::vector<uint8_t> under;                              // the elements under every match of a round
   const int big_endian = config.endianness == Endianness::Big ? 1 : 0;
   const uint32_t match_bytes = st.plan.L * static_cast<uint32_t>(sizeof(DataType));
   const std::string path = config.file_path.string();

   // a round = one partition per device, consecutive in the file; rounds follow each other
   for (uint64_t first_block = 0; first_block < num_blocks; first_block += blocks_per_partition * ctxs.size()) {
      const uint64_t round_blocks = std::min<uint64_t>(blocks_per_partition * ctxs.size(), num_blocks - first_block);
      std::vector<Partition> parts(ctxs.size());
      for (size_t i = 0; i < ctxs.size(); i++) {
         // whole blocks, dealt out evenly in device order (the rule of mmh_partition), 64-bit offsets unlike the reference
         const uint64_t b0 = first_block + round_blocks * i / ctxs.size();
         const uint64_t b1 = first_block + round_blocks * (i + 1) / ctxs.size();
         parts[i].blocks = b1 - b0;
         parts[i].base = b0 * block;
         parts[i].bytes = b1 > b0 ? std::min((b1 - b0) * block + overlap, file_size - parts[i].base) : 0;   // pattern-length overlap
      }
      // file -> HBM through parallel readers and overlapped copies (mm_ingest.hip), every device at once over its
      // own PCIe link; no host copy is kept.  The ingest is where a file search spends its time (80 ms per 4 GiB
      // against < 1 ms of scan), so it is what reports progress and listens for the abort flag: the loaders run on
      // threads of their own, this thread polls what has landed -- one progress tick per reference block whose bytes
      // are in HBM, like the reference's workers tick as they finish blocks (search_engine.cpp:161-165) -- and
      // raises the loaders' abort word as soon as the caller's flag is up (:177-187: polled every 5 ms there;
      // here every 0.2 ms; the library's supervising thread looks at the word every 0.1 ms and returns without
      // waiting for readers inside a blocking call -- measured 0.1-0.5 ms from flag to return on a 4 GiB file).
      std::vector<std::string> load_error(ctxs.size());
      std::vector<int> load_rc(ctxs.size(), MMH_OK);
      std::vector<uint64_t> landed(ctxs.size(), 0);           // raised by the library (relaxed atomics)
      int32_t stop_loading = 0;
      std::atomic<int> loading{(int)ctxs.size()};
      auto load = [&](size_t i) {
         load_rc[i] = mmh_rom_load_file_watched(ctxs[i], path.c_str(), parts[i].base, parts[i].bytes, 0, &stop_loading, &landed[i]);
         if (load_rc[i] != MMH_OK) {
            load_error[i] = mmh_last_error();
         }
         loading--;
      };
      std::vector<std::thread> loaders;
      for (size_t i = 0; i < ctxs.size(); i++) {
         loaders.emplace_back(load, i);
      }
      uint64_t ticked = 0;                                     // progress ticks of this round so far
      bool aborted = false;
      auto tick_landed_blocks = [&]() {
         uint64_t bytes = 0;
         for (size_t i = 0; i < ctxs.size(); i++) {
            bytes += __atomic_load_n(&landed[i], __ATOMIC_RELAXED);
         }
         // the round's last tick waits for the scan: progress only reaches a round's share once its results exist
         const uint64_t may = std::min<uint64_t>(bytes / block, round_blocks - 1);
         while (ticked < may && !aborted) {
            ticked++;
            progress += progress_step;
            on_progress(static_cast<int>(progress), SearchStep::Searching);
            aborted = abort_flag;
         }
      };
      const auto poll_start = std::chrono::steady_clock::now();
      while (loading > 0 && !aborted) {
         // (small files are in HBM within microseconds: spin first, sleep only once the load turns out to take a while)
         if (std::chrono::steady_clock::now() - poll_start < std::chrono::microseconds(300)) {
            std::this_thread::yield();
         }
         else {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
         }
         aborted = aborted || abort_flag;
         if (!aborted) {
            tick_landed_blocks();
         }
      }
      if (aborted) {
         __atomic_store_n(&stop_loading, 1, __ATOMIC_RELAXED);
      }
      for (auto &t : loaders) {
         t.join();
      }
      if (aborted || abort_flag) {
         return {};
      }
      for (size_t i = 0; i < ctxs.size(); i++) {
         const std::string &why = load_error[i];
         if (load_rc[i] == MMH_OK) {
            continue;
         }
         if (why.find("short read") != std::string::npos) {
            throw std::runtime_error("Short read from " + path);
         }
         throw std::runtime_error("streaming a file partition to the GPU failed: " + why);
      }
      uint64_t count = 0;
      for (;;) {
         int rc;
         if (multi) {
            std::vector<uint64_t> bases;
            for (const Partition &p : parts) {
               bases.push_back(p.base);
            }
            rc = mmh_scan_multi(ctxs.data(), (int)ctxs.size(), &st.plan, block, big_endian, bases.data(), offsets.data(),
                                offsets.size(), &count);
         }
         else {
            rc = mmh_scan(ctxs[0], &st.plan, block, big_endian, parts[0].base, offsets.data(), offsets.size(), &count);
         }
         if (rc == MMH_E_CAPACITY) {
            offsets.resize(count + 16);
            continue;
         }
         if (rc != MMH_OK) {
            throw_last_error("GPU scan failed");
         }
         break;
      }
      if (abort_flag) {
         return {};
      }
      if (count && !st.value_scan) {
         // the elements under the matches, from the device that holds them: a match that starts in
         // partition i lies entirely inside it (that is what the overlap is for)
         under.resize(count * match_bytes);
         uint64_t at = 0;
         for (size_t i = 0; i < ctxs.size() && at < count; i++) {
            const uint64_t limit = i + 1 < ctxs.size() ? parts[i + 1].base : ~0ull;
            uint64_t end = at;
            while (end < count && offsets[end] < limit) {
               end++;
            }
            rom_offsets.resize(end - at);
            for (uint64_t k = at; k < end; k++) {
               rom_offsets[k - at] = offsets[k] - parts[i].base;
            }
            if (end > at && mmh_rom_gather(ctxs[i], rom_offsets.data(), end - at, match_bytes, under.data() + at * match_bytes) != MMH_OK) {
               throw_last_error("fetching the matched elements failed");
            }
            at = end;
         }
      }
      for (uint64_t i = 0; i < count; i++) {
         const uint8_t *at = st.value_scan ? nullptr : under.data() + i * match_bytes;
         const Endianness order = config.endianness;
         auto elem = [at, order](int k) { return element_at<DataType>(at + static_cast<size_t>(k) * sizeof(DataType), order); };
         results.push_back({offsets[i], build_values_map<DataType>(st, elem), std::string()});
      }
      // the round's remaining ticks (one per reference block in all), each followed by the abort poll (search_engine.cpp:161-187)
      for (uint64_t b = ticked; b < round_blocks; b++) {
         progress += progress_step;
         on_progress(static_cast<int>(progress), SearchStep::Searching);
         if (abort_flag) {
            return {};
         }
      }
   }
   if (abort_flag) {
      return {};
   }

   on_progress(100, SearchStep::GeneratingPreviews);
   // rounds follow the file and each returns ascending offsets, so the list is already ordered
   // the way the reference's std::sort leaves it (:193-197)

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
