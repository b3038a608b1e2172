// SPDX-License-Identifier: GPL-3.0-or-later
// search_engine.hpp -- SearchEngine<T>, the file-level driver of the mmoore API, backed by
// the MI355X engine.
//
// Public surface = the reference's include/mmoore/search_engine.hpp:14-59 (SearchResult,
// SearchConfig with the same fields and defaults, SearchStep, SearchEngine<T>::run).
// What changed behind it: the thread-per-block dispatcher (src/core/search_engine.cpp:66-188)
// is gone.  The file is streamed to HBM in block-aligned partitions and each partition is
// scanned with the reference's block semantics (chain restart per block and per byte
// alignment, (L-1)*sizeof(T) overlap) by the GPU; offsets are global byte offsets.
//   * preferred_search_block_size keeps its meaning (it changes results in the reference too)
//   * preferred_num_threads is accepted and ignored: it never changed results
//   * block offsets are 64 bit (the reference multiplies in 32 bit, :241-242, and breaks
//     above 4 GiB)
//   * progress: (0, Initializing), (0, Searching), one (pct, Searching) per reference block,
//     (100, GeneratingPreviews) -- blocks + 3 calls, monotone, all from the calling thread; a block's tick
//     comes when its bytes have landed in HBM (the ingest is where the time goes), a round's last one
//     after its scan
//   * abort: polled after every progress callback, every 0.2 ms while a partition streams to the GPU (the
//     readers stop at the next 4 MiB piece) and between ingest, scan and gather; run() returns {}
//   * a missing file throws std::runtime_error("File not found") before any callback
#ifndef MMOORE_AMD_SEARCH_ENGINE_HPP
#define MMOORE_AMD_SEARCH_ENGINE_HPP

#include <atomic>
#include <filesystem>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "mmoore/byteswap.hpp"
#include "mmoore/monkey_moore.hpp"

namespace mmoore {

// One match.  Member order and types are those of the reference so that aggregate
// initialisation ({offset, map, preview}) in existing tests keeps working.
template <typename DataType>
struct SearchResult {
   // where the match starts, as a BYTE offset from the beginning of the file (also for
   // 16-bit searches; odd offsets occur)
   uint64_t offset;
   // symbol -> element value for this particular match ('A'/'a' bases for ASCII keywords,
   // every symbol for custom sequences, empty for value scans)
   typename MonkeyMoore<DataType>::equivalency_map values_map;
   // the bytes around the match decoded through values_map; empty unless requested
   std::string preview;
};

// What to search for and how.  Same fields and defaults as the reference.
struct SearchConfig {
   std::filesystem::path file_path;

   // true: relative search for `keyword`; false: value scan of `reference_values`
   bool is_relative_search = true;
   // byte order of 16-bit elements in the file (ignored for 8-bit searches)
   mmoore::Endianness endianness = Endianness::Little;

   std::vector<CharType> keyword;                  // UTF-32 code points
   std::vector<CharType> custom_char_seq = {};     // optional alphabet: symbols are valued by their index in it
   CharType wildcard = '*';                        // matches any element

   std::vector<short> reference_values = {};

   // accepted for compatibility, not used: the GPU engine has no worker threads and the
   // value never influenced results
   int preferred_num_threads = std::thread::hardware_concurrency();
   // the reference's block size in bytes; it DOES influence results (the skip chain
   // restarts at every block), so it is honoured exactly
   int preferred_search_block_size = 524288;
   // elements of context shown in a preview
   int preferred_preview_width = 50;
};

// phase reported to the progress callback
enum SearchStep { Initializing, Searching, GeneratingPreviews, Aborting };

template <typename DataType>
class SearchEngine {
public:
   using ProgressCallback = std::function<void(int, const SearchStep)>;

   explicit SearchEngine(const SearchConfig &cfg) : config(cfg) {}

   std::vector<SearchResult<DataType>> run(ProgressCallback on_progress, std::atomic<bool> &abort_flag,
                                           bool generate_previews = false);

private:
   SearchConfig config;
};

} // namespace mmoore

#endif
