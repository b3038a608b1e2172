// SPDX-License-Identifier: GPL-3.0-or-later
// result_utils.hpp -- what a front end does with a result list (MI355X build; host only, header only).
//
// The reference keeps this logic inside its wxWidgets GUI; here it is a small library
// over the public result types so that a command line (tools/mmoore_search.cpp) or any other front end
// shows the same rows and exports the same tables:
//   visible_results / result_rows   src/gui/monkey_frame.cpp:1215-1273  (MonkeyFrame::ShowResults: one row per
//                                   distinct equivalency map unless "show all", offset as 0x%llX or %lld,
//                                   "%c=%0NX " per map entry with the value's bytes in the order the GUI shows)
//   table_rows / table_text         src/gui/dialogs/table_creator.cpp:164-194 (InitTableData: the 26-letter
//                                   expansion of the 'A' / 'a' bases with wrap at max + 1, rows keyed and
//                                   ordered by the hex string) and :100-108 (the saved "HEX=symbol\r\n" text)
// Nothing here touches the GPU.
#ifndef MMOORE_AMD_RESULT_UTILS_HPP
#define MMOORE_AMD_RESULT_UTILS_HPP

#include <cstdint>
#include <cstdio>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "mmoore/byteswap.hpp"
#include "mmoore/search_engine.hpp"

namespace mmoore {

// UTF-8 of one code point (the GUI formats symbols with %c on a wide string)
inline std::string symbol_utf8(CharType cp)
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

// The value as the GUI prints it (monkey_frame.cpp:1257-1260, table_creator.cpp:182,189): after a little-endian
// search the bytes are reversed on a little-endian host, after a big-endian search on a big-endian host -- on the
// (little-endian) hosts this runs on, a 16-bit value of a little-endian search prints in FILE byte order
// (0x0041 -> "4100"), one of a big-endian search as the number it is ("0041").  Single bytes never change.
template <typename T>
T displayed_value(T value, Endianness searched)
{
   return searched == Endianness::Little ? swap_on_little_endian<T>(value) : swap_on_big_endian<T>(value);
}

// upper-case hex, two digits per byte of T
template <typename T>
std::string hex_of(T value)
{
   char buf[2 * sizeof(T) + 1];
   std::snprintf(buf, sizeof buf, "%0*llX", static_cast<int>(2 * sizeof(T)), static_cast<unsigned long long>(value));
   return buf;
}

// a match offset the way the result list shows it: "0x1F40" or "8000" (monkey_frame.cpp:1243-1244)
inline std::string format_offset(uint64_t offset, bool hex)
{
   char buf[32];
   if (hex) {
      std::snprintf(buf, sizeof buf, "0x%llX", static_cast<unsigned long long>(offset));
   }
   else {
      std::snprintf(buf, sizeof buf, "%lld", static_cast<long long>(offset));
   }
   return buf;
}

// "A=41 a=61 " -- every entry of the equivalency map in symbol order, each followed by a blank (:1250-1263)
template <typename T>
std::string format_values(const typename MonkeyMoore<T>::equivalency_map &values, Endianness searched)
{
   std::string text;
   for (const auto &[symbol, value] : values) {
      text += symbol_utf8(symbol);
      text += '=';
      text += hex_of<T>(displayed_value<T>(value, searched));
      text += ' ';
   }
   return text;
}

// Indices of the results a list shows, in order.  show_all: every one.  Otherwise a result whose equivalency map
// equals that of a result already shown is left out (:1223, :1236-1241) -- matches of the same text encoding in
// different places of a ROM collapse into one row.
template <typename T>
std::vector<size_t> visible_results(const std::vector<SearchResult<T>> &results, bool show_all)
{
   std::vector<size_t> shown;
   std::vector<const typename MonkeyMoore<T>::equivalency_map *> seen;
   for (size_t i = 0; i < results.size(); i++) {
      bool duplicate = false;
      for (const auto *m : seen) {
         if (*m == results[i].values_map) {
            duplicate = true;
            break;
         }
      }
      if (duplicate) {
         continue;
      }
      if (!show_all) {
         seen.push_back(&results[i].values_map);
      }
      shown.push_back(i);
   }
   return shown;
}

struct ResultRow {
   size_t index;            // into the result vector (the GUI's item data)
   std::string offset;      // column 0
   std::string values;      // column 1 of a relative search
   std::string preview;     // column 2 of a relative search, column 1 of a value scan
};

// the rows of the result list (MonkeyFrame::ShowResults); rows.size() is the counter label's number
template <typename T>
std::vector<ResultRow> result_rows(const std::vector<SearchResult<T>> &results, bool show_all, bool hex_offsets, Endianness searched)
{
   std::vector<ResultRow> rows;
   for (size_t i : visible_results<T>(results, show_all)) {
      rows.push_back({i, format_offset(results[i].offset, hex_offsets), format_values<T>(results[i].values_map, searched),
                      results[i].preview});
   }
   return rows;
}

// The table a result exports: hex string -> symbol, ordered by the hex string (the dialog keeps them in a map keyed
// by it).  Entries 'A' and 'a' stand for their whole alphabets: 26 consecutive values each, wrapping to 0 behind the
// element type's maximum (table_creator.cpp:173-186); any other entry is one row (:187-191).  Later entries
// overwrite earlier ones that print the same hex.
template <typename T>
std::map<std::string, std::string> table_rows(const typename MonkeyMoore<T>::equivalency_map &values, Endianness searched)
{
   std::map<std::string, std::string> rows;
   for (const auto &[symbol, first_value] : values) {
      if (symbol == U'A' || symbol == U'a') {
         int counter = first_value;
         for (int letter = 0; letter < 26; letter++, counter++) {
            if (counter == static_cast<int>(std::numeric_limits<T>::max()) + 1) {
               counter = 0;
            }
            rows[hex_of<T>(displayed_value<T>(static_cast<T>(counter), searched))] = symbol_utf8(symbol + letter);
         }
      }
      else {
         rows[hex_of<T>(displayed_value<T>(first_value, searched))] = symbol_utf8(symbol);
      }
   }
   return rows;
}

// the text of a saved table: "HEX=symbol" lines, CR LF terminated (table_creator.cpp:100-108); UTF-8
inline std::string table_text(const std::map<std::string, std::string> &rows)
{
   std::string text;
   for (const auto &[hex, symbol] : rows) {
      text += hex + "=" + symbol + "\r\n";
   }
   return text;
}

} // namespace mmoore

#endif
