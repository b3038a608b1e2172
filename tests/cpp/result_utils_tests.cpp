// SPDX-License-Identifier: GPL-3.0-or-later
// result_utils_tests.cpp -- include/mmoore/result_utils.hpp on hand-derived vectors (CPU only, nothing links
// against the GPU library).  Each expectation is worked out by hand from the reference's GUI code:
//   src/gui/monkey_frame.cpp:1215-1273 (ShowResults), src/gui/dialogs/table_creator.cpp:100-108, :164-194.
#include <cstdio>
#include <string>
#include <vector>

#include "mmoore/result_utils.hpp"

static int failures = 0, checks = 0;
#define CHECK(cond) do { checks++; if (!(cond)) { failures++; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)
#define CHECK_EQ(a, b) do { checks++; if (!((a) == (b))) { failures++; std::printf("FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); } } while (0)

using mmoore::Endianness;

int main()
{
   using Map8 = MonkeyMoore<uint8_t>::equivalency_map;
   using Map16 = MonkeyMoore<uint16_t>::equivalency_map;

   // ---- offsets (:1243-1244: "0x%llX" / "%lld") ------------------------------------------------------------
   CHECK_EQ(mmoore::format_offset(0, true), "0x0");
   CHECK_EQ(mmoore::format_offset(8000, true), "0x1F40");
   CHECK_EQ(mmoore::format_offset(8000, false), "8000");
   CHECK_EQ(mmoore::format_offset(0x123456789ABCull, true), "0x123456789ABC");
   CHECK_EQ(mmoore::format_offset(68719476736ull, false), "68719476736");          // 64 GiB: beyond 32 bits

   // ---- values (:1250-1263): "%c=%0NX " per entry, N = 2 * sizeof(T), symbol order -------------------------
   const Map8 ascii8{{U'A', 0x41}, {U'a', 0x61}};
   CHECK_EQ(mmoore::format_values<uint8_t>(ascii8, Endianness::Little), "A=41 a=61 ");
   CHECK_EQ(mmoore::format_values<uint8_t>(ascii8, Endianness::Big), "A=41 a=61 ");   // single bytes never swap
   const Map8 shifted{{U'A', 0x0A}, {U'a', 0xF3}};
   CHECK_EQ(mmoore::format_values<uint8_t>(shifted, Endianness::Little), "A=0A a=F3 ");
   // 16 bit: after a little-endian search the value prints byte-reversed on a little-endian host
   // (byteorder_little ? swap_on_le : swap_on_be), after a big-endian search as it is
   const Map16 ascii16{{U'A', 0x0041}, {U'a', 0x0161}};
   CHECK_EQ(mmoore::format_values<uint16_t>(ascii16, Endianness::Little), "A=4100 a=6101 ");
   CHECK_EQ(mmoore::format_values<uint16_t>(ascii16, Endianness::Big), "A=0041 a=0161 ");
   // custom sequence: every symbol of the sequence, UTF-8 for what is not ASCII (hiragana a, i)
   const Map16 kana{{U'あ', 0x0100}, {U'い', 0x0101}};
   CHECK_EQ(mmoore::format_values<uint16_t>(kana, Endianness::Big), "\xE3\x81\x82=0100 \xE3\x81\x84=0101 ");
   CHECK_EQ(mmoore::format_values<uint8_t>(Map8{}, Endianness::Little), "");         // value scans: empty maps

   // ---- which results are shown (:1223, :1236-1241) --------------------------------------------------------
   std::vector<mmoore::SearchResult<uint8_t>> results = {
      {0x10, ascii8, "p0"}, {0x20, shifted, "p1"}, {0x30, ascii8, "p2"}, {0x40, Map8{{U'A', 0x41}}, "p3"}, {0x50, shifted, "p4"}};
   CHECK_EQ(mmoore::visible_results<uint8_t>(results, false), (std::vector<size_t>{0, 1, 3}));
   CHECK_EQ(mmoore::visible_results<uint8_t>(results, true), (std::vector<size_t>{0, 1, 2, 3, 4}));
   const auto rows = mmoore::result_rows<uint8_t>(results, false, true, Endianness::Little);
   CHECK_EQ(rows.size(), 3u);
   CHECK_EQ(rows[1].index, 1u);
   CHECK_EQ(rows[1].offset, "0x20");
   CHECK_EQ(rows[1].values, "A=0A a=F3 ");
   CHECK_EQ(rows[1].preview, "p1");
   CHECK_EQ(rows[2].offset, "0x40");
   CHECK_EQ(mmoore::result_rows<uint8_t>(results, true, false, Endianness::Little)[4].offset, "80");
   CHECK(mmoore::result_rows<uint8_t>({}, false, true, Endianness::Little).empty());
   // value scans carry empty maps: all equal, so only the first row shows unless "show all"
   std::vector<mmoore::SearchResult<uint8_t>> scans = {{4, {}, ""}, {21, {}, ""}};
   CHECK_EQ(mmoore::visible_results<uint8_t>(scans, false), (std::vector<size_t>{0}));

   // ---- table export (table_creator.cpp:164-194) -----------------------------------------------------------
   // 'A' -> 26 rows 41..5A = A..Z, 'a' -> 61..7A = a..z; ordered by the hex string
   auto t = mmoore::table_rows<uint8_t>(ascii8, Endianness::Little);
   CHECK_EQ(t.size(), 52u);
   CHECK_EQ(t.begin()->first, "41");
   CHECK_EQ(t.begin()->second, "A");
   CHECK_EQ(t["5A"], "Z");
   CHECK_EQ(t["61"], "a");
   CHECK_EQ(t.rbegin()->first, "7A");
   CHECK_EQ(t.rbegin()->second, "z");
   // wrap at max + 1 (:177-178): 'a' based at 0xF3 -> F3..FF = a..m, then 00..0C = n..z
   t = mmoore::table_rows<uint8_t>(Map8{{U'a', 0xF3}}, Endianness::Little);
   CHECK_EQ(t.size(), 26u);
   CHECK_EQ(t["F3"], "a");
   CHECK_EQ(t["FF"], "m");
   CHECK_EQ(t["00"], "n");
   CHECK_EQ(t["0C"], "z");
   CHECK_EQ(t.begin()->first, "00");                                               // ordered by the string, not by letter
   // overlapping alphabets: 'A' at 0x50 covers 50..69, 'a' at 0x60 covers 60..79 -- map order 'A' < 'a', so the
   // lower-case rows overwrite 60..69 (K..T become a..j)
   t = mmoore::table_rows<uint8_t>(Map8{{U'A', 0x50}, {U'a', 0x60}}, Endianness::Little);
   CHECK_EQ(t.size(), 42u);
   CHECK_EQ(t["5F"], "P");
   CHECK_EQ(t["60"], "a");
   CHECK_EQ(t["69"], "j");
   // 16 bit, little-endian search: keys are the byte-reversed values, and ordered as such; wrap at 65536
   auto t16 = mmoore::table_rows<uint16_t>(Map16{{U'A', 0xFFFE}}, Endianness::Little);
   CHECK_EQ(t16.size(), 26u);
   CHECK_EQ(t16["FEFF"], "A");
   CHECK_EQ(t16["FFFF"], "B");
   CHECK_EQ(t16["0000"], "C");
   CHECK_EQ(t16["0100"], "D");                                                     // value 0x0001
   CHECK_EQ(t16["1700"], "Z");                                                     // value 0x0017
   t16 = mmoore::table_rows<uint16_t>(Map16{{U'A', 0xFFFE}}, Endianness::Big);
   CHECK_EQ(t16["FFFE"], "A");
   CHECK_EQ(t16["0001"], "D");
   // custom sequence: one row per symbol, no expansion -- except that a sequence holding 'a' or 'A' itself
   // gets the 26 letters from there (the dialog tests the symbol, not the search mode)
   t16 = mmoore::table_rows<uint16_t>(kana, Endianness::Big);
   CHECK_EQ(t16.size(), 2u);
   CHECK_EQ(t16["0101"], "\xE3\x81\x84");
   t = mmoore::table_rows<uint8_t>(Map8{{U'#', 0x05}, {U'a', 0x10}}, Endianness::Little);
   CHECK_EQ(t.size(), 27u);
   CHECK_EQ(t["05"], "#");
   CHECK_EQ(t["29"], "z");
   // the saved text (:100-108)
   CHECK_EQ(mmoore::table_text(mmoore::table_rows<uint16_t>(kana, Endianness::Big)), "0100=\xE3\x81\x82\r\n0101=\xE3\x81\x84\r\n");
   CHECK_EQ(mmoore::table_text({}), "");

   std::printf("%d checks, %d failures\n", checks, failures);
   return failures ? 1 : 0;
}
