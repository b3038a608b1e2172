// SPDX-License-Identifier: GPL-3.0-or-later
// mmoore_search -- a command line over the include/mmoore facade: SearchEngine<T>::run on the MI355X engine,
// results shown and exported the way the reference's GUI does (include/mmoore/result_utils.hpp; reference
// src/gui/monkey_frame.cpp:483-572 fills the SearchConfig, :1215-1273 lists the results,
// src/gui/dialogs/table_creator.cpp:164-194 exports a result's table).
//
//   mmoore_search [options] FILE KEYWORD            relative search (KEYWORD in UTF-8)
//   mmoore_search [options] --values 1,5,9 FILE     value scan
//     --bits 8|16        element width (8)            --be            16-bit elements are big endian
//     --wildcard C       wildcard symbol ('*')        --charseq STR   custom character sequence
//     --block N          search block size (524288)   --preview [W]   previews, W elements wide (50)
//     --all              one row per match (default: one per distinct equivalency map)
//     --dec              decimal offsets              --table OUT     save the first shown row's table to OUT
// Output: one tab-separated row per shown result -- offset, values, preview -- and the count on stderr.
// Exit code: 0 found something, 1 nothing found, 2 usage / error.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mmoore/result_utils.hpp"
#include "mmoore/search_engine.hpp"

namespace {

std::vector<CharType> code_points(const std::string &utf8)
{
   std::vector<CharType> out;
   for (size_t i = 0; i < utf8.size();) {
      const unsigned char c = static_cast<unsigned char>(utf8[i]);
      int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
      if (extra < 0 || i + extra >= utf8.size() + (extra == 0)) {
         throw std::runtime_error("keyword is not valid UTF-8");
      }
      CharType cp = extra == 0 ? c : c & (0x3F >> extra);
      for (int k = 1; k <= extra; k++) {
         cp = (cp << 6) | (static_cast<unsigned char>(utf8[i + k]) & 0x3F);
      }
      out.push_back(cp);
      i += extra + 1;
   }
   return out;
}

struct Options {
   mmoore::SearchConfig cfg;
   int bits = 8;
   bool show_all = false, hex = true, previews = false;
   std::string table_path;
};

template <typename T>
int run(const Options &o)
{
   mmoore::SearchEngine<T> engine(o.cfg);
   std::atomic<bool> abort{false};
   int last = -1;
   auto progress = [&](int pct, const mmoore::SearchStep) {
      if (pct != last && pct % 10 == 0) {
         std::fprintf(stderr, "\r%3d%%", pct);
         last = pct;
      }
   };
   const auto results = engine.run(progress, abort, o.previews);
   std::fprintf(stderr, "\r");
   const auto rows = mmoore::result_rows<T>(results, o.show_all, o.hex, o.cfg.endianness);
   for (const auto &r : rows) {
      std::printf("%s\t%s\t%s\n", r.offset.c_str(), r.values.c_str(), r.preview.c_str());
   }
   std::fprintf(stderr, "%zu shown, %zu matches\n", rows.size(), results.size());
   if (!o.table_path.empty() && !rows.empty()) {
      const std::string text = mmoore::table_text(mmoore::table_rows<T>(results[rows[0].index].values_map, o.cfg.endianness));
      std::ofstream out(o.table_path, std::ios::binary);
      out.write(text.data(), static_cast<std::streamsize>(text.size()));
      if (!out) {
         throw std::runtime_error("The table file couldn't be created: " + o.table_path);
      }
   }
   return results.empty() ? 1 : 0;
}

int usage()
{
   std::fprintf(stderr, "usage: mmoore_search [--bits 8|16] [--be] [--wildcard C] [--charseq STR] [--block N] [--preview [W]] [--all] "
                        "[--dec] [--table OUT] (FILE KEYWORD | --values a,b,c FILE)\n");
   return 2;
}

} // namespace

int main(int argc, char **argv)
{
   Options o;
   std::vector<std::string> positional;
   try {
      for (int i = 1; i < argc; i++) {
         const std::string a = argv[i];
         auto value = [&]() -> std::string {
            if (i + 1 >= argc) {
               throw std::runtime_error(a + " needs a value");
            }
            return argv[++i];
         };
         if (a == "--bits") o.bits = std::atoi(value().c_str());
         else if (a == "--be") o.cfg.endianness = mmoore::Endianness::Big;
         else if (a == "--wildcard") {
            const auto cps = code_points(value());
            if (cps.size() != 1) {
               throw std::runtime_error("--wildcard takes one symbol");
            }
            o.cfg.wildcard = cps[0];
         }
         else if (a == "--charseq") o.cfg.custom_char_seq = code_points(value());
         else if (a == "--block") o.cfg.preferred_search_block_size = std::atoi(value().c_str());
         else if (a == "--preview") {
            o.previews = true;
            if (i + 1 < argc && std::atoi(argv[i + 1]) > 0 && std::strspn(argv[i + 1], "0123456789") == std::strlen(argv[i + 1])) {
               o.cfg.preferred_preview_width = std::atoi(argv[++i]);
            }
         }
         else if (a == "--all") o.show_all = true;
         else if (a == "--dec") o.hex = false;
         else if (a == "--table") o.table_path = value();
         else if (a == "--values") {
            o.cfg.is_relative_search = false;
            const std::string list = value();
            for (size_t at = 0; at < list.size();) {
               size_t used = 0;
               o.cfg.reference_values.push_back(static_cast<short>(std::stoi(list.substr(at), &used)));
               at += used;
               if (at < list.size() && list[at] == ',') {
                  at++;
               }
            }
         }
         else if (a.rfind("--", 0) == 0) return usage();
         else positional.push_back(a);
      }
      if ((o.bits != 8 && o.bits != 16) || positional.size() != (o.cfg.is_relative_search ? 2u : 1u)) {
         return usage();
      }
      o.cfg.file_path = positional[0];
      if (o.cfg.is_relative_search) {
         o.cfg.keyword = code_points(positional[1]);
         // the GUI's rule (monkey_frame.cpp:1040, :1099): at least three symbols that are not wildcards
         size_t literals = 0;
         for (CharType c : o.cfg.keyword) {
            literals += c != o.cfg.wildcard;
         }
         if (literals < 3) {
            throw std::runtime_error("the keyword needs at least 3 characters that are not wildcards");
         }
      }
      else if (o.cfg.reference_values.size() < 3) {
         throw std::runtime_error("a value scan needs at least 3 values");
      }
      return o.bits == 8 ? run<uint8_t>(o) : run<uint16_t>(o);
   }
   catch (const std::exception &e) {
      std::fprintf(stderr, "mmoore_search: %s\n", e.what());
      return 2;
   }
}
