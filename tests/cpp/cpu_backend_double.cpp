// SPDX-License-Identifier: GPL-3.0-or-later
// TEST DOUBLE, never shipped, never linked into the product: a CPU stand-in for the entries of
// include/mmoore_hip.h that the C++ facade (monkey-moore_amd/host/) calls, so that the facade
// itself -- partition rounds, progress / abort, equivalency maps, preview windows -- can run
// under AddressSanitizer + UndefinedBehaviorSanitizer in the build container (GPU sanitizers
// are not available on the pool).  The plan comes from the real builder (csrc/mm_plan.cpp,
// compiled into the same binary); the "scan" walks it one alignment at a time exactly as the
// header documents mmh_plan_desc (compare from i = L-1 down, jump min(wst, max(skip, 1)),
// match_jump after a match) with the domain rules of mmh_scan (block restarts, both byte
// alignments of 16-bit elements, the odd-end rule of src/core/search_engine.cpp:137-141).
// MMOORE_DOUBLE_DEVICES=n makes it pose as n devices (the multi-device path of run()).
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "mmoore_hip.h"

struct mmh_ctx {
   int device = 0;
   std::vector<uint8_t> rom;
   int rank = 0, nranks = 0;
};

static thread_local std::string g_error;

extern "C" void mmh_set_error(const char *fmt, ...)
{
   char buf[512];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof buf, fmt, ap);
   va_end(ap);
   g_error = buf;
}
extern "C" const char *mmh_last_error(void) { return g_error.c_str(); }

extern "C" int mmh_device_count(int *count)
{
   const char *e = getenv("MMOORE_DOUBLE_DEVICES");
   *count = e && *e ? atoi(e) : 1;
   return MMH_OK;
}

extern "C" int mmh_create(int device, mmh_ctx **out)
{
   int n = 0;
   mmh_device_count(&n);
   if (device < 0 || device >= n) {
      mmh_set_error("double: no device %d", device);
      return MMH_E_DEVICE;
   }
   *out = new mmh_ctx;
   (*out)->device = device;
   return MMH_OK;
}

extern "C" void mmh_destroy(mmh_ctx *c) { delete c; }

extern "C" int mmh_set_timing(mmh_ctx *c, int) { return c ? MMH_OK : MMH_E_ARG; }     // (nothing to time here)

extern "C" int mmh_rom_upload(mmh_ctx *c, const void *host, uint64_t nbytes)
{
   c->rom.assign((const uint8_t *)host, (const uint8_t *)host + nbytes);
   return MMH_OK;
}

extern "C" int mmh_rom_load_file(mmh_ctx *c, const char *path, uint64_t file_offset, uint64_t nbytes, int)
{
   FILE *f = std::fopen(path, "rb");
   if (!f) {
      mmh_set_error("double: cannot open %s", path);
      return MMH_E_ARG;
   }
   c->rom.assign(nbytes, 0);
   bool ok = std::fseek(f, (long)file_offset, SEEK_SET) == 0 && std::fread(c->rom.data(), 1, nbytes, f) == nbytes;
   std::fclose(f);
   if (!ok) {
      mmh_set_error("double: short read of %s", path);
      return MMH_E_ARG;
   }
   return MMH_OK;
}

// the watched load: 4 MiB pieces, the abort word read before each, bytes_done raised behind each;
// MMOORE_DOUBLE_PIECE_US slows every piece down (a stand-in for a file that takes a while to cross PCIe)
extern "C" int mmh_rom_load_file_watched(mmh_ctx *c, const char *path, uint64_t file_offset, uint64_t nbytes, int,
                                         const volatile int32_t *abort_word, volatile uint64_t *bytes_done)
{
   FILE *f = std::fopen(path, "rb");
   if (!f) {
      mmh_set_error("double: cannot open %s", path);
      return MMH_E_ARG;
   }
   if (bytes_done) {
      __atomic_store_n(bytes_done, (uint64_t)0, __ATOMIC_RELAXED);
   }
   const char *slow = getenv("MMOORE_DOUBLE_PIECE_US");
   const long piece_us = slow && *slow ? atol(slow) : 0;
   c->rom.assign(nbytes, 0);
   int rc = std::fseek(f, (long)file_offset, SEEK_SET) == 0 ? MMH_OK : MMH_E_ARG;
   for (uint64_t at = 0; rc == MMH_OK && at < nbytes; at += 4u << 20) {
      if (abort_word && __atomic_load_n(abort_word, __ATOMIC_RELAXED) != 0) {
         mmh_set_error("double: aborted by the caller");
         rc = MMH_E_ABORTED;
         break;
      }
      const uint64_t len = std::min<uint64_t>(4u << 20, nbytes - at);
      if (std::fread(c->rom.data() + at, 1, len, f) != len) {
         mmh_set_error("double: short read of %s", path);
         rc = MMH_E_ARG;
         break;
      }
      if (piece_us) {
         struct timespec ts = {0, piece_us * 1000};
         nanosleep(&ts, nullptr);
      }
      if (bytes_done) {
         __atomic_fetch_add(bytes_done, len, __ATOMIC_RELAXED);
      }
   }
   std::fclose(f);
   return rc;
}

extern "C" int mmh_rom_gather(mmh_ctx *c, const uint64_t *offs, uint64_t n, uint32_t each, void *host_out)
{
   uint8_t *out = (uint8_t *)host_out;
   for (uint64_t k = 0; k < n; k++) {
      for (uint32_t b = 0; b < each; b++) {
         const uint64_t o = offs[k] + b;
         out[k * each + b] = o < c->rom.size() ? c->rom[o] : 0;
      }
   }
   return MMH_OK;
}

namespace {
struct View {
   const uint8_t *p;
   uint32_t S;
   bool be;
   int at(uint64_t j) const
   {
      if (S == 1) {
         return p[j];
      }
      const int lo = p[2 * j], hi = p[2 * j + 1];
      return be ? (lo << 8 | hi) : (hi << 8 | lo);
   }
};

int skip_of(const mmh_plan_desc &pl, int d)
{
   int s = pl.default_skip;
   for (uint32_t k = 0; k < pl.n_skip; k++) {
      if (pl.skip_diff[k] == d) {
         s = pl.skip_val[k];
      }
   }
   return s;
}

// one chain over `count` elements; emits element indices
template <class Emit> void chain(const mmh_plan_desc &pl, const View &v, uint64_t count, Emit emit)
{
   if (count < pl.L) {
      return;
   }
   const uint64_t nv = count - pl.L + 1;
   for (uint64_t h = 0; h < nv;) {
      uint64_t jump = pl.match_jump;
      bool matched = true;
      for (int i = (int)pl.L - 1; i >= 0; i--) {
         const int d = v.at(h + i) - v.at(h + i + pl.bridge[i]);
         if (((uint32_t)(d ^ pl.expected[i]) & pl.cmp_mask[i]) != 0) {
            const int s = std::max(skip_of(pl, d), 1);
            jump = (uint64_t)std::min<int>(s, pl.wst[i]);
            matched = false;
            break;
         }
      }
      if (matched) {
         emit(h);
      }
      h += jump;
   }
}

int scan_one(const mmh_ctx *c, const mmh_plan_desc *pl, uint64_t block, int big_endian, uint64_t base, std::vector<uint64_t> *out)
{
   const uint32_t S = pl->elem_bytes;
   const uint64_t n = c->rom.size();
   if (block == 0) {
      chain(*pl, View{c->rom.data(), S, false}, n / S, [&](uint64_t h) { out->push_back(h); });
      return MMH_OK;
   }
   for (uint64_t off = 0; off < n; off += block) {
      const uint64_t size = std::min<uint64_t>(block + (uint64_t)(pl->L - 1) * S, n - off);
      for (uint32_t p = 0; p < S; p++) {
         uint64_t count = size / S;
         if (p + count * S > size) {
            count--;                                   // search_engine.cpp:137-141
         }
         chain(*pl, View{c->rom.data() + off + p, S, S == 2 && big_endian != 0}, count,
               [&](uint64_t h) { out->push_back(base + off + h * S + p); });
      }
   }
   std::sort(out->begin(), out->end());
   return MMH_OK;
}

int deliver(const std::vector<uint64_t> &found, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   *out_count = found.size();
   if (found.size() > cap) {
      mmh_set_error("double: %zu matches do not fit", found.size());
      return MMH_E_CAPACITY;
   }
   std::copy(found.begin(), found.end(), out);
   return MMH_OK;
}
} // namespace

extern "C" int mmh_scan(mmh_ctx *c, const mmh_plan_desc *pl, uint64_t block, int big_endian, uint64_t base, uint64_t *out,
                        uint64_t cap, uint64_t *out_count)
{
   std::vector<uint64_t> found;
   scan_one(c, pl, block, big_endian, base, &found);
   return deliver(found, out, cap, out_count);
}

extern "C" int mmh_comm_init_all(mmh_ctx *const *ctxs, int n)
{
   for (int i = 0; i < n; i++) {
      ctxs[i]->rank = i;
      ctxs[i]->nranks = n;
   }
   return MMH_OK;
}

extern "C" int mmh_comm_info(mmh_ctx *c, int *rank, int *nranks)
{
   *rank = c->rank;
   *nranks = c->nranks;
   return MMH_OK;
}

extern "C" int mmh_scan_multi(mmh_ctx *const *ctxs, int n, const mmh_plan_desc *pl, uint64_t block, int big_endian,
                              const uint64_t *bases, uint64_t *out, uint64_t cap, uint64_t *out_count)
{
   std::vector<uint64_t> all;
   for (int i = 0; i < n; i++) {
      std::vector<uint64_t> found;
      scan_one(ctxs[i], pl, block, big_endian, bases[i], &found);
      all.insert(all.end(), found.begin(), found.end());     // partitions in rank order: already ascending
   }
   return deliver(all, out, cap, out_count);
}
