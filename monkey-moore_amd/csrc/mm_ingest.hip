// SPDX-License-Identifier: GPL-3.0-or-later
// mm_ingest.hip -- getting a file into HBM, and match bytes back out of it.
//
// The reference's workers each read their own blocks with an ifstream and scan them in
// place (src/core/search_engine.cpp:120-145); with the scan on the GPU the file has to cross
// PCIe instead.  One reader cannot feed a PCIe 5 x16 link from the page cache (a single
// pread() stream copies at a few GB/s), so mmh_rom_load_file runs several readers, each
// filling pinned 4 MiB pieces and queueing the host-to-device copy of a piece right behind
// its read: reads, copies of different readers and the DMA engines all overlap.
//
// mmh_rom_gather is the way back: the elements under every match (for the equivalency maps,
// monkey_moore.cpp:374-393) are packed by one small kernel and copied out in one piece, so
// that the host never needs its own copy of the file.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mm_context.h"
#include "mm_internal.h"
#include "mm_kernels.h"

namespace {

bool ingest_resources(mmh_ctx *c, int threads, std::string *err)
{
   MmIngest &in = c->ingest;
   auto fail = [&](const char *what, hipError_t e) {
      *err = std::string(what) + ": " + hipGetErrorString(e);
      return false;
   };
   while ((int)in.streams.size() < threads) {
      hipStream_t s = nullptr;
      hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      if (e != hipSuccess) {
         return fail("hipStreamCreate", e);
      }
      in.streams.push_back(s);
   }
   while ((int)in.staging.size() < 2 * threads) {
      void *p = nullptr;
      hipError_t e = hipHostMalloc(&p, MmIngest::kPiece, hipHostMallocDefault);
      if (e != hipSuccess) {
         return fail("hipHostMalloc (ingest staging)", e);
      }
      in.staging.push_back(p);
      hipEvent_t ev = nullptr;
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) {
         return fail("hipEventCreate", e);
      }
      in.events.push_back(ev);
   }
   return true;
}

} // namespace

// One load: what its readers share.  It lives on the heap and the readers hold it by shared_ptr, because an aborted
// load returns to its caller while readers may still sit inside a blocking call (measured on the MI355X box: a
// reader's first hipMemcpyAsync of a load occasionally blocks for 9-33 ms); such stragglers finish against this
// block -- never against the caller's abort word or byte counter, which may be gone by then.
struct LoadJob {
   mmh_ctx *c = nullptr;
   uint8_t *rom = nullptr;         // the ROM's base when the load began: a straggler of an aborted load never looks at c->rom,
                                   // which a later upload may have pointed somewhere else by the time it queues its copy
   std::string path;
   uint64_t file_offset = 0, nbytes = 0, npieces = 0;
   std::atomic<uint64_t> next{0}, bytes_done{0};
   std::atomic<int> running{0};
   std::atomic<bool> failed{false}, stop{false};
   std::mutex err_lock;
   std::string err;

   void give_up(const std::string &why)
   {
      std::lock_guard<std::mutex> hold(err_lock);
      if (!failed.exchange(true)) {
         err = why;
      }
   }
};

static void read_pieces(std::shared_ptr<LoadJob> job, int t)
{
   LoadJob &j = *job;
   MmIngest &in = j.c->ingest;
   struct Leave {
      LoadJob &j;
      ~Leave() { j.running--; }
   } leave{j};
   if (hipSetDevice(j.c->device) != hipSuccess) {
      return j.give_up("hipSetDevice failed in a reader thread");
   }
   const int fd = open(j.path.c_str(), O_RDONLY);
   if (fd < 0) {
      return j.give_up(std::string("cannot open ") + j.path + ": " + strerror(errno));
   }
   const uint64_t piece = MmIngest::kPiece;
   bool used[2] = {false, false};
   for (unsigned turn = 0; !j.failed && !j.stop; turn++) {
      const uint64_t k = j.next.fetch_add(1);
      if (k >= j.npieces) {
         break;
      }
      const int slot = 2 * t + (int)(turn & 1);
      if (used[turn & 1] && hipEventSynchronize(in.events[slot]) != hipSuccess) {
         j.give_up("hipEventSynchronize failed");
         break;
      }
      const uint64_t at = k * piece;
      const uint64_t len = std::min(piece, j.nbytes - at);
      uint8_t *dst = static_cast<uint8_t *>(in.staging[slot]);
      uint64_t got = 0;
      while (got < len && !j.stop) {
         // (1 MiB per call: a reader is never more than ~0.3 ms of reading away from seeing the stop flag)
         const ssize_t r = pread(fd, dst + got, std::min<uint64_t>(len - got, 1u << 20), (off_t)(j.file_offset + at + got));
         if (r < 0 && errno == EINTR) {
            continue;
         }
         if (r <= 0) {
            j.give_up(r == 0 ? std::string("short read from ") + j.path : std::string("read error on ") + j.path + ": " + strerror(errno));
            break;
         }
         got += (uint64_t)r;
      }
      if (got < len) {
         break;
      }
      if (hipMemcpyAsync(j.rom + at, dst, len, hipMemcpyHostToDevice, in.streams[t]) != hipSuccess ||
          hipEventRecord(in.events[slot], in.streams[t]) != hipSuccess) {
         j.give_up("host-to-device copy failed");
         break;
      }
      used[turn & 1] = true;
      j.bytes_done += len;
   }
   // A stopped load does not wait for the copies it has queued (up to two pieces per reader): the caller wants its
   // thread back now.  They are drained before the staging buffers or the ROM are touched again (mm_ingest_drain).
   if (!j.stop && hipStreamSynchronize(in.streams[t]) != hipSuccess) {
      j.give_up("hipStreamSynchronize failed in a reader thread");
   }
   close(fd);
}

// what an aborted load left behind: readers still inside a blocking call, copies in flight -- wait for both before the
// ROM or the staging buffers are used again.  With an abort word the wait itself can be called off (MMH_E_ABORTED,
// everything still undrained): the next load of a search that was aborted a moment ago is aborted just as promptly.
static int drain_watched(mmh_ctx *c, const volatile int32_t *abort_word)
{
   if (!c || !c->ingest.undrained) {
      return MMH_OK;
   }
   MmIngest &in = c->ingest;
   auto called_off = [&]() {
      if (abort_word && __atomic_load_n(abort_word, __ATOMIC_RELAXED) != 0) {
         mmh_set_error("mmh_rom_load_file: aborted by the caller");
         return true;
      }
      return false;
   };
   while (abort_word && in.straggling && in.straggling->load() > 0) {
      if (called_off()) {
         return MMH_E_ABORTED;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(100));
   }
   for (auto &th : in.stragglers) {
      th.join();
   }
   in.stragglers.clear();
   in.straggling.reset();
   if (hipSetDevice(c->device) != hipSuccess) {
      return MMH_E_DEVICE;
   }
   for (hipStream_t s : in.streams) {
      hipError_t e;
      while (abort_word && (e = hipStreamQuery(s)) == hipErrorNotReady) {
         if (called_off()) {
            return MMH_E_ABORTED;
         }
         std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      if (hipStreamSynchronize(s) != hipSuccess) {
         mmh_set_error("draining an aborted file load failed");
         return MMH_E_DEVICE;
      }
   }
   in.undrained = false;
   return MMH_OK;
}

int mm_ingest_drain(mmh_ctx *c) { return drain_watched(c, nullptr); }

extern "C" int mmh_rom_load_file(mmh_ctx *c, const char *path, uint64_t file_offset, uint64_t nbytes, int threads)
{
   return mmh_rom_load_file_watched(c, path, file_offset, nbytes, threads, nullptr, nullptr);
}

extern "C" int mmh_rom_load_file_watched(mmh_ctx *c, const char *path, uint64_t file_offset, uint64_t nbytes, int threads,
                                         const volatile int32_t *abort_word, volatile uint64_t *bytes_done)
{
   if (!c || !path || threads < 0) {
      mmh_set_error("mmh_rom_load_file: bad argument");
      return MMH_E_ARG;
   }
   if (bytes_done) {
      __atomic_store_n(bytes_done, (uint64_t)0, __ATOMIC_RELAXED);
   }
   const auto t0 = std::chrono::steady_clock::now();
   int rc = drain_watched(c, abort_word);
   if (rc != MMH_OK) {
      return rc;
   }
   rc = mmh_rom_alloc(c, nbytes);
   if (rc != MMH_OK) {
      return rc;
   }
   // the padding memset of mmh_rom_alloc runs on the scan stream; the copies do not
   if (hipStreamSynchronize(c->stream) != hipSuccess) {
      mmh_set_error("mmh_rom_load_file: stream synchronisation failed");
      return MMH_E_DEVICE;
   }
   const uint64_t piece = MmIngest::kPiece;
   const uint64_t npieces = (nbytes + piece - 1) / piece;
   if (threads == 0) {
      const unsigned hw = std::thread::hardware_concurrency();
      threads = (int)std::min<unsigned>(16, hw ? hw : 1);
   }
   threads = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, npieces));
   std::string err;
   if (!ingest_resources(c, threads, &err)) {
      mmh_set_error("%s", err.c_str());
      return MMH_E_DEVICE;
   }
   const int probe = open(path, O_RDONLY);
   if (probe < 0) {
      mmh_set_error("mmh_rom_load_file: cannot open %s: %s", path, strerror(errno));
      return MMH_E_ARG;
   }
   close(probe);

   auto job = std::make_shared<LoadJob>();
   job->c = c;
   job->rom = c->rom;
   job->path = path;
   job->file_offset = file_offset;
   job->nbytes = nbytes;
   job->npieces = npieces;
   job->running = threads;
   MmIngest &in = c->ingest;
   // The calling thread reads pieces too unless it has an abort word to watch: then it only supervises -- it carries
   // the caller's word in and the byte count out every 0.1 ms and leaves the moment the word is up.
   const bool supervise = abort_word != nullptr;
   std::vector<std::thread> pool;
   for (int t = supervise ? 0 : 1; t < threads; t++) {
      pool.emplace_back(read_pieces, job, t);
   }
   bool aborted = false;
   if (!supervise) {
      read_pieces(job, 0);
      if (bytes_done) {
         __atomic_store_n(bytes_done, job->bytes_done.load(), __ATOMIC_RELAXED);
      }
   }
   else {
      const auto poll_start = std::chrono::steady_clock::now();
      for (;;) {
         if (bytes_done) {
            __atomic_store_n(bytes_done, job->bytes_done.load(), __ATOMIC_RELAXED);
         }
         if (job->running == 0) {
            break;
         }
         if (__atomic_load_n(abort_word, __ATOMIC_RELAXED) != 0) {
            job->stop = true;
            aborted = true;
            break;
         }
         // (small files are in HBM within microseconds: spin first, sleep once the load turns out to take a while)
         if (std::chrono::steady_clock::now() - poll_start < std::chrono::microseconds(300)) {
            std::this_thread::yield();
         }
         else {
            std::this_thread::sleep_for(std::chrono::microseconds(100));
         }
      }
   }
   if (aborted) {
      for (auto &th : pool) {
         in.stragglers.push_back(std::move(th));
      }
      in.straggling = std::shared_ptr<std::atomic<int>>(job, &job->running);
      in.undrained = true;
   }
   else {
      for (auto &th : pool) {
         th.join();
      }
      // (a reader may have queued its last piece between the supervisor's last look at the count and at `running`)
      if (bytes_done) {
         __atomic_store_n(bytes_done, job->bytes_done.load(), __ATOMIC_RELAXED);
      }
   }
   in.last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   in.last_bytes = nbytes;
   in.last_threads = threads;
   if (aborted) {
      static const bool trace = mm_trace("ingest");
      if (trace) {
         fprintf(stderr, "mmh_rom_load_file: aborted after %.2f ms, %d of %d readers still busy\n", in.last_seconds * 1e3,
                 job->running.load(), threads);
      }
      mmh_set_error("mmh_rom_load_file: aborted by the caller");
      return MMH_E_ABORTED;
   }
   if (job->failed) {
      mmh_set_error("mmh_rom_load_file: %s", job->err.c_str());
      return job->err.find("short read") != std::string::npos || job->err.find("cannot open") != std::string::npos ? MMH_E_ARG : MMH_E_DEVICE;
   }
   return MMH_OK;
}

extern "C" int mmh_last_load_stats(mmh_ctx *c, double *seconds, uint64_t *bytes, int *threads)
{
   if (!c || !seconds || !bytes || !threads) {
      mmh_set_error("mmh_last_load_stats: bad argument");
      return MMH_E_ARG;
   }
   *seconds = c->ingest.last_seconds;
   *bytes = c->ingest.last_bytes;
   *threads = c->ingest.last_threads;
   return MMH_OK;
}

extern "C" int mmh_rom_gather(mmh_ctx *c, const uint64_t *offsets, uint64_t n, uint32_t bytes_each, void *host_out)
{
   if (!c || (n && (!offsets || !host_out)) || bytes_each == 0) {
      mmh_set_error("mmh_rom_gather: bad argument");
      return MMH_E_ARG;
   }
   if (!c->rom) {
      mmh_set_error("mmh_rom_gather: no ROM attached");
      return MMH_E_STATE;
   }
   if (n == 0) {
      return MMH_OK;
   }
   int rc = mmh_workspace(c);
   if (rc != MMH_OK) {
      return rc;
   }
   // scan workspace doubles as staging: d_out takes the offsets, d_cand the packed bytes
   const uint64_t batch = std::min<uint64_t>(c->ws[0].out_cap, c->ws[0].cand_cap * sizeof(uint64_t) / bytes_each);
   if (batch == 0) {
      mmh_set_error("mmh_rom_gather: %u bytes per offset is too much", bytes_each);
      return MMH_E_ARG;
   }
   uint8_t *out = static_cast<uint8_t *>(host_out);
   for (uint64_t first = 0; first < n; first += batch) {
      const uint64_t m = std::min(batch, n - first);
      if (hipMemcpyAsync(c->ws[0].d_out, offsets + first, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
         mmh_set_error("mmh_rom_gather: offset upload failed");
         return MMH_E_DEVICE;
      }
      mm::launch_gather(c->stream, c->rom, c->rom_bytes, c->ws[0].d_out, m, bytes_each, reinterpret_cast<uint8_t *>(c->ws[0].d_cand));
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(out + first * bytes_each, c->ws[0].d_cand, m * bytes_each, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
          hipStreamSynchronize(c->stream) != hipSuccess) {
         mmh_set_error("mmh_rom_gather: device gather failed");
         return MMH_E_DEVICE;
      }
   }
   return MMH_OK;
}
