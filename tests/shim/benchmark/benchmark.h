// SPDX-License-Identifier: GPL-3.0-or-later
// Minimal stand-in for the part of google-benchmark the reference's benchmarks/bench_search.cpp
// uses: State (range, range-for, iterations, SetBytesProcessed), DoNotOptimize,
// BENCHMARK_TEMPLATE(...)->Name()->RangeMultiplier()->Range(), BENCHMARK_MAIN.  The library is
// not in the image; this lets the reference's benchmark compile unmodified against
// include/mmoore + libmonkey-core.so and print the same quantities.  Test infrastructure only.
#ifndef MM_SHIM_BENCHMARK_H
#define MM_SHIM_BENCHMARK_H
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace benchmark {
class State {
public:
   State(int64_t range0, int64_t iters) : range0_(range0), todo_(iters), iters_(iters) {}
   int64_t range(int) const { return range0_; }
   int64_t iterations() const { return iters_; }
   void SetBytesProcessed(int64_t b) { bytes_ = b; }
   int64_t bytes_processed() const { return bytes_; }
   double seconds() const { return seconds_; }
   struct Iterator {
      State *s;
      bool operator!=(const Iterator &) {
         if (s->todo_ > 0) return true;
         s->seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - s->start_).count();
         return false;
      }
      void operator++() { s->todo_--; }
      int operator*() const { return 0; }
   };
   Iterator begin() { start_ = std::chrono::steady_clock::now(); return {this}; }
   Iterator end() { return {this}; }
private:
   int64_t range0_, todo_, iters_, bytes_ = 0;
   double seconds_ = 0;
   std::chrono::steady_clock::time_point start_;
};
template <class T> inline void DoNotOptimize(T &value) { asm volatile("" : "+m"(value) : : "memory"); }

namespace internal {
struct Benchmark {
   std::string name;
   void (*fn)(State &);
   int64_t mult = 8, lo = 0, hi = 0;
   Benchmark *Name(const std::string &n) { name = n; return this; }
   Benchmark *RangeMultiplier(int m) { mult = m; return this; }
   Benchmark *Range(int64_t a, int64_t b) { lo = a; hi = b; return this; }
};
inline std::vector<Benchmark *> &all() { static std::vector<Benchmark *> v; return v; }
inline Benchmark *Register(const char *name, void (*fn)(State &)) { all().push_back(new Benchmark{name, fn}); return all().back(); }
inline int RunAll() {
   std::printf("%-52s %14s %12s %14s\n", "Benchmark", "Time", "Iterations", "bytes_per_second");
   for (Benchmark *b : all()) {
      std::vector<int64_t> sizes;                     // lo, lo*mult, ... and the upper end itself, as google-benchmark's Range
      for (int64_t r = b->lo; r < b->hi && b->mult > 1; r *= b->mult) sizes.push_back(r);
      sizes.push_back(b->hi);
      for (int64_t r : sizes) {
         int64_t iters = 1;
         for (;;) {                               // grow the iteration count until a run lasts long enough to time
            State st(r, iters);
            b->fn(st);
            if (st.seconds() >= 0.25 || iters >= (1 << 20)) {
               std::printf("%-52s %11.0f ns %12lld %11.3f G/s\n", (b->name + "/" + std::to_string(r)).c_str(), st.seconds() / iters * 1e9,
                           (long long)iters, st.bytes_processed() / st.seconds() / 1e9);
               break;
            }
            iters = st.seconds() > 0.005 ? (int64_t)(iters * 0.3 / st.seconds()) + 1 : iters * 10;
         }
      }
   }
   return 0;
}
} // namespace internal
} // namespace benchmark

#define MM_SHIM_BM_CAT2(a, b) a##b
#define MM_SHIM_BM_CAT(a, b) MM_SHIM_BM_CAT2(a, b)
#define BENCHMARK_TEMPLATE(fn, ...) \
   static ::benchmark::internal::Benchmark *MM_SHIM_BM_CAT(mm_shim_bm_, __LINE__) = ::benchmark::internal::Register(#fn "<" #__VA_ARGS__ ">", fn<__VA_ARGS__>)
#define BENCHMARK_MAIN() int main() { return ::benchmark::internal::RunAll(); }
#endif
