// SPDX-License-Identifier: GPL-3.0-or-later
// Minimal stand-in for the part of Catch2 v3 the reference's tests use: TEST_CASE,
// nested SECTION, GENERATE, REQUIRE / CHECK / FAIL, REQUIRE_THAT(v, Equals(vec)),
// REQUIRE_THROWS_AS, INFO / CAPTURE.  Catch2 is not in the image; this lets the reference's own
// test sources compile -- from where they lie, unmodified -- against include/mmoore +
// libmonkey-core.so (the MI355X facade).  Test infrastructure only.
//
// Semantics kept: a TEST_CASE body is re-run until every leaf SECTION path has executed once,
// one leaf per run; a GENERATE makes the section (or test case) it sits in run once per value,
// several GENERATEs give the cross product.
#ifndef MM_SHIM_CATCH_TEST_MACROS_HPP
#define MM_SHIM_CATCH_TEST_MACROS_HPP
#include <cstdio>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace catch_shim {
struct TestFailure : std::runtime_error { using std::runtime_error::runtime_error; };
struct Case { const char *name; void (*fn)(); };
inline std::vector<Case> &cases() { static std::vector<Case> v; return v; }
struct Registrar { Registrar(const char *n, void (*f)()) { cases().push_back({n, f}); } };

struct Gen { size_t index = 0, count = 1; };
struct Frame { std::string key; bool entered_child = false, pending = false; std::vector<std::string> gens; };
struct State {
   std::set<std::string> done;                 // section paths that have run to completion
   std::map<std::string, Gen> gens;            // generator position by "section path#line"
   std::vector<Frame> stack;                   // open sections of the run under way (stack[0] = the test case)
   int checks = 0, failures = 0;
};
inline State &state() { static State s; return s; }

// a section (or the test case itself) is finished: advance its generators like an odometer; returns true when it must run again
inline bool advance_generators(Frame &f) {
   State &s = state();
   for (size_t i = f.gens.size(); i-- > 0;) {
      Gen &g = s.gens[f.gens[i]];
      if (g.index + 1 < g.count) {
         g.index++;
         for (size_t k = i + 1; k < f.gens.size(); k++) s.gens[f.gens[k]].index = 0;
         for (auto it = s.done.begin(); it != s.done.end();)          // everything below runs again with the new value
            it = it->compare(0, f.key.size() + 1, f.key + "/") == 0 ? s.done.erase(it) : ++it;
         return true;
      }
   }
   for (auto &k : f.gens) s.gens[k].index = 0;
   return false;
}
struct Section {
   bool entered = false;
   Section(const char *name, int line) {
      State &s = state();
      Frame &parent = s.stack.back();
      const std::string key = parent.key + "/" + name + "@" + std::to_string(line);
      if (s.done.count(key)) return;
      if (parent.entered_child) { parent.pending = true; return; }      // one sibling per run; this one waits for the next run
      parent.entered_child = true;
      entered = true;
      s.stack.push_back(Frame{key});
   }
   ~Section() {
      if (!entered) return;
      State &s = state();
      Frame f = s.stack.back();
      s.stack.pop_back();
      if (std::uncaught_exceptions()) { s.done.insert(f.key); return; }   // a failed REQUIRE ends this path
      if (f.pending || advance_generators(f)) s.stack.back().pending = true;
      else s.done.insert(f.key);
   }
   explicit operator bool() const { return entered; }
};
template <class T, class... Rest> T generate(int line, T first, Rest... rest) {
   State &s = state();
   Frame &f = s.stack.back();
   const std::string key = f.key + "#" + std::to_string(line);
   bool known = false;
   for (auto &k : f.gens) known = known || k == key;
   if (!known) f.gens.push_back(key);
   const std::vector<T> values{first, static_cast<T>(rest)...};
   Gen &g = s.gens[key];
   g.count = values.size();
   return values[g.index];
}
inline void report(bool ok, const char *expr, const char *file, int line, bool fatal) {
   State &s = state();
   s.checks++;
   if (ok) return;
   s.failures++;
   std::printf("FAILED %s:%d: %s   [%s]\n", file, line, expr, s.stack.empty() ? "" : s.stack.back().key.c_str());
   if (fatal) throw TestFailure(expr);
}
inline int run_all() {
   State &s = state();
   for (auto &c : cases()) {
      s.done.clear(); s.gens.clear();
      for (int runs = 0; runs < 100000; runs++) {
         s.stack.assign(1, Frame{std::string(c.name)});
         try { c.fn(); }
         catch (const TestFailure &) {}
         catch (const std::exception &e) { s.failures++; std::printf("FAILED %s: unexpected exception: %s\n", c.name, e.what()); }
         Frame root = s.stack.front();
         if (!root.pending && !advance_generators(root)) break;
      }
   }
   std::printf("%zu test cases, %d assertions, %d failures\n", cases().size(), s.checks, s.failures);
   return s.failures ? 1 : 0;
}
} // namespace catch_shim

#define MM_SHIM_CAT2(a, b) a##b
#define MM_SHIM_CAT(a, b) MM_SHIM_CAT2(a, b)
#define TEST_CASE(name, ...)                                                                                   \
   static void MM_SHIM_CAT(mm_shim_test_, __LINE__)();                                                         \
   static catch_shim::Registrar MM_SHIM_CAT(mm_shim_reg_, __LINE__)(name, MM_SHIM_CAT(mm_shim_test_, __LINE__)); \
   static void MM_SHIM_CAT(mm_shim_test_, __LINE__)()
#define SECTION(name) if (catch_shim::Section MM_SHIM_CAT(mm_shim_sec_, __LINE__){name, __LINE__})
#define GENERATE(...) catch_shim::generate(__LINE__, __VA_ARGS__)
#define REQUIRE(...) catch_shim::report(static_cast<bool>(__VA_ARGS__), #__VA_ARGS__, __FILE__, __LINE__, true)
#define CHECK(...) catch_shim::report(static_cast<bool>(__VA_ARGS__), #__VA_ARGS__, __FILE__, __LINE__, false)
#define REQUIRE_THAT(value, matcher) catch_shim::report((matcher).match(value), #value " " #matcher, __FILE__, __LINE__, true)
#define FAIL(msg) catch_shim::report(false, "FAIL: " #msg, __FILE__, __LINE__, true)
#define INFO(...) do { } while (0)
#define CAPTURE(...) do { } while (0)
#define REQUIRE_THROWS_AS(expr, type)                                                          \
   do {                                                                                        \
      bool mm_shim_threw = false;                                                              \
      try { (void)(expr); } catch (const type &) { mm_shim_threw = true; } catch (...) { }     \
      catch_shim::report(mm_shim_threw, #expr " throws " #type, __FILE__, __LINE__, true);     \
   } while (0)
#endif
