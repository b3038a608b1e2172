// SPDX-License-Identifier: GPL-3.0-or-later
// Catch::Matchers::Equals(std::vector) of the Catch2 stand-in (tests/shim/catch2/catch_test_macros.hpp)
#ifndef MM_SHIM_CATCH_MATCHERS_VECTOR_HPP
#define MM_SHIM_CATCH_MATCHERS_VECTOR_HPP
#include <vector>
namespace Catch { namespace Matchers {
template <class T> struct VectorEquals {
   const std::vector<T> &expected;
   bool match(const std::vector<T> &got) const { return got == expected; }   // element operator== by ADL (tests/common.hpp)
};
template <class T> VectorEquals<T> Equals(const std::vector<T> &expected) { return {expected}; }
} }
#endif
