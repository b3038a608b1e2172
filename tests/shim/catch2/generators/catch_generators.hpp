// SPDX-License-Identifier: GPL-3.0-or-later
// GENERATE lives in the Catch2 stand-in's main header (tests/shim/catch2/catch_test_macros.hpp)
#include <catch2/catch_test_macros.hpp>
