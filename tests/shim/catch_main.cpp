// SPDX-License-Identifier: GPL-3.0-or-later
// main() of the Catch2 stand-in (the reference links Catch2::Catch2WithMain, tests/CMakeLists.txt)
#include <catch2/catch_test_macros.hpp>
int main() { return catch_shim::run_all(); }
