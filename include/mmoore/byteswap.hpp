// SPDX-License-Identifier: GPL-3.0-or-later
// byteswap.hpp -- endianness helpers of the mmoore API (MI355X build).
//
// Same names and call signatures as the helpers the reference exposes in its header of the
// same name (Endianness, get_system_endianness, swap_always, swap_on_little_endian,
// swap_on_big_endian, adjust_endianness), so harness and GUI code keeps compiling.
//
// On the GPU path nothing calls adjust_endianness any more: a big-endian 16-bit ROM is
// byte-swapped while the elements are assembled in registers (v_perm_b32 in
// mm_filter_u16, two byte reads in the tile kernels).  The helpers remain for callers and
// for the host-side preview decoding of the facade.
#ifndef MMOORE_AMD_BYTESWAP_HPP
#define MMOORE_AMD_BYTESWAP_HPP

#include <cstddef>
#include <cstdint>
#include <type_traits>

namespace mmoore {

// byte order of multi-byte elements in a file
enum class Endianness { Little, Big };

// byte order of the machine this code runs on
inline Endianness get_system_endianness()
{
   const uint16_t probe = 0x0102;
   const auto *first_byte = reinterpret_cast<const uint8_t *>(&probe);
   return *first_byte == 0x02 ? Endianness::Little : Endianness::Big;
}

// Reverses the bytes of a 16- or 32-bit unsigned value; every other type (single bytes
// included) is returned unchanged, like the reference's generic template.
template <typename T>
constexpr T swap_always(T value)
{
   if constexpr (std::is_same_v<T, uint16_t>) {
      return static_cast<uint16_t>(static_cast<uint16_t>(value >> 8) | static_cast<uint16_t>(value << 8));
   }
   else if constexpr (std::is_same_v<T, uint32_t>) {
      const uint32_t halves = (value >> 16) | (value << 16);                       // swap the 16-bit halves ...
      return ((halves & 0xFF00FF00u) >> 8) | ((halves & 0x00FF00FFu) << 8);       // ... then the bytes in each
   }
   else {
      return value;
   }
}

// reverse the bytes only on a little-endian / only on a big-endian host
template <typename T>
T swap_on_little_endian(T value)
{
   const bool host_is_little = get_system_endianness() == Endianness::Little;
   return host_is_little ? swap_always<T>(value) : value;
}

template <typename T>
T swap_on_big_endian(T value)
{
   const bool host_is_big = get_system_endianness() == Endianness::Big;
   return host_is_big ? swap_always<T>(value) : value;
}

// In place: make `count` elements that a file stores in `stored_as` order read correctly
// on this host.  A no-op when the orders already agree.
template <typename T>
void adjust_endianness(T *data, size_t count, Endianness stored_as)
{
   if (stored_as != get_system_endianness()) {
      for (T *p = data, *end = data + count; p != end; ++p) {
         *p = swap_always<T>(*p);
      }
   }
}

} // namespace mmoore

#endif
