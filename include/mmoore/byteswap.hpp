// byteswap.hpp -- endianness helpers of the mmoore API (MI355X build).
//
// Mirrors the names of the reference's include/mmoore/byteswap.hpp:9-79.  On the GPU path
// big-endian 16-bit ROMs are swapped while the elements are assembled in the kernels, so
// the engine itself never calls adjust_endianness; the helpers remain for callers (and for
// the host-side preview decoding).
#ifndef MMOORE_AMD_BYTESWAP_HPP
#define MMOORE_AMD_BYTESWAP_HPP

#include <cstddef>
#include <cstdint>

namespace mmoore {

enum class Endianness { Little, Big };

inline Endianness get_system_endianness()
{
   const uint16_t probe = 0x0102;
   return *reinterpret_cast<const uint8_t *>(&probe) == 0x02 ? Endianness::Little : Endianness::Big;
}

// generic case: single bytes (and anything we do not know how to swap) stay as they are
template <typename T>
constexpr T swap_always(T value)
{
   return value;
}

template <>
constexpr uint16_t swap_always<uint16_t>(uint16_t value)
{
   return static_cast<uint16_t>((value >> 8) | (value << 8));
}

template <>
constexpr uint32_t swap_always<uint32_t>(uint32_t value)
{
   return (value >> 24) | ((value >> 8) & 0x0000FF00u) | ((value << 8) & 0x00FF0000u) | (value << 24);
}

template <typename T>
T swap_on_little_endian(T value)
{
   return get_system_endianness() == Endianness::Little ? swap_always<T>(value) : value;
}

template <typename T>
T swap_on_big_endian(T value)
{
   return get_system_endianness() == Endianness::Big ? swap_always<T>(value) : value;
}

// make `count` elements at `data` read correctly on this host when the file stores them
// in `stored_as` order (in place)
template <typename T>
void adjust_endianness(T *data, size_t count, Endianness stored_as)
{
   if (stored_as == get_system_endianness()) {
      return;
   }
   for (size_t i = 0; i < count; i++) {
      data[i] = swap_always<T>(data[i]);
   }
}

} // namespace mmoore

#endif
