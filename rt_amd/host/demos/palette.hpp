// palette.hpp -- the five colour families the demo scenes pick sphere colours from.
// Colour values are the reference's scene content (demos/palette.hpp:14-53), kept so the rebuilt demos
// look like the originals; selection goes family first, then colour, as RandomFromAll does
// (demos/palette.hpp:62-69).
#pragma once

#include "ray_tracing/vec3.hpp"
#include "scene_random.hpp"

#include <cstdint>

namespace demo
{
class Palette
{
public:
  static constexpr int kFamilies = 5;

  static ray_tracing::vec3 RandomFromAll(SceneRandom & rnd)
  {
    int const family = rnd.Int(0, kFamilies - 1);
    int const first = kFirst[family];
    int const count = kFirst[family + 1] - first;
    return Convert(kColors[first + rnd.Int(0, count - 1)]);
  }

  // 0xRRGGBB -> components / 255 (demos/palette.hpp:72-77).
  static ray_tracing::vec3 Convert(uint32_t color)
  {
    return ray_tracing::vec3(static_cast<float>((color >> 16) & 0xFFu) / 255.0f,
                             static_cast<float>((color >> 8) & 0xFFu) / 255.0f,
                             static_cast<float>(color & 0xFFu) / 255.0f);
  }

private:
  // blue [0,20), warm [20,40), green [40,60), yellow [60,70), red [70,80)
  static constexpr int kFirst[kFamilies + 1] = {0, 20, 40, 60, 70, 80};
  static constexpr uint32_t kColors[80] = {
      // blue
      0x011f4b, 0x03396c, 0x005b96, 0x6497b1, 0xb3cde0, 0x2a4d69, 0x4b86b4, 0xadcbe3, 0xe7eff6, 0x63ace5,
      0x6e7f80, 0x536872, 0x708090, 0x536878, 0x36454f, 0x3385c6, 0x4279a3, 0x476c8a, 0x49657b, 0x7f8e9e,
      // warm
      0xeee3e7, 0xead5dc, 0xeec9d2, 0xf4b6c2, 0xf6abb6, 0xf6cd61, 0xfe8a71, 0xfe9c8f, 0xfeb2a8, 0xfec8c1,
      0xfad9c1, 0xf9caa7, 0xee4035, 0xf37736, 0xfdf498, 0xff77aa, 0xff99cc, 0xffbbee, 0xff5588, 0xff3377,
      // green
      0x0e9aa7, 0x3da4ab, 0x009688, 0x35a79c, 0x54b2a9, 0x65c3ba, 0x83d0c9, 0x7bc043, 0x96ceb4, 0x88d8b0,
      0xa8e6cf, 0xdcedc1, 0x00b159, 0xe5fcc2, 0x9de0ad, 0x45ada8, 0xc8c8a9, 0x83af9b, 0xa8e6ce, 0xdcedc2,
      // yellow
      0xfed766, 0xf6cd61, 0xfdf498, 0xffeead, 0xffcc5c, 0xffa700, 0xffc425, 0xedc951, 0xf1c27d, 0xffdbac,
      // red
      0xfe4a49, 0x851e3e, 0xee4035, 0xff6f69, 0xd62d20, 0xff8b94, 0xd11141, 0xcc2a36, 0xec2049, 0xfe4365};
};
}  // namespace demo
