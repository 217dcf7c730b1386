// scene_random.hpp -- the seeded generator the demo scenes are built from.
//
// The reference seeds std::mt19937 from std::random_device and draws through the standard library's
// distributions (demos/pretty_spheres.hpp:38-43), so no two runs -- and no two standard libraries -- give
// the same scene.  Benchmarks and parity checks need the same spheres every time, so the scenes here are
// drawn from a small generator with a fully specified output: PCG32 (O'Neill 2014, XSH-RR 64/32).
#pragma once

#include <cstdint>

namespace demo
{
class SceneRandom
{
public:
  explicit SceneRandom(uint64_t seed)
  {
    m_state = 0;
    NextU32();
    m_state += seed;
    NextU32();
  }

  uint32_t NextU32()
  {
    uint64_t const old = m_state;
    m_state = old * 6364136223846793005ULL + 1442695040888963407ULL;
    uint32_t const xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
    uint32_t const rot = static_cast<uint32_t>(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
  }

  // Uniform in [0, 1) with 24 random bits.
  float Float01() { return static_cast<float>(NextU32() >> 8) * (1.0f / 16777216.0f); }

  // Uniform integer in [lo, hi].
  int Int(int lo, int hi)
  {
    uint32_t const span = static_cast<uint32_t>(hi - lo + 1);
    return lo + static_cast<int>(NextU32() % span);
  }

private:
  uint64_t m_state;
};
}  // namespace demo
