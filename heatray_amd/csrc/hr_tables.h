// hr_tables.h — the arithmetic of the sample tables that have no closed form per index: util::uniformRandomFloats and
// util::randomPolygonal (std::mt19937 + the standard distributions, /root/reference/Source/Utility/Random.h:113-130, 293-355) and
// util::blueNoise (best-candidate points over an FNV-1a hash, /root/reference/Source/Utility/BlueNoise.h:32-100, Random.h:158-165).
// Pure integer / float32 functions, no memory, so that the kernels in hr_build.hip and the CPU unit test
// (tests/host/tables_arith_test.cpp, against the host's own <random>) compile the same lines.
//
// The reference leaves the distributions to the C++ library it is built with.  The contract here is libstdc++'s (GCC 11+), the library
// the reference's headers were compiled with to make tests/golden/ref_vectors.npz:
//   std::mt19937                                   the standard's MT19937 (seed, twist and tempering are fixed by [rand.eng.mers])
//   std::uniform_real_distribution<float>(0, 1)    generate_canonical<float, 24>: ONE draw u, float(u) / 2^32, a result of 1.0f replaced by
//                                                  nextafter(1, 0) (bits/random.tcc:3348-3380)
//   std::uniform_int_distribution<int>(0, n - 1)   Lemire's multiply-and-reject on the 32-bit draw (bits/uniform_int_dist.h:246-270, 311-317)
#pragma once
#include <stdint.h>

#ifdef HRD
#define HRT HRD
#else
#define HRT inline
#endif

namespace hr {

static constexpr int kMtN = 624, kMtM = 397;

// state[i] from state[i-1] ([rand.eng.mers]: f = 1812433253, w = 32)
HRT uint32_t mtSeedNext(uint32_t prev, uint32_t i) { return 1812433253u * (prev ^ (prev >> 30)) + i; }
// the twist of one word: `cur` and `next` are words i and i+1 before the twist, `far` is word (i + 397) mod 624 — the value the serial
// loop would see there (not yet twisted for i < 227, already twisted after)
HRT uint32_t mtTwist(uint32_t cur, uint32_t next, uint32_t far)
{
    const uint32_t y = (cur & 0x80000000u) | (next & 0x7FFFFFFFu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
HRT uint32_t mtTemper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}

// uniform_real_distribution<float>(0, 1) of one draw
HRT float mtCanonical(uint32_t u)
{
    const float r = (float)u * 0x1p-32f; // float(u) rounds to nearest (it may reach 2^32); the scale by 2^-32 is exact
    return r >= 1.0f ? 0.99999994f : r;
}

// uniform_int_distribution<int>(0, range - 1): the state of Lemire's method between draws.  accept(u, true) takes the first draw; while
// it returns false the caller passes further draws to accept(u, false); value() is the result.
struct MtIntDraw {
    uint32_t range, threshold;
    uint64_t product;
    HRT bool accept(uint32_t u, bool firstDraw)
    {
        product = (uint64_t)u * (uint64_t)range;
        const uint32_t low = (uint32_t)product;
        if (firstDraw) {
            if (low >= range) return true;
            threshold = (0u - range) % range;
        }
        return low >= threshold;
    }
    HRT int value() const { return (int)(product >> 32); }
};

// FNV-1a over the bytes of a 32-bit / 64-bit value as the reference hashes them: every byte is sign-extended before the xor
// (BlueNoise.h:97-100 through Hash.h; oracle/oracle_qmc.cpp::fnv1a)
HRT uint64_t fnv1aByte(uint64_t h, uint32_t byte)
{
    h ^= (uint64_t)(int64_t)(int8_t)byte;
    return h * 0x100000001b3ull;
}
HRT uint64_t fnv1a32(uint32_t v)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < 4; ++i) h = fnv1aByte(h, (v >> (8 * i)) & 255u);
    return h;
}
HRT uint64_t fnv1a64(uint64_t v)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < 8; ++i) h = fnv1aByte(h, (uint32_t)(v >> (8 * i)) & 255u);
    return h;
}
// BlueNoise.h:97-100: a float in [0, 1] (1.0 when the hash rounds up to 2^64)
HRT float blueRandom(uint32_t seed) { return (float)fnv1a64(fnv1a32(seed)) * 0x1p-64f; }
// the generator's first seed for a sequence (BlueNoise.h:57: the hash of the index, truncated to int)
HRT uint32_t blueSeed(int32_t sequenceIndex) { return (uint32_t)fnv1a32((uint32_t)sequenceIndex); }

} // namespace hr
