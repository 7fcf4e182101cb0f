// tables_arith_test.cpp — heatray_amd/csrc/hr_tables.h against the host's <random> (libstdc++: the contract the header states) and against a
// literal restatement of BlueNoise.h's hash.  The MT19937 part runs the kernel's schedule on the CPU: the state is twisted in three rounds of
// 208 "lanes", every lane of a round reading before any lane of it writes, exactly as k_mt_tables does with a barrier between.
// g++ -std=c++17 -O1 -ffp-contract=off -I heatray_amd/csrc tables_arith_test.cpp && ./a.out      (prints "tables arith: ok")
#include "hr_tables.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace hr;

struct MtLanes { // the kernel's view of the generator
    uint32_t st[kMtN], draws[kMtN];
    int next = kMtN;
    explicit MtLanes(uint32_t seed)
    {
        st[0] = seed;
        for (uint32_t i = 1; i < (uint32_t)kMtN; ++i) st[i] = mtSeedNext(st[i - 1], i);
    }
    void block()
    {
        for (int r = 0; r < 3; ++r) {
            uint32_t w[208];
            for (int t = 0; t < 208; ++t) {
                const uint32_t i = (uint32_t)(r * 208 + t);
                w[t] = mtTwist(st[i], st[(i + 1) % kMtN], st[(i + kMtM) % kMtN]);
            }
            for (int t = 0; t < 208; ++t) st[r * 208 + t] = w[t];
        }
        for (int k = 0; k < kMtN; ++k) draws[k] = mtTemper(st[k]);
        next = 0;
    }
    uint32_t operator()()
    {
        if (next == kMtN) block();
        return draws[next++];
    }
};

static int fail(const char *what, long i)
{
    std::fprintf(stderr, "tables arith: %s differs at %ld\n", what, i);
    return 1;
}

int main()
{
    for (uint32_t seed : {0u, 1u, 15u, 5489u, 0xFFFFFFFFu}) {
        { // raw draws across several twists
            std::mt19937 ref(seed);
            MtLanes mine(seed);
            for (long i = 0; i < 5000; ++i)
                if ((uint32_t)ref() != mine()) return fail("mt19937 draw", i);
        }
        { // uniform_real_distribution<float>(0, 1)
            std::mt19937 ref(seed);
            MtLanes mine(seed);
            std::uniform_real_distribution<float> d(0.0f, 1.0f);
            for (long i = 0; i < 5000; ++i) {
                const float a = d(ref), b = mtCanonical(mine());
                if (std::memcmp(&a, &b, 4)) return fail("uniform_real", i);
            }
        }
        for (uint32_t range : {5u, 6u, 8u, 3u, 1000003u, 0x80000001u}) { // uniform_int_distribution<int>(0, range - 1), interleaved with floats
            if (range > 0x7FFFFFFFu) continue;
            std::mt19937 ref(seed);
            MtLanes mine(seed);
            std::uniform_int_distribution<int> di(0, (int)range - 1);
            std::uniform_real_distribution<float> df(0.0f, 1.0f);
            for (long i = 0; i < 3000; ++i) {
                MtIntDraw pick{range, 0u, 0ull};
                bool first = true;
                while (!pick.accept(mine(), first)) first = false;
                if (di(ref) != pick.value()) return fail("uniform_int", i);
                const float a = df(ref), b = mtCanonical(mine());
                if (std::memcmp(&a, &b, 4)) return fail("uniform_real after int", i);
            }
        }
    }
    // the rounding edge of generate_canonical: draws that convert to 2^32
    if (mtCanonical(0xFFFFFFFFu) != std::nextafter(1.0f, 0.0f) || mtCanonical(0xFFFFFF80u) != std::nextafter(1.0f, 0.0f)) return fail("canonical edge", 0);
    if (mtCanonical(0xFFFFFF7Fu) >= 1.0f || mtCanonical(0u) != 0.0f) return fail("canonical edge", 1);
    // FNV-1a with sign-extended bytes (BlueNoise.h:97-100 / Hash.h as oracle/oracle_qmc.cpp restates them)
    for (uint32_t seed : {0u, 1u, 0x80u, 0xFFu, 0x12345678u, 0xFEDCBA98u, 0xFFFFFFFFu}) {
        uint64_t a = 0xcbf29ce484222325ull;
        const signed char *p = (const signed char *)&seed;
        for (int i = 0; i < 4; ++i) a ^= (uint64_t)(int64_t)p[i], a *= 0x100000001b3ull;
        uint64_t b = 0xcbf29ce484222325ull;
        p = (const signed char *)&a;
        for (int i = 0; i < 8; ++i) b ^= (uint64_t)(int64_t)p[i], b *= 0x100000001b3ull;
        const float want = (float)b / (float)UINT64_MAX, got = blueRandom(seed);
        if (std::memcmp(&want, &got, 4) || fnv1a32(seed) != a) return fail("blueRandom", (long)seed);
        int idx = (int)seed;
        uint64_t h = 0xcbf29ce484222325ull;
        p = (const signed char *)&idx;
        for (int i = 0; i < 4; ++i) h ^= (uint64_t)(int64_t)p[i], h *= 0x100000001b3ull;
        if ((uint32_t)(int)h != blueSeed((int32_t)seed)) return fail("blueSeed", (long)seed);
    }
    std::puts("tables arith: ok");
    return 0;
}
