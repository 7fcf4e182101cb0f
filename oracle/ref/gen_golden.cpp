// gen_golden.cpp — TEST INFRASTRUCTURE, not product code.
//
// Compiles the reference's OWN header-only pieces where they lie under
// /root/reference (never copied into this repo) and dumps their outputs as raw
// little-endian arrays.  tests/golden/make_golden.py packs the dump into the
// committed fixture tests/golden/ref_vectors.npz.  Built by oracle/Makefile
// target `ref` into oracle/_ref/ (git-ignored).
//
// Reference pieces exercised (file:line in /root/reference/Source):
//   Utility/Random.h:85-355          sobol / halton / hammersley / blueNoise /
//                                    uniformRandomFloats / radialSobol / randomPolygonal
//   HeatrayRenderer/OrbitCamera.h:32-45   createViewMatrix
//   HeatrayRenderer/Scene/SphereMeshProvider.h, PlaneMeshProvider.h  built-in geometry
//
// The only accommodation for this toolchain: libstdc++ 11 does not declare
// std::sqrtf/cosf/sinf (MSVC and libc++ do); compat_std_math.h adds the
// using-declarations so the unmodified reference headers compile.
#include "compat_std_math.h"

#include <glm/glm/glm.hpp>
#include <glm/glm/gtc/constants.hpp>

#include <Utility/Random.h>
#include <HeatrayRenderer/OrbitCamera.h>
#include <HeatrayRenderer/Scene/MeshProvider.h>
#include <HeatrayRenderer/Scene/SphereMeshProvider.h>
#include <HeatrayRenderer/Scene/PlaneMeshProvider.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static std::string g_dir;

static void dump(const std::string &name, const void *data, size_t bytes)
{
    std::string path = g_dir + "/" + name + ".bin";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        perror(path.c_str());
        exit(1);
    }
    fwrite(data, 1, bytes, f);
    fclose(f);
}

static void dumpProvider(MeshProvider &p, const std::string &prefix)
{
    for (size_t b = 0; b < p.GetVertexBufferCount(); ++b) {
        std::vector<uint8_t> buf(p.GetVertexBufferSize(b));
        p.FillVertexBuffer(b, buf.data());
        dump(prefix + "_vb" + std::to_string(b), buf.data(), buf.size());
    }
    std::vector<uint8_t> ib(p.GetIndexBufferSize(0));
    p.FillIndexBuffer(0, ib.data());
    dump(prefix + "_ib", ib.data(), ib.size());
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: gen_golden <outdir>\n");
        return 2;
    }
    g_dir = argv[1];

    const uint32_t lengths[2] = {32, 1024};
    for (uint32_t P : lengths) {
        std::vector<glm::vec2> v(P);
        for (uint32_t seq = 0; seq < 16; ++seq) {
            std::string tag = "_p" + std::to_string(P) + "_s" + std::to_string(seq);
            util::sobol(v.data(), P, seq);
            dump("sobol" + tag, v.data(), sizeof(glm::vec2) * P);
            util::halton(v.data(), P, (int)seq);
            dump("halton" + tag, v.data(), sizeof(glm::vec2) * P);
            util::hammersley(v.data(), P, (int)seq);
            dump("hammersley" + tag, v.data(), sizeof(glm::vec2) * P);
            util::radialSobol(v.data(), P, seq);
            dump("radialsobol" + tag, v.data(), sizeof(glm::vec2) * P);
            if (P == 32) {
                util::blueNoise(v.data(), P, (int)seq);
                dump("bluenoise" + tag, v.data(), sizeof(glm::vec2) * P);
                // libstdc++-specific (std::uniform_*_distribution) — pins THIS toolchain only.
                util::uniformRandomFloats<glm::vec2>(v.data(), P, seq, 0.0f, 1.0f);
                dump("random" + tag, v.data(), sizeof(glm::vec2) * P);
                for (uint32_t edges : {5u, 6u, 8u}) {
                    util::randomPolygonal(v.data(), edges, P, seq);
                    dump("polygon" + std::to_string(edges) + tag, v.data(), sizeof(glm::vec2) * P);
                }
            }
        }
    }

    // generateSequenceOffsets(64, 64): sobol(W*H points, sequence 0) (PassGenerator.cpp:150-159).
    {
        std::vector<glm::vec2> v(64 * 64);
        util::sobol(v.data(), (uint32_t)v.size(), 0);
        dump("seqoffsets_64x64", v.data(), sizeof(glm::vec2) * v.size());
    }
    // The 4096-point sequence the multiscatter LUT generator integrates with
    // (MultiScatterUtil.cpp:102-104).
    {
        std::vector<glm::vec2> v(4096);
        util::sobol(v.data(), 4096, 0);
        dump("sobol_p4096_s0", v.data(), sizeof(glm::vec2) * v.size());
    }
    // Integer building blocks (Random.h:26-82).
    {
        std::vector<uint32_t> in, h, rb, lk, nus;
        uint32_t x = 0x12345678u;
        for (int i = 0; i < 256; ++i) {
            x = x * 1664525u + 1013904223u;
            uint32_t seed = util::burleyHash((uint32_t)i + 1);
            in.push_back(x);
            h.push_back(util::burleyHash(x));
            rb.push_back(util::reverseBits(x));
            lk.push_back(util::laineKarrasPermutation(x, seed));
            nus.push_back(util::nestedUniformScramble(x, seed));
        }
        dump("int_in", in.data(), 4 * in.size());
        dump("int_burleyhash", h.data(), 4 * h.size());
        dump("int_reversebits", rb.data(), 4 * rb.size());
        dump("int_lainekarras", lk.data(), 4 * lk.size());
        dump("int_nestedscramble", nus.data(), 4 * nus.size());
    }

    // OrbitCamera view matrices: rows of (distance, phi, theta, tx, ty, tz) -> mat4 (column-major).
    {
        const float params[][6] = {
            {19.0f, 0.0f, 0.0f, 0, 0, 0},   {3.0f, 0.6f, 0.3f, 0, 0, 0},      {5.5f, 2.1f, -0.4f, 0.25f, -1.0f, 2.0f},
            {10.0f, 5.9f, 1.2f, 1, 2, 3},   {0.5f, 3.14159f, -1.5f, 0, 0.5f, 0}};
        std::vector<float> in, out;
        for (auto &p : params) {
            OrbitCamera cam;
            cam.distance = p[0];
            cam.phi = p[1];
            cam.theta = p[2];
            cam.target = glm::vec3(p[3], p[4], p[5]);
            glm::mat4 m = cam.createViewMatrix();
            in.insert(in.end(), p, p + 6);
            out.insert(out.end(), &m[0][0], &m[0][0] + 16);
        }
        dump("orbit_params", in.data(), 4 * in.size());
        dump("orbit_matrices", out.data(), 4 * out.size());
    }

    // Built-in fixture geometry.
    {
        SphereMeshProvider sphere(50, 50, 1.0f, "Sphere");
        dumpProvider(sphere, "sphere50");
        SphereMeshProvider sphereSmall(8, 6, 0.5f, "SphereSmall");
        dumpProvider(sphereSmall, "sphere8x6");
        PlaneMeshProvider plane(15, 15, "Plane");
        dumpProvider(plane, "plane15");
    }
    return 0;
}
