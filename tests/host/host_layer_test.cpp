// host_layer_test — exercises the C++ drop-in layer (heatray_amd/host) the way the Heatray viewer does:
// PassGenerator::init / loadScene / changeLighting / renderPass with a completion callback on the worker
// thread, Scene::addMesh through a MeshProvider, material and light classes, openrl::PixelPackBuffer.
//
//   host_layer_test --cpu-checks                     host-only checks (no device): baking, light packing, defaults
//   host_layer_test <scene.bin> <out_dir> <passes>   render on the GPU and dump pixels + the baked blocks, so that
//                                                    tests/test_gpu_host_layer.py can replay the same inputs through
//                                                    the CPU oracle and demand a bit-exact buffer
#include <HeatrayRenderer/PassGenerator.h>
#include <HeatrayRenderer/Scene/Scene.h>
#include <HeatrayRenderer/Scene/MeshProvider.h>
#include <HeatrayRenderer/Materials/PhysicallyBasedMaterial.h>
#include <HeatrayRenderer/Materials/GlassMaterial.h>
#include <HeatrayRenderer/Lights/DirectionalLight.h>
#include <HeatrayRenderer/Lights/PointLight.h>
#include <HeatrayRenderer/Lights/SpotLight.h>
#include <HeatrayRenderer/Lights/EnvironmentLight.h>
#include <RLWrapper/PixelPackBuffer.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            exit(1);                                                         \
        }                                                                    \
    } while (0)

// ---- a MeshProvider over arrays read from a file: planar buffers like the application's providers
struct FileMesh {
    std::vector<float> positions, normals, uvs;
    std::vector<int> indices;
    int strip = 0;
    int material = 0;
    glm::mat4 transform = glm::mat4(1.0f);
};

class ArrayMeshProvider : public MeshProvider
{
public:
    explicit ArrayMeshProvider(const FileMesh& mesh) : MeshProvider("array mesh"), m(mesh) {}
    size_t GetVertexBufferCount() override { return m.uvs.empty() ? 2 : 3; }
    size_t GetVertexBufferSize(size_t i) override { return buffer(i).size() * sizeof(float); }
    void FillVertexBuffer(size_t i, uint8_t* out) override { memcpy(out, buffer(i).data(), buffer(i).size() * sizeof(float)); }
    size_t GetIndexBufferCount() override { return 1; }
    size_t GetIndexBufferSize(size_t) override { return m.indices.size() * sizeof(int); }
    void FillIndexBuffer(size_t, uint8_t* out) override { memcpy(out, m.indices.data(), m.indices.size() * sizeof(int)); }
    size_t GetSubmeshCount() override { return 1; }
    Submesh GetSubmesh(size_t) override
    {
        Submesh s;
        s.vertexAttributeCount = m.uvs.empty() ? 2 : 3;
        const VertexAttributeUsage usage[3] = { VertexAttributeUsage_Position, VertexAttributeUsage_Normal, VertexAttributeUsage_TexCoord };
        const int comps[3] = { 3, 3, 2 };
        for (int k = 0; k < s.vertexAttributeCount; ++k) {
            s.vertexAttributes[k].usage = usage[k];
            s.vertexAttributes[k].buffer = k;
            s.vertexAttributes[k].componentCount = comps[k];
            s.vertexAttributes[k].size = sizeof(float);
            s.vertexAttributes[k].offset = 0;
            s.vertexAttributes[k].stride = comps[k] * (int)sizeof(float);
        }
        s.indexBuffer = 0;
        s.indexOffset = 0;
        s.elementCount = m.indices.size();
        s.drawMode = m.strip ? DrawMode::TriangleStrip : DrawMode::Triangles;
        s.localTransform = glm::mat4(1.0f);
        s.name = "array";
        return s;
    }

private:
    const std::vector<float>& buffer(size_t i) const { return i == 0 ? m.positions : (i == 1 ? m.normals : m.uvs); }
    const FileMesh& m;
};

template <class T> static void readVec(FILE* f, std::vector<T>& v)
{
    int32_t n = 0;
    CHECK(fread(&n, 4, 1, f) == 1);
    v.resize(n);
    if (n) CHECK(fread(v.data(), sizeof(T), n, f) == (size_t)n);
}

static void dump(const std::string& path, const void* p, size_t bytes)
{
    FILE* f = fopen(path.c_str(), "wb");
    CHECK(f);
    fwrite(p, 1, bytes, f);
    fclose(f);
}

// ------------------------------------------------------------------------------------------------
static int cpuChecks()
{
    // RenderOptions defaults of the reference (PassGenerator.h:49-150)
    PassGenerator::RenderOptions opts;
    CHECK(opts.enableInteractiveMode && !opts.enableOfflineMode && opts.resetInternalState);
    CHECK(opts.maxRenderPasses == 32 && opts.maxRayDepth == 10);
    CHECK(std::fabs(opts.maxChannelValue - 3.14159265f) < 1e-6f);
    CHECK(opts.camera.fstop == 32.0f && opts.camera.focalLength == 50.0f && opts.camera.aspectRatio == -1.0f);
    CHECK(opts.sampleMode == PassGenerator::RenderOptions::SampleMode::kSobol);
    CHECK(PassGenerator::kNumRandomSequences == 16);
    opts.camera.fstop = 2.8f;
    opts.camera.setApertureRadius();
    CHECK(std::fabs(opts.camera.apertureRadius - (50.0f / 2.8f) / 1000.0f) < 1e-9f);

    // PBR baking (PhysicallyBasedMaterial.cpp:127-146 of the reference)
    PhysicallyBasedMaterial::Parameters p;
    p.baseColor = glm::vec3(1.5f, 0.5f, -0.2f);
    p.roughness = 0.0f;
    p.metallic = 2.0f;
    p.specularF0 = 0.5f;
    p.clearCoat = 1.0f;
    p.clearCoatRoughness = 0.3f;
    p.alphaMask = true;
    hr_material row;
    PhysicallyBasedMaterial::bake(p, true, 7, &row);
    CHECK(row.type == HR_MAT_PBR);
    CHECK(row.base_color[0] == 1.0f && row.base_color[1] == 0.5f && row.base_color[2] == 0.0f);
    CHECK(row.roughness == 0.01f && row.roughness_alpha == 0.01f * 0.01f && row.metallic == 1.0f);
    CHECK(row.specular_f0 == 0.5f * 0.08f && row.clear_coat == 0.2f);
    CHECK(row.clear_coat_roughness == 0.3f && row.clear_coat_roughness_alpha == 0.3f * 0.3f);
    CHECK(row.flags == (HR_MF_DOUBLE_SIDED | HR_MF_ALPHA_MASK | HR_MF_VERTEX_COLORS));
    CHECK(row.multiscatter_lut == 7 && row.base_color_texture == HR_TEX_NONE);
    p.forceEnableAllTextures = true;
    PhysicallyBasedMaterial::bake(p, false, -1, &row);
    CHECK(row.flags == (HR_MF_HAS_BASE_COLOR_TEXTURE | HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | HR_MF_HAS_CLEARCOAT_TEXTURE |
                        HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE | HR_MF_DOUBLE_SIDED | HR_MF_ALPHA_MASK));

    // glass baking (GlassMaterial.cpp:88-107 of the reference)
    GlassMaterial::Parameters g;
    g.ior = 1.5f;
    g.roughness = 0.2f;
    g.density = 0.5f;
    GlassMaterial::bake(g, false, &row);
    CHECK(row.type == HR_MAT_GLASS && row.ior == 1.5f && row.density == 0.5f);
    CHECK(std::fabs(row.specular_f0 - 0.04f) < 1e-7f && row.roughness_alpha == 0.2f * 0.2f);

    // light packing: tightly packed block, swap-with-last on removal (Lighting.cpp:301-333 of the reference)
    Lighting lighting; // no device context: the block is only kept on the host
    auto d0 = lighting.addDirectionalLight("d0");
    auto d1 = lighting.addDirectionalLight("d1");
    auto d2 = lighting.addDirectionalLight("d2");
    DirectionalLight::Params dp = d2->params();
    dp.color = glm::vec3(0.5f, 0.25f, 1.0f);
    dp.illuminance = 683.0f * 2.0f;
    d2->setParams(dp);
    lighting.updateLight(d2);
    CHECK(lighting.block().n_directional == 3);
    CHECK(std::fabs(lighting.block().directional_colors[2][0] - 1.0f) < 1e-6f && std::fabs(lighting.block().directional_colors[2][2] - 2.0f) < 1e-6f);
    // default orientation (theta = pi/2): the light is straight up
    CHECK(std::fabs(lighting.block().directional_directions[0][1] - 1.0f) < 1e-6f);
    lighting.removeLight(d0);
    CHECK(lighting.block().n_directional == 2);
    CHECK(lighting.directionalLights()[0] == d2 && lighting.directionalLights()[1] == d1);
    CHECK(std::fabs(lighting.block().directional_colors[0][2] - 2.0f) < 1e-6f);
    for (int i = 0; i < 3; ++i) CHECK(lighting.addDirectionalLight("x") != nullptr);
    CHECK(lighting.addDirectionalLight("too many") == nullptr);

    auto pl = lighting.addPointLight("p");
    PointLight::Params pp = pl->params();
    pp.position = glm::vec3(1, 2, 3);
    pp.luminousIntensity = 683.0f;
    pl->setParams(pp);
    lighting.updateLight(pl);
    CHECK(lighting.block().n_point == 1 && lighting.block().point_positions[0][1] == 2.0f);
    CHECK(std::fabs(lighting.block().point_colors[0][0] - 4.0f * 3.14159265f) < 1e-5f);

    auto sl = lighting.addSpotLight("s");
    SpotLight::Params sp = sl->params();
    sp.innerAngle = 1.0f;
    sp.outerAngle = 0.5f; // inner > outer: clamped to outer - 1 degree (SpotLight.cpp:58-69)
    sl->setParams(sp);
    lighting.updateLight(sl);
    CHECK(std::fabs(sl->params().innerAngle - (0.5f - 0.0174532925f)) < 1e-6f);
    CHECK(std::fabs(lighting.block().spot_angles[0][1] - std::cos(0.5f)) < 1e-6f);
    CHECK(lighting.block().env_enabled == 0);
    auto env = lighting.addEnvironmentLight();
    env->setExposure(2.0f);
    lighting.updateLight(env);
    CHECK(lighting.block().env_enabled == 1 && lighting.block().env_exposure == 4.0f);
    lighting.clearAllButEnvironment();
    CHECK(lighting.block().n_directional == 0 && lighting.block().n_point == 0 && lighting.block().env_enabled == 1);
    lighting.clear();
    CHECK(lighting.block().env_enabled == 0);
    printf("host layer cpu checks: ok\n");
    return 0;
}

// ------------------------------------------------------------------------------------------------
int main(int argc, char** argv)
{
    if (argc >= 2 && std::string(argv[1]) == "--cpu-checks") return cpuChecks();
    if (argc < 4) {
        fprintf(stderr, "usage: host_layer_test --cpu-checks | <scene.bin> <out_dir> <passes>\n");
        return 2;
    }
    const std::string outDir = argv[2];
    const int passes = atoi(argv[3]);

    // scene file: int32 width, height, depth, nMeshes; per mesh: strip, material, mat4, positions, normals, uvs, indices
    FILE* f = fopen(argv[1], "rb");
    CHECK(f);
    int32_t hdr[4];
    CHECK(fread(hdr, 4, 4, f) == 4);
    const int width = hdr[0], height = hdr[1], depth = hdr[2], nMeshes = hdr[3];
    float view[16];
    CHECK(fread(view, 4, 16, f) == 16);
    std::vector<FileMesh> meshes(nMeshes);
    for (FileMesh& m : meshes) {
        int32_t mh[2];
        CHECK(fread(mh, 4, 2, f) == 2);
        m.strip = mh[0], m.material = mh[1];
        CHECK(fread(&m.transform[0][0], 4, 16, f) == 16);
        readVec(f, m.positions), readVec(f, m.normals), readVec(f, m.uvs), readVec(f, m.indices);
    }
    fclose(f);

    PassGenerator renderer;
    renderer.init(width, height);

    // materials as the viewer's built-in "Multi-Material" scene creates them (HeatrayRenderer.cpp:157-236)
    std::vector<std::shared_ptr<Material>> created;
    renderer.loadScene([&](std::shared_ptr<Scene> scene) {
        for (FileMesh& m : meshes) {
            std::shared_ptr<Material> material;
            if (m.material == 2) {
                auto glass = std::make_shared<GlassMaterial>("Glass");
                GlassMaterial::Parameters& params = glass->parameters();
                params.roughness = 0.1f;
                params.baseColor = glm::vec3(0.9f, 0.6f, 0.6f);
                params.ior = 1.57f;
                params.density = 0.5f;
                material = glass;
            } else {
                auto pbr = std::make_shared<PhysicallyBasedMaterial>("PBR");
                PhysicallyBasedMaterial::Parameters& params = pbr->parameters();
                if (m.material == 0) {
                    params.metallic = 0.0f, params.roughness = 1.0f, params.baseColor = glm::vec3(0.9f), params.specularF0 = 0.0f;
                } else if (m.material == 1) {
                    params.metallic = 1.0f, params.roughness = 0.1f, params.baseColor = glm::vec3(0.4f), params.specularF0 = 0.3f;
                } else {
                    params.metallic = 0.0f, params.roughness = 0.4f, params.baseColor = glm::vec3(0.2f, 0.5f, 0.9f);
                    params.clearCoat = 1.0f, params.clearCoatRoughness = 0.1f;
                }
                material = pbr;
            }
            created.push_back(material);
            ArrayMeshProvider provider(m);
            scene->addMesh(&provider, { material }, m.transform);
        }
    });
    renderer.changeLighting([](std::shared_ptr<Lighting> lighting) {
        auto dl = lighting->addDirectionalLight("sun");
        DirectionalLight::Params dp = dl->params();
        dp.illuminance = 683.0f * 2.0f;
        dp.orientation.phi = 0.5f;
        dp.orientation.theta = 1.0f;
        dl->setParams(dp);
        lighting->updateLight(dl);
        auto pl = lighting->addPointLight("bulb");
        PointLight::Params pp = pl->params();
        pp.position = glm::vec3(0.0f, 2.5f, 1.5f);
        pp.color = glm::vec3(1.0f, 0.9f, 0.8f);
        pp.luminousIntensity = 683.0f * 1.5f;
        pl->setParams(pp);
        lighting->updateLight(pl);
        auto sl = lighting->addSpotLight("spot");
        SpotLight::Params sp = sl->params();
        sp.position = glm::vec3(-2.0f, 3.0f, 2.0f);
        sp.color = glm::vec3(0.7f, 0.8f, 1.0f);
        sp.luminousIntensity = 683.0f * 6.0f;
        sp.orientation.phi = -0.7f;
        sp.orientation.theta = 0.9f;
        sp.innerAngle = 0.2617994f;
        sp.outerAngle = 0.6108652f;
        sl->setParams(sp);
        lighting->updateLight(sl);
    });

    PassGenerator::RenderOptions options;
    options.enableInteractiveMode = false;
    options.maxRayDepth = (uint32_t)depth;
    options.maxRenderPasses = 16;
    options.environment.map = std::string(EnvironmentLight::SOLID_COLOR);
    options.environment.solidColor = glm::vec3(0.5f);
    options.camera.aspectRatio = float(width) / float(height);
    options.camera.focusDistance = 8.0f;
    options.camera.fstop = PassGenerator::RenderOptions::Camera::fstopOptions[0];
    options.camera.setApertureRadius();
    memcpy(&options.camera.viewMatrix[0][0], view, sizeof(view));

    std::atomic<int> completed{0};
    std::vector<float> pixels;
    std::vector<uint32_t> displayPixels;
    // the viewer's PostProcessingParams (HeatrayRenderer.h:104-117), non-default values
    struct {
        bool tonemapping_enabled = true;
        float exposure = 0.75f, brightness = 0.05f, contrast = 1.05f, hue = 1.0f, saturation = 1.3f, vibrance = 0.4f;
        float red = 1.1f, green = 0.9f, blue = 1.0f, vignetteIntensity = 0.5f, vignetteFalloff = 0.4f;
    } post;
    size_t lastIndex = 0;
    float lastSamples = 0.0f;
    for (int p = 0; p < passes; ++p) {
        options.resetInternalState = (p == 0);
        renderer.renderPass(options, [&](bool frameDataAvailable, std::shared_ptr<openrl::PixelPackBuffer> results, float passTime, size_t passIndex) {
            CHECK(frameDataAvailable);
            CHECK(results->width() == width && results->height() == height);
            const float* mapped = results->mapPixelData(); // left mapped, like the viewer does between frames
            pixels.assign(mapped, mapped + (size_t)width * height * 4);
            // progressive display: complete passes only, at least one, never more than requested; the last one is complete
            const float samples = pixels[3];
            CHECK(samples >= 1.0f && samples <= float(passIndex));
            CHECK(samples >= lastSamples);
            if (passIndex == (size_t)passes) CHECK(samples == float(passes));
            lastSamples = samples;
            // display-ready pixels from the device (the extension of SURVEY §8f row 1)
            const uint32_t* shown = (const uint32_t*)results->resolveForDisplay(openrl::PixelPackBuffer::displayParams(post), HR_DISPLAY_RGBA8);
            CHECK(shown != nullptr);
            displayPixels.assign(shown, shown + (size_t)width * height);
            lastIndex = passIndex;
            CHECK(passTime >= 0.0f);
            completed++;
        });
    }
    renderer.waitIdle();
    CHECK(completed == passes && lastIndex == (size_t)passes);

    // dump what the oracle needs to replay this render exactly: the packed light block and the material rows
    dump(outDir + "/pixels.bin", pixels.data(), pixels.size() * sizeof(float));
    dump(outDir + "/display.bin", displayPixels.data(), displayPixels.size() * sizeof(uint32_t));
    const hr_display_params shownWith = openrl::PixelPackBuffer::displayParams(post);
    dump(outDir + "/display_params.bin", &shownWith, sizeof(shownWith));
    dump(outDir + "/lights.bin", &renderer.scene()->lighting()->block(), sizeof(hr_lights));
    std::vector<hr_material> rows;
    std::vector<int32_t> rowIds;
    for (auto& material : created) {
        hr_material row;
        if (material->type() == Material::Type::Glass) {
            GlassMaterial::bake(std::static_pointer_cast<GlassMaterial>(material)->parameters(), false, &row);
        } else {
            PhysicallyBasedMaterial::bake(std::static_pointer_cast<PhysicallyBasedMaterial>(material)->parameters(), false, 0 /* LUT = first texture */, &row);
        }
        rows.push_back(row);
        rowIds.push_back(material->tableIndex());
    }
    dump(outDir + "/materials.bin", rows.data(), rows.size() * sizeof(hr_material));
    dump(outDir + "/material_ids.bin", rowIds.data(), rowIds.size() * sizeof(int32_t));
    {
        // the camera uniforms exactly as PassGenerator::runRenderFrameJob derives them (same float expressions)
        const float fovY = 2.0f * std::atan2(24.0f, 2.0f * options.camera.focalLength);
        const float cam[4] = { std::tan(fovY * 0.5f), options.camera.aspectRatio, options.camera.focusDistance, options.camera.apertureRadius };
        dump(outDir + "/camera.bin", cam, sizeof(cam));
    }

    // exercise the offline mode, a scene edit and a resize before shutting down
    options.enableOfflineMode = true;
    options.resetInternalState = true;
    options.maxRenderPasses = 4;
    int offlineCallbacks = 0, offlineFrames = 0;
    renderer.renderPass(options, [&](bool frameDataAvailable, std::shared_ptr<openrl::PixelPackBuffer>, float, size_t) {
        ++offlineCallbacks;
        offlineFrames += frameDataAvailable ? 1 : 0;
    });
    renderer.waitIdle();
    CHECK(offlineCallbacks == 4 && offlineFrames == 1); // every pass reports, only the last one carries pixels
    renderer.modifyScene([](std::shared_ptr<Scene> scene) {
        glm::mat4 t(1.0f);
        t[3][1] = 0.25f;
        scene->applyTransform(t);
    });
    renderer.resize(width / 2, height / 2);
    options.enableOfflineMode = false;
    options.resetInternalState = true;
    options.camera.aspectRatio = float(width / 2) / float(height / 2);
    bool resizedOk = false;
    renderer.renderPass(options, [&](bool, std::shared_ptr<openrl::PixelPackBuffer> results, float, size_t) {
        resizedOk = results->width() == width / 2 && results->height() == height / 2;
    });
    renderer.waitIdle();
    CHECK(resizedOk);
    renderer.destroy();
    printf("host layer render: ok (%d passes, %dx%d)\n", passes, width, height);
    return 0;
}
