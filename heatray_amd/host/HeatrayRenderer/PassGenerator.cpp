#include "PassGenerator.h"

#include "Lights/EnvironmentLight.h"
#include "Scene/Scene.h"

#include <RLWrapper/HrContext.h>
#include <RLWrapper/PixelPackBuffer.h>
#include <RLWrapper/Texture.h>

#if __has_include(<Utility/Random.h>)
#include <Utility/Random.h> // the application's table generators (blue noise, std::-based tables, polygonal bokeh)
#define HR_HOST_HAS_RANDOM_H 1
#endif

#include <algorithm>
#include <assert.h>
#include <chrono>
#include <cmath>
#include <random>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

PassGenerator::~PassGenerator()
{
    if (m_running) {
        destroy();
    }
}

// ---------------------------------------------------------------------------------------------
// Worker thread.  All libhrcore calls happen here, in FIFO order — the threading contract of the
// reference's "OpenRL thread" (PassGenerator.h:229-231, Utility/AsyncTaskQueue.h:129-164).
void PassGenerator::enqueue(Job job)
{
    {
        std::lock_guard<std::mutex> lock(m_queueMutex);
        m_jobs.push_back(std::move(job));
    }
    m_queueSignal.notify_one();
}

void PassGenerator::workerLoop()
{
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> lock(m_queueMutex);
            m_queueSignal.wait(lock, [this] { return !m_jobs.empty(); });
            job = std::move(m_jobs.front());
            m_jobs.pop_front();
            m_busy = true;
        }
        const bool finished = job();
        {
            std::lock_guard<std::mutex> lock(m_queueMutex);
            m_busy = false;
        }
        m_idleSignal.notify_all();
        if (finished) {
            return;
        }
    }
}

void PassGenerator::waitIdle()
{
    std::unique_lock<std::mutex> lock(m_queueMutex);
    m_idleSignal.wait(lock, [this] { return m_jobs.empty() && !m_busy; });
}

// ---------------------------------------------------------------------------------------------
void PassGenerator::init(const RLint renderWidth, const RLint renderHeight)
{
    m_running = true;
    m_worker = std::thread([this] { workerLoop(); });
    enqueue([this, renderWidth, renderHeight] { return !runInitJob(renderWidth, renderHeight); });
}

void PassGenerator::destroy()
{
    enqueue([this] {
        runDestroyJob();
        return true;
    });
    if (m_worker.joinable()) {
        m_worker.join();
    }
    m_running = false;
}

void PassGenerator::resize(RLint newWidth, RLint newHeight)
{
    enqueue([this, newWidth, newHeight] {
        runResizeJob(newWidth, newHeight);
        return false;
    });
}

void PassGenerator::renderPass(const RenderOptions& newOptions, PassCompleteCallback callback)
{
    // The callback travels WITH the job: the reference stores it in a member on the caller's thread and reads it on the worker
    // without synchronisation (PassGenerator.cpp:117 of the reference), a data race this layer does not reproduce.
    RenderOptions options = newOptions;
    enqueue([this, options, callback] {
        runRenderFrameJob(options, callback);
        return false;
    });
}

void PassGenerator::loadScene(LoadSceneCallback callback, bool clearOldScene)
{
    enqueue([this, callback, clearOldScene] {
        if (clearOldScene) {
            m_scene->clearMeshesAndMaterials();
        }
        callback(m_scene);
        return false;
    });
}

void PassGenerator::changeLighting(LightingCallback callback)
{
    enqueue([this, callback] {
        callback(m_scene->lighting());
        return false;
    });
}

void PassGenerator::modifyScene(ModifySceneCallback callback)
{
    enqueue([this, callback] {
        callback(m_scene);
        return false;
    });
}

void PassGenerator::runOpenRLTask(OpenRLTask task)
{
    enqueue([task] {
        task();
        return false;
    });
}

// ---------------------------------------------------------------------------------------------
// PassGenerator.cpp:161-299 of the reference: context, sample tables, accumulation buffer, scene.
bool PassGenerator::runInitJob(const RLint renderWidth, const RLint renderHeight)
{
    hr_ctx_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.world = 1;
    if (hr_abi_version() != HR_ABI_VERSION) { // (a library with longer structs than this layer allocates would corrupt memory silently)
        fprintf(stderr, "PassGenerator: libhrcore speaks ABI %u, this layer was built against %u: rebuild one of them\n", hr_abi_version(), HR_ABI_VERSION);
        return false;
    }
    if (const char* mb = getenv("HEATRAY_MEMORY_BUDGET_MB")) desc.memory_budget = (uint64_t)strtoull(mb, nullptr, 10) << 20;
    if (hr_ctx_create(&desc, &m_context) != HR_OK) {
        fprintf(stderr, "PassGenerator: no usable MI355X / HIP device (there is no CPU fallback)\n");
        return false;
    }
    openrl::currentContext() = m_context;
    {
        const char* est = getenv("HEATRAY_ESTIMATOR");
        m_envMis = est && std::string(est) == "env_mis";
        m_allLights = est && std::string(est) == "all_lights";
        const char* lod = getenv("HEATRAY_TEXTURE_LOD");
        m_textureLodCone = lod && std::string(lod) == "cone";
    }
    m_width = renderWidth;
    m_height = renderHeight;
    if (!HRFunc(hr_frame_resize(m_context, renderWidth, renderHeight))) return false;
    if (!generateRandomSequences(m_renderOptions.maxRenderPasses, m_renderOptions.sampleMode, m_renderOptions.bokehShape)) return false;
    // generateSequenceOffsets(W, H): sobol(W*H points, sequence 0), on the device (PassGenerator.cpp:150-159)
    if (!HRFunc(hr_seq_offsets_generate(m_context))) return false;

    m_resultPixels = openrl::PixelPackBuffer::create(renderWidth * renderHeight * (RLint)sizeof(float) * openrl::PixelPackBuffer::kNumChannels);

    m_scene = Scene::create();
    m_environmentLight = m_scene->lighting()->addEnvironmentLight();

    // The block pixel offsets of interactive rendering (PassGenerator.cpp:267-294 of the reference): the list of (row, col) pairs of
    // a block, shuffled "so that there is some inherit randomness to make the visualization less regular", handed to the ray
    // core as a table instead of an RL texture.  HEATRAY_BLOCK_SHUFFLE_SEED makes the shuffle reproducible (tests).
    {
        std::vector<int32_t> coords;
        for (int row = 0; row < RenderOptions::kInteractiveBlockSize.x; ++row) {
            for (int col = 0; col < RenderOptions::kInteractiveBlockSize.y; ++col) {
                coords.push_back(row);
                coords.push_back(col);
            }
        }
        std::vector<int> order(coords.size() / 2);
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        const char* seed = getenv("HEATRAY_BLOCK_SHUFFLE_SEED");
        std::mt19937 generator(seed ? (unsigned)atoi(seed) : std::random_device()());
        std::shuffle(order.begin(), order.end(), generator);
        std::vector<int32_t> shuffled(coords.size());
        for (size_t i = 0; i < order.size(); ++i) {
            shuffled[2 * i] = coords[2 * order[i]];
            shuffled[2 * i + 1] = coords[2 * order[i] + 1];
        }
        if (!HRFunc(hr_interactive_blocks_set(m_context, shuffled.data(), RenderOptions::kInteractiveBlockSize.x, RenderOptions::kInteractiveBlockSize.y))) return false;
    }
    return true;
}

// PassGenerator.cpp:301-323 of the reference.
void PassGenerator::runResizeJob(const RLint newRenderWidth, const RLint newRenderHeight)
{
    if (m_resultPixels && m_resultPixels->mapped()) {
        m_resultPixels->unmapPixelData();
    }
    m_width = newRenderWidth;
    m_height = newRenderHeight;
    HRFunc(hr_frame_resize(m_context, newRenderWidth, newRenderHeight));
    HRFunc(hr_seq_offsets_generate(m_context));
    m_resultPixels = openrl::PixelPackBuffer::create(newRenderWidth * newRenderHeight * (RLint)sizeof(float) * openrl::PixelPackBuffer::kNumChannels);
    m_renderOptions.resetInternalState = true;
}

namespace {
int visualizerModeOf(PassGenerator::RenderOptions::DebugVisualizationMode mode)
{
    using M = PassGenerator::RenderOptions::DebugVisualizationMode;
    switch (mode) {
        case M::kGeometricNormals:   return HR_VIS_GEOMETRIC_NORMALS;
        case M::kUVs:                return HR_VIS_UVS;
        case M::kTangents:           return HR_VIS_TANGENTS;
        case M::kBitangents:         return HR_VIS_BITANGENTS;
        case M::kNormalmap:          return HR_VIS_NORMALMAP;
        case M::kFinalNormals:       return HR_VIS_FINAL_NORMALS;
        case M::kBaseColor:          return HR_VIS_BASE_COLOR;
        case M::kRoughness:          return HR_VIS_ROUGHNESS;
        case M::kMetallic:           return HR_VIS_METALLIC;
        case M::kEmissive:           return HR_VIS_EMISSIVE;
        case M::kClearcoat:          return HR_VIS_CLEARCOAT;
        case M::kClearcoatRoughness: return HR_VIS_CLEARCOAT_ROUGHNESS;
        case M::kClearcoatNormalmap: return HR_VIS_CLEARCOAT_NORMALMAP;
        case M::kShader:             return HR_VIS_SHADER;
        default:                     return HR_VIS_NONE;
    }
}
} // namespace

// PassGenerator.cpp:325-401 of the reference: the per-pass driver.
void PassGenerator::runRenderFrameJob(const RenderOptions& newOptions, const PassCompleteCallback& passCompleteCallback)
{
    const auto start = std::chrono::steady_clock::now();

    // The callback of the previous pass may have left the pixels mapped (:331-333).
    if (m_resultPixels->mapped()) {
        m_resultPixels->unmapPixelData();
    }

    if ((newOptions.enableInteractiveMode != m_renderOptions.enableInteractiveMode) ||
        (newOptions.enableOfflineMode != m_renderOptions.enableOfflineMode) ||
        (newOptions.resetInternalState)) {
        resetRenderingState(newOptions);
    }

    // Geometry edits since the last pass: one BVH rebuild on the device.
    if (m_scene->geometryDirty()) {
        m_scene->commit();
    }

    // 35 mm film (36 x 24): vertical field of view from the focal length (:341-343).
    const float fovY = 2.0f * std::atan2(24.0f, 2.0f * m_renderOptions.camera.focalLength);

    bool jobCompleted = false;
    do {
        hr_pass_params params;
        memset(&params, 0, sizeof(params));
        params.sample_index = (int32_t)m_currentSampleIndex;
        params.max_ray_depth = m_globalMaxRayDepth;
        params.max_channel_value = m_globalMaxChannelValue;
        params.fov_tan = std::tan(fovY * 0.5f);
        params.aspect_ratio = m_renderOptions.camera.aspectRatio;
        params.focus_distance = m_renderOptions.camera.focusDistance;
        params.aperture_radius = m_renderOptions.camera.apertureRadius;
        memcpy(params.view_matrix, &m_renderOptions.camera.viewMatrix[0][0], sizeof(params.view_matrix));
        params.interactive_mode = m_renderOptions.enableInteractiveMode ? 1 : 0;
        params.block_size[0] = RenderOptions::kInteractiveBlockSize.x;
        params.block_size[1] = RenderOptions::kInteractiveBlockSize.y;
        params.current_block_pixel[0] = m_currentBlockPixelSample.x;
        params.current_block_pixel[1] = m_currentBlockPixelSample.y;
        params.max_sample_index = float(m_renderOptions.maxRenderPasses);
        using M = RenderOptions::DebugVisualizationMode;
        params.visualizer_mode = visualizerModeOf(m_globalDebugMode);
        params.enable_visualizer = params.visualizer_mode != HR_VIS_NONE ? 1 : 0;
        params.show_nans = m_globalDebugMode == M::kNANs ? 1 : 0;
        params.show_inf = m_globalDebugMode == M::kInf ? 1 : 0;
        params.enable_accumulator_visualizer = (params.show_nans || params.show_inf) ? 1 : 0;
        // RenderOptions is the reference's struct, unchanged, so the estimator is chosen out of band: HEATRAY_ESTIMATOR=env_mis selects
        // the importance-sampled environment + MIS estimator of include/hrcore.h (default: the reference's estimator)
        params.estimator = m_allLights ? HR_ESTIMATOR_ALL_LIGHTS : (m_envMis ? HR_ESTIMATOR_ENV_MIS : HR_ESTIMATOR_REFERENCE);
        params.texture_lod = m_textureLodCone ? HR_TEXTURE_LOD_CONE : HR_TEXTURE_LOD_BASE; // HEATRAY_TEXTURE_LOD=cone: mip chain + ray cones

        // Interactive mode walks the 3x3 block; the sample index advances once per full block (:372-384).
        if (m_renderOptions.enableInteractiveMode) {
            m_currentBlockPixelSample.x += 1;
            if (m_currentBlockPixelSample.x == RenderOptions::kInteractiveBlockSize.x) {
                m_currentBlockPixelSample.x = 0;
                m_currentBlockPixelSample.y += 1;
                if (m_currentBlockPixelSample.y == RenderOptions::kInteractiveBlockSize.y) {
                    m_currentBlockPixelSample = glm::ivec2(0, 0);
                    ++m_currentSampleIndex;
                }
            }
        } else {
            ++m_currentSampleIndex;
        }

        HRFunc(hr_render_pass(m_context, &params)); // rlRenderFrame() (:386)

        // Offline mode keeps rendering inside this job and reports pixels only after the last pass (:390-394);
        // the passes of one offline job overlap in libhrcore's pass pipeline.
        jobCompleted = !(m_renderOptions.enableOfflineMode && (m_currentSampleIndex < m_renderOptions.maxRenderPasses));
        if (jobCompleted) {
            // The last pass of a render (and every offline job) delivers the complete image; the passes in between are
            // displayed progressively from what is complete already, so libhrcore's pass pipeline stays full.
            const bool lastPass = m_currentSampleIndex >= m_renderOptions.maxRenderPasses;
            m_resultPixels->setPixelData(lastPass || m_renderOptions.enableOfflineMode);
        }

        const float passTime = std::chrono::duration<float>(std::chrono::steady_clock::now() - start).count();
        if (passCompleteCallback) {
            passCompleteCallback(jobCompleted, m_resultPixels, passTime, m_currentSampleIndex);
        }
    } while (!jobCompleted);
}

void PassGenerator::runDestroyJob()
{
    m_environmentLight.reset();
    m_scene.reset();
    if (m_resultPixels && m_resultPixels->mapped()) {
        m_resultPixels->unmapPixelData();
    }
    m_resultPixels.reset();
    if (m_context) {
        hr_ctx_destroy(m_context);
        m_context = nullptr;
        openrl::currentContext() = nullptr;
    }
}

// PassGenerator.cpp:435-577 of the reference: clear and re-read whatever changed.
void PassGenerator::resetRenderingState(const RenderOptions& newOptions)
{
    m_currentSampleIndex = 0;
    m_currentBlockPixelSample = glm::ivec2(0, 0);
    HRFunc(hr_clear(m_context)); // rlClear(RL_COLOR_BUFFER_BIT)

    if ((m_renderOptions.environment.map != newOptions.environment.map) ||
        (m_renderOptions.environment.exposureCompensation != newOptions.environment.exposureCompensation) ||
        (m_renderOptions.environment.thetaRotation != newOptions.environment.thetaRotation) ||
        (m_renderOptions.environment.solidColor != newOptions.environment.solidColor)) {
        changeEnvironment(newOptions.environment);
    }

    if (m_renderOptions.sampleMode != newOptions.sampleMode ||
        m_renderOptions.maxRenderPasses != newOptions.maxRenderPasses ||
        m_renderOptions.bokehShape != newOptions.bokehShape) {
        generateRandomSequences(newOptions.maxRenderPasses, newOptions.sampleMode, newOptions.bokehShape);
    }

    // The Globals block is only touched when an option CHANGES (:457-473); the untouched initial depth is 5.
    if (m_renderOptions.maxRayDepth != newOptions.maxRayDepth) {
        m_globalMaxRayDepth = (int)newOptions.maxRayDepth;
    }
    if (m_renderOptions.maxChannelValue != newOptions.maxChannelValue) {
        m_globalMaxChannelValue = newOptions.maxChannelValue;
    }
    if (m_renderOptions.debugVisMode != newOptions.debugVisMode) {
        m_globalDebugMode = newOptions.debugVisMode;
    }

    if (newOptions.debugPassRendering) {
        m_currentSampleIndex = newOptions.debugPassIndex;
    }

    m_renderOptions = newOptions;
    m_renderOptions.resetInternalState = false;
}

// PassGenerator.cpp:579-601 of the reference.
void PassGenerator::changeEnvironment(const RenderOptions::Environment &newEnv)
{
    if (!m_environmentLight) {
        m_environmentLight = m_scene->lighting()->addEnvironmentLight();
    }

    m_environmentLight->rotate(newEnv.thetaRotation);
    m_environmentLight->setExposure(newEnv.exposureCompensation);

    if (newEnv.map == EnvironmentLight::SOLID_COLOR) {
        m_environmentLight->enableSolidColor(newEnv.solidColor);
    } else if (newEnv.map == "<none>") {
        m_scene->lighting()->removeEnvironmentLight();
        m_environmentLight = nullptr;
    } else if (!newEnv.map.empty()) {
        m_environmentLight->changeImageSource(newEnv.map.c_str(), newEnv.builtInMap);
    }

    if (m_environmentLight) {
        m_scene->lighting()->updateLight(m_environmentLight);
    }
}

// PassGenerator.cpp:603-684 of the reference: 16 sequences x sampleCount points plus the aperture table, every sample mode and bokeh
// shape made by HIP kernels behind hr_sequences_generate.  The tables the reference builds on <random> (kRandom and the polygonal apertures)
// are libstdc++'s there (include/hrcore.h says which algorithms); an application that must keep another standard library's tables defines
// HR_HOST_TABLES_FROM_RANDOM_H and gets them from its own Utility/Random.h, uploaded with hr_sequences_set.
bool PassGenerator::generateRandomSequences(const RLint sampleCount, RenderOptions::SampleMode sampleMode, RenderOptions::BokehShape bokehShape)
{
#if defined(HR_HOST_TABLES_FROM_RANDOM_H) && defined(HR_HOST_HAS_RANDOM_H)
    std::vector<glm::vec2> values((size_t)kNumRandomSequences * sampleCount), aperture(values.size());
    for (unsigned int iSequence = 0; iSequence < (unsigned int)kNumRandomSequences; ++iSequence) {
        glm::vec2* seq = &values[(size_t)iSequence * sampleCount];
        switch (sampleMode) {
            case RenderOptions::SampleMode::kRandom:     util::uniformRandomFloats<glm::vec2>(seq, sampleCount, iSequence, 0.0f, 1.0f); break;
            case RenderOptions::SampleMode::kHalton:     util::halton(seq, sampleCount, iSequence); break;
            case RenderOptions::SampleMode::kHammersley: util::hammersley(seq, sampleCount, iSequence); break;
            case RenderOptions::SampleMode::kBlueNoise:  util::blueNoise(seq, sampleCount, iSequence); break;
            case RenderOptions::SampleMode::kSobol:      util::sobol(seq, sampleCount, iSequence); break;
        }
        glm::vec2* ap = &aperture[(size_t)iSequence * sampleCount];
        switch (bokehShape) {
            case RenderOptions::BokehShape::kCircular: util::radialSobol(ap, sampleCount, iSequence); break;
            case RenderOptions::BokehShape::kPentagon: util::randomPolygonal(ap, 5, sampleCount, iSequence); break;
            case RenderOptions::BokehShape::kHexagon:  util::randomPolygonal(ap, 6, sampleCount, iSequence); break;
            case RenderOptions::BokehShape::kOctagon:  util::randomPolygonal(ap, 8, sampleCount, iSequence); break;
        }
    }
    return HRFunc(hr_sequences_set(m_context, &values[0].x, &aperture[0].x, kNumRandomSequences, sampleCount));
#else
    int mode = HR_SAMPLE_SOBOL, shape = HR_BOKEH_CIRCULAR;
    switch (sampleMode) {
        case RenderOptions::SampleMode::kRandom:     mode = HR_SAMPLE_RANDOM; break;
        case RenderOptions::SampleMode::kHalton:     mode = HR_SAMPLE_HALTON; break;
        case RenderOptions::SampleMode::kHammersley: mode = HR_SAMPLE_HAMMERSLEY; break;
        case RenderOptions::SampleMode::kBlueNoise:  mode = HR_SAMPLE_BLUE_NOISE; break;
        case RenderOptions::SampleMode::kSobol:      mode = HR_SAMPLE_SOBOL; break;
    }
    switch (bokehShape) {
        case RenderOptions::BokehShape::kCircular: shape = HR_BOKEH_CIRCULAR; break;
        case RenderOptions::BokehShape::kPentagon: shape = HR_BOKEH_PENTAGON; break;
        case RenderOptions::BokehShape::kHexagon:  shape = HR_BOKEH_HEXAGON; break;
        case RenderOptions::BokehShape::kOctagon:  shape = HR_BOKEH_OCTAGON; break;
    }
    return HRFunc(hr_sequences_generate(m_context, mode, shape, sampleCount));
#endif
}
