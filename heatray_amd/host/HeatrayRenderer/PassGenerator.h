//
//  PassGenerator.h
//  heatray_amd host layer
//
//  Produces one pass (one sample per pixel) of the path-traced image per request, on a worker thread, and
//  hands the accumulated pixels to a completion callback.  Public API — init / destroy / resize /
//  renderPass / loadScene / changeLighting / modifyScene / runOpenRLTask, RenderOptions and the callback
//  signatures — as in /root/reference/Source/HeatrayRenderer/PassGenerator.h:33-193, so that the viewer
//  compiles against it unchanged.  The OpenRL context, framebuffer and frame shader behind it are
//  replaced by one libhrcore context (HIP kernels on the MI355X).
//

#pragma once

#include <RLWrapper/RLTypes.h>

#include <glm/glm/mat4x4.hpp>
#include <glm/glm/ext/scalar_constants.hpp>

#include <condition_variable>
#include <deque>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace openrl {
    class PixelPackBuffer;
    class Texture;
} // namespace openrl.
class EnvironmentLight;
class Lighting;
class Scene;
struct hr_ctx;

class PassGenerator
{
public:
    PassGenerator() = default;
    ~PassGenerator();

    // Start the worker thread and create the device context.  Sizes are in pixels.
    void init(const RLint renderWidth, const RLint renderHeight);

    //-------------------------------------------------------------------------
    // Everything that configures a pass.  Passed by value with every renderPass() call.
    struct RenderOptions {
        // Render one pixel of every 3x3 block per call (faster feedback, slower convergence).
        bool enableInteractiveMode = true;

        // Run all passes inside one request and report pixels only at the end.
        bool enableOfflineMode = false;

        static constexpr glm::ivec2 kInteractiveBlockSize = glm::ivec2(3, 3);

        bool resetInternalState = true;
        uint32_t maxRenderPasses = 32;
        uint32_t maxRayDepth = 10;
        float maxChannelValue = glm::pi<float>();

        std::string scene;

        struct Environment {
            std::string map;
            glm::vec3 solidColor = glm::vec3(0.5f);
            bool builtInMap = true;
            float exposureCompensation = 0.0f;
            float thetaRotation = 0.0f;
        } environment;

        struct Camera {
            static constexpr size_t NUM_FSTOPS = 12;
            static constexpr float fstopOptions[NUM_FSTOPS] = {
                std::numeric_limits<float>::max(), 32.0f, 22.0f, 16.0f, 11.0f, 8.0f, 5.6f, 4.0f, 2.8f, 2.0f, 1.4f, 1.0f
            };

            float aspectRatio    = -1.0f;  // width / height
            float focusDistance  = 1.0f;   // meters
            float focalLength    = 50.0f;  // millimeters
            float apertureRadius = 0.0f;   // meters; derived, see setApertureRadius()
            float fstop          = fstopOptions[1];
            glm::mat4 viewMatrix = glm::mat4(1.0f);

            void setApertureRadius() {
                apertureRadius = (focalLength / fstop) / 1000.0f;
            }
        } camera;

        enum class SampleMode {
            kRandom,
            kHalton,
            kHammersley,
            kBlueNoise,
            kSobol
        };

        SampleMode sampleMode = SampleMode::kSobol;

        enum class BokehShape {
            kCircular,
            kPentagon,
            kHexagon,
            kOctagon
        };

        BokehShape bokehShape = BokehShape::kCircular;

        enum class DebugVisualizationMode {
            kNone,
            kGeometricNormals,
            kUVs,
            kTangents,
            kBitangents,
            kNormalmap,
            kFinalNormals,
            kBaseColor,
            kRoughness,
            kMetallic,
            kEmissive,
            kClearcoat,
            kClearcoatRoughness,
            kClearcoatNormalmap,
            kShader,
            kNANs,
            kInf
        };

        DebugVisualizationMode debugVisMode = DebugVisualizationMode::kNone;

        // Render only the pass with index debugPassIndex.
        bool debugPassRendering = false;
        int debugPassIndex = 0;
    };

    // Stop the worker thread and release the device context.
    void destroy();

    //-------------------------------------------------------------------------
    // Queue one pass.  The callback runs on the worker thread when the pass is done.
    using PassCompleteCallback = std::function<void(bool frameDataAvailable, std::shared_ptr<openrl::PixelPackBuffer> resultPixels, float passTime, size_t passIndex)>;
    void renderPass(const RenderOptions& newOptions, PassCompleteCallback callback);

    void resize(const RLint newWidth, const RLint newHeight);

    using LoadSceneCallback = std::function<void(std::shared_ptr<Scene> scene)>;
    void loadScene(LoadSceneCallback callback, bool clearOldScene = true);

    using LightingCallback = std::function<void(std::shared_ptr<Lighting> lighting)>;
    void changeLighting(LightingCallback callback);

    using ModifySceneCallback = std::function<void(std::shared_ptr<Scene> scene)>;
    void modifyScene(ModifySceneCallback callback);

    // Run an arbitrary task on the worker thread (the thread that owns the device context).
    using OpenRLTask = std::function<void()>;
    void runOpenRLTask(OpenRLTask task);

    std::shared_ptr<Scene> scene() const { return m_scene; }

    static constexpr RLint kNumRandomSequences = 16;

    // Block until every queued job has run (headless callers and tests; the viewer never needs it).
    void waitIdle();

private:
    PassGenerator(const PassGenerator& other) = delete;
    PassGenerator(const PassGenerator&& other) = delete;
    PassGenerator& operator=(const PassGenerator& other) = delete;
    PassGenerator& operator=(const PassGenerator&& other) = delete;

    void changeEnvironment(const RenderOptions::Environment& newEnv);
    bool generateRandomSequences(const RLint sampleCount, RenderOptions::SampleMode sampleMode, RenderOptions::BokehShape bokehShape);
    void resetRenderingState(const RenderOptions& newOptions);

    bool runInitJob(const RLint renderWidth, const RLint renderHeight);
    void runResizeJob(const RLint newRenderWidth, const RLint newRenderHeight);
    void runRenderFrameJob(const RenderOptions& newOptions, const PassCompleteCallback& passCompleteCallback);
    void runDestroyJob();

    // ---- worker thread: a FIFO of closures; a closure returning true ends the thread.
    using Job = std::function<bool()>;
    void enqueue(Job job);
    void workerLoop();
    std::thread m_worker;
    std::mutex m_queueMutex;
    std::condition_variable m_queueSignal;
    std::condition_variable m_idleSignal;
    std::deque<Job> m_jobs;
    bool m_busy = false;
    bool m_running = false;

    hr_ctx* m_context = nullptr;
    bool m_envMis = false; // HEATRAY_ESTIMATOR=env_mis
    bool m_allLights = false; // HEATRAY_ESTIMATOR=all_lights
    bool m_textureLodCone = false; // HEATRAY_TEXTURE_LOD=cone
    RLint m_width = 0, m_height = 0;

    std::shared_ptr<openrl::PixelPackBuffer> m_resultPixels = nullptr;
    std::shared_ptr<EnvironmentLight> m_environmentLight = nullptr;

    unsigned int m_currentSampleIndex = 0;

    RenderOptions m_renderOptions;
    glm::ivec2 m_currentBlockPixelSample = glm::ivec2(0, 0);

    // The state of the reference's Globals uniform block that survives between passes
    // (PassGenerator.h:269-294): note the initial depth of 5.
    int m_globalMaxRayDepth = 5;
    float m_globalMaxChannelValue = glm::pi<float>();
    RenderOptions::DebugVisualizationMode m_globalDebugMode = RenderOptions::DebugVisualizationMode::kNone;

    std::shared_ptr<Scene> m_scene = nullptr;
};
