//
//  Scene.h
//  heatray_amd host layer
//
//  Geometry, materials and lighting of the scene being rendered.  Public API of
//  /root/reference/Source/HeatrayRenderer/Scene/Scene.h:27-85.  Geometry edits are batched: they mark the
//  scene dirty and PassGenerator commits (rebuilds the BVH on the device) before the next pass.
//

#pragma once

#include "Lighting.h"
#include "Mesh.h"

#include <Utility/AABB.h>

#include <glm/glm/mat4x4.hpp>

#include <functional>
#include <memory>
#include <string_view>
#include <vector>

namespace openrl {
class Program; // kept so that callers written against the OpenRL-era signature still compile
} // namespace openrl
class Material;
class MeshProvider;

class Scene
{
public:
    static std::shared_ptr<Scene> create();
    ~Scene() { clearAll(); }

    // There are no shader programs in this implementation; the callback is accepted and never invoked.
    using NewProgramCreatedCallback = std::function<void(const std::shared_ptr<openrl::Program>)>;
    void installNewProgramCreatedCallback(NewProgramCreatedCallback &&callback) { m_newProgramCreatedCallback = std::move(callback); }

    // Load a model through the application's Assimp provider (only in builds that have it).
    void loadFromDisk(const std::string_view path, bool convertToMeters);

    // Submit a provider's geometry; returns the index of the new mesh in meshes().
    size_t addMesh(MeshProvider *meshProvider, std::vector<std::shared_ptr<Material>> &&materials, const glm::mat4 &transform);

    void removeMesh(size_t meshIndex);

    // Apply a transform on top of every submesh's own transform.
    void applyTransform(const glm::mat4 &transform);

    void clearMeshesAndMaterials();
    void clearLighting() { m_lighting->clear(); }
    void clearAll() {
        clearMeshesAndMaterials();
        clearLighting();
    }

    std::shared_ptr<Lighting> lighting() { return m_lighting; }
    const std::vector<Mesh> &meshes() { return m_meshes; }

    const util::AABB &aabb() const { return m_aabb; }

    // Geometry changed since the last commit (PassGenerator commits before rendering).
    bool geometryDirty() const { return m_geometryDirty; }
    // hr_scene_commit: world transform, LBVH build, 4-wide collapse — all on the device.
    void commit();

private:
    Scene() {
        m_lighting = std::shared_ptr<Lighting>(new Lighting);
    }

    std::vector<Mesh> m_meshes;
    std::shared_ptr<Lighting> m_lighting = nullptr;

    NewProgramCreatedCallback m_newProgramCreatedCallback;

    util::AABB m_aabb;
    bool m_geometryDirty = true;
};
