// Mesh.h (heatray_amd host layer)
// One loaded model: the geometry handles and materials of its submeshes.  Public surface of
// /root/reference/Source/HeatrayRenderer/Scene/Mesh.h:29-71; a submesh is one libhrcore geometry (hr_geom_add) instead of an
// RL primitive with vertex / index buffers.
#pragma once

#include <RLWrapper/RLTypes.h>

#include <glm/glm/mat4x4.hpp>

#include <functional>
#include <memory>
#include <vector>

class Material;
class MeshProvider;

class Mesh
{
public:
    struct Submesh {
        std::shared_ptr<Material> material;
        glm::mat4 transform = glm::mat4(1.0f);
        size_t elementCount = 0, offset = 0;
        RLenum mode = 0;
        int geometry = -1; // hr_geom_id
    };

    // Pulls every buffer out of the provider and submits one geometry per submesh.  The scene is NOT committed here:
    // Scene::addMesh / PassGenerator commit once per batch of changes.
    Mesh(MeshProvider *meshProvider, std::vector<std::shared_ptr<Material>> &materials, const glm::mat4 &transform);
    Mesh() = delete;
    Mesh(Mesh&&) = default;
    Mesh& operator=(Mesh&&) = default;
    ~Mesh() = default;

    void destroy(); // removes the submitted geometry from libhrcore
    bool valid() const { return !m_submeshes.empty(); }
    const std::vector<Submesh> &submeshes() const { return m_submeshes; }
    const std::vector<std::shared_ptr<Material>>& materials() const { return m_materials; }

private:
    std::vector<std::shared_ptr<Material>> m_materials;
    std::vector<Submesh> m_submeshes;
};
