//
//  Mesh.h
//  heatray_amd host layer
//
//  One loaded model: the geometry handles and materials of its submeshes.  Public surface of
//  /root/reference/Source/HeatrayRenderer/Scene/Mesh.h:29-71; each submesh is one libhrcore geometry
//  (hr_geom_add) instead of an RL primitive with vertex / index buffers.
//

#pragma once

#include <RLWrapper/RLTypes.h>

#include <glm/glm/mat4x4.hpp>

#include <functional>
#include <memory>
#include <vector>

class MeshProvider;
class Material;

class Mesh
{
public:
    Mesh() = delete;

    // Pull every buffer out of the provider and submit one geometry per submesh.  The scene is NOT
    // committed here: Scene::addMesh / PassGenerator commit once per batch of changes.
    Mesh(MeshProvider *meshProvider, std::vector<std::shared_ptr<Material>> &materials, const glm::mat4 &transform);
    ~Mesh() = default;
    Mesh(Mesh&&) = default;
    Mesh& operator=(Mesh&&) = default;

    // Remove the submitted geometry from libhrcore.
    void destroy();

    bool valid() const { return !m_submeshes.empty(); }

    const std::vector<std::shared_ptr<Material>>& materials() const { return m_materials;  }

    struct Submesh {
        int geometry = -1;             // hr_geom_id
        size_t elementCount = 0;
        size_t offset = 0;
        RLenum mode = 0;
        std::shared_ptr<Material> material = nullptr;
        glm::mat4 transform = glm::mat4(1.0f);
    };
    const std::vector<Submesh> &submeshes() const { return m_submeshes; }

private:
    std::vector<Submesh> m_submeshes;
    std::vector<std::shared_ptr<Material>> m_materials;
};
