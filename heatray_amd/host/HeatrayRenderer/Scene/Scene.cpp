#include "Scene.h"

#include "MeshProvider.h"

#if __has_include("AssimpMeshProvider.h")
#include "AssimpMeshProvider.h" // the application's loader (untouched reference code, needs assimp)
#define HR_HOST_HAS_ASSIMP 1
#endif

#include <HeatrayRenderer/Materials/Material.h>
#include <RLWrapper/HrContext.h>

#include <stdio.h>
#include <string.h>

std::shared_ptr<Scene> Scene::create()
{
    return std::shared_ptr<Scene>(new Scene());
}

// Scene.cpp:14-27 of the reference.
void Scene::loadFromDisk(const std::string_view path, bool convertToMeters)
{
#if defined(HR_HOST_HAS_ASSIMP)
    m_lighting->clearAllButEnvironment();

    AssimpMeshProvider provider(path, convertToMeters, m_lighting);
    m_aabb = provider.sceneAABB();

    auto &materials = provider.GetMaterials();
    m_meshes.emplace_back(&provider, materials, glm::mat4(1.0f));
    m_geometryDirty = true;
#else
    (void)convertToMeters;
    fprintf(stderr, "Scene::loadFromDisk(%.*s): this build has no Assimp provider\n", (int)path.size(), path.data());
#endif
}

size_t Scene::addMesh(MeshProvider *meshProvider, std::vector<std::shared_ptr<Material>>&& materials, const glm::mat4& transform)
{
    m_meshes.emplace_back(meshProvider, materials, transform);
    m_geometryDirty = true;
    return (m_meshes.size() - 1);
}

void Scene::removeMesh(size_t meshIndex)
{
    m_meshes[meshIndex].destroy();
    m_meshes.erase(m_meshes.begin() + meshIndex);
    m_geometryDirty = true;
}

// Scene.cpp:38-49 of the reference: worldFromEntity = transform * submesh.transform.
void Scene::applyTransform(const glm::mat4 &transform)
{
    for (auto &mesh : m_meshes) {
        for (auto &submesh : mesh.submeshes()) {
            if (submesh.geometry < 0) continue;
            const glm::mat4 newTransform = transform * submesh.transform;
            float m[16];
            memcpy(m, &newTransform[0][0], sizeof(m));
            HRFunc(hr_geom_set_transform(openrl::currentContext(), submesh.geometry, m));
        }
    }
    m_geometryDirty = true;
}

void Scene::clearMeshesAndMaterials()
{
    for (auto &mesh : m_meshes) {
        mesh.destroy();
    }
    m_meshes.clear();
    m_geometryDirty = true;
}

void Scene::commit()
{
    if (openrl::currentContext() && HRFunc(hr_scene_commit(openrl::currentContext()))) {
        m_geometryDirty = false;
    }
}
