#include "Mesh.h"

#include "MeshProvider.h"

#include <HeatrayRenderer/Materials/Material.h>
#include <HeatrayRenderer/Materials/PhysicallyBasedMaterial.h>

#include <RLWrapper/HrContext.h>

#include <glm/glm/glm.hpp>

#include <stdio.h>
#include <string.h>

// What Mesh::Mesh of the reference does with OpenRL buffers and primitives
// (/root/reference/Source/HeatrayRenderer/Scene/Mesh.cpp:15-156), expressed as hr_mesh_desc submissions.
Mesh::Mesh(MeshProvider* meshProvider, std::vector<std::shared_ptr<Material>>& materials, const glm::mat4& transform)
{
    m_materials = std::move(materials);
    for (auto& material : m_materials) {
        material->build(); // allocates the material's table row
    }

    // The provider fills raw byte buffers; libhrcore copies what it needs during hr_geom_add.
    std::vector<std::vector<uint8_t>> vertexBuffers(meshProvider->GetVertexBufferCount());
    for (size_t ii = 0; ii < vertexBuffers.size(); ++ii) {
        vertexBuffers[ii].resize(meshProvider->GetVertexBufferSize(ii));
        meshProvider->FillVertexBuffer(ii, vertexBuffers[ii].data());
    }
    std::vector<std::vector<uint8_t>> indexBuffers(meshProvider->GetIndexBufferCount());
    for (size_t ii = 0; ii < indexBuffers.size(); ++ii) {
        const size_t bytes = meshProvider->GetIndexBufferSize(ii);
        if (bytes == 0) {
            fprintf(stderr, "Found a 0-sized index buffer - skipping.\n");
            continue;
        }
        indexBuffers[ii].resize(bytes);
        meshProvider->FillIndexBuffer(ii, indexBuffers[ii].data());
    }

    const size_t submeshCount = meshProvider->GetSubmeshCount();
    m_submeshes.resize(submeshCount);
    for (size_t ii = 0; ii < submeshCount; ++ii) {
        MeshProvider::Submesh submesh = meshProvider->GetSubmesh(ii);
        Mesh::Submesh& out = m_submeshes[ii];

        // Material lookup rule of Mesh.cpp:60-66.
        if (submesh.materialIndex != -1) {
            out.material = m_materials[submesh.materialIndex];
        } else {
            out.material = m_materials.size() > 1 ? m_materials[ii] : m_materials[0];
        }

        hr_mesh_desc desc;
        memset(&desc, 0, sizeof(desc));
        size_t vertexCount = 0;
        for (int jj = 0; jj < submesh.vertexAttributeCount; ++jj) {
            const VertexAttribute& attribute = submesh.vertexAttributes[jj];
            if (attribute.buffer < 0 || (size_t)attribute.buffer >= vertexBuffers.size()) continue;
            const std::vector<uint8_t>& buffer = vertexBuffers[attribute.buffer];
            const float* data = reinterpret_cast<const float*>(buffer.data() + attribute.offset);
            const int stride = attribute.stride ? attribute.stride : attribute.componentCount * (int)sizeof(float);
            switch (attribute.usage) {
                case VertexAttributeUsage_Position:
                    desc.positions = data;
                    desc.position_stride = stride;
                    vertexCount = stride ? (buffer.size() - attribute.offset) / stride : 0;
                    break;
                case VertexAttributeUsage_Normal:
                    desc.normals = data;
                    desc.normal_stride = stride;
                    break;
                case VertexAttributeUsage_TexCoord:
                    desc.uvs = data;
                    desc.uv_stride = stride;
                    break;
                case VertexAttributeUsage_Tangents:
                    desc.tangents = data;
                    desc.tangent_stride = stride;
                    break;
                case VertexAttributeUsage_Bitangents:
                    desc.bitangents = data;
                    desc.bitangent_stride = stride;
                    break;
                case VertexAttributeUsage_Colors:
                    desc.colors = data;
                    desc.color_stride = stride;
                    break;
                default:
                    fprintf(stderr, "Unknown vertex attribute usage %d for submesh %s\n", attribute.usage, submesh.name.c_str());
            }
        }
        desc.n_vertices = (int32_t)vertexCount;

        out.transform = submesh.localTransform * transform; // Mesh.cpp:85
        memcpy(desc.world_from_entity, &out.transform[0][0], sizeof(desc.world_from_entity));
        desc.front_face_cw = glm::determinant(out.transform) < 0.0f ? 1 : 0; // Mesh.cpp:86-91

        // Alpha-masked PBR surfaces must let occlusion rays run the alpha test (Mesh.cpp:95-100).
        desc.is_occluder = 1;
        if (out.material->type() == Material::Type::PBR) {
            auto pbr = std::static_pointer_cast<PhysicallyBasedMaterial>(out.material);
            if (pbr->parameters().alphaMask) {
                desc.is_occluder = 0;
            }
        }
        desc.material_id = out.material->tableIndex();

        switch (submesh.drawMode) {
            case DrawMode::Triangles:
                out.mode = RL_TRIANGLES;
                desc.mode = HR_TRIANGLES;
                break;
            case DrawMode::TriangleStrip:
                out.mode = RL_TRIANGLE_STRIP;
                desc.mode = HR_TRIANGLE_STRIP;
                break;
            default:
                fprintf(stderr, "Unsupported draw mode for submesh %s!\n", submesh.name.c_str());
                break;
        }
        out.elementCount = submesh.elementCount;
        out.offset = submesh.indexOffset;

        if (submesh.indexBuffer < indexBuffers.size() && !indexBuffers[submesh.indexBuffer].empty() && desc.positions && desc.normals) {
            const std::vector<uint8_t>& ib = indexBuffers[submesh.indexBuffer];
            desc.indices = reinterpret_cast<const uint32_t*>(ib.data() + submesh.indexOffset); // byte offset, like rlDrawElements
            desc.n_indices = (int32_t)submesh.elementCount;
            hr_geom_id id = -1;
            if (HRFunc(hr_geom_add(openrl::currentContext(), &desc, &id))) {
                out.geometry = id;
            }
        }
    }
}

void Mesh::destroy()
{
    for (Submesh& submesh : m_submeshes) {
        if (submesh.geometry >= 0 && openrl::currentContext()) {
            hr_geom_remove(openrl::currentContext(), submesh.geometry);
            submesh.geometry = -1;
        }
    }
    m_submeshes.clear();
    m_materials.clear();
}
