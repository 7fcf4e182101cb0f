// Standalone builds only: the geometry hand-over interface Scene::addMesh consumes.
// In the Heatray tree the application's own Scene/MeshProvider.h (next to Scene.h) is found first and this
// file is never seen; it exists so that the drop-in layer and its tests build without the application.
// The member names and meanings are the application's (planar vertex buffers addressed by usage, 32-bit
// indices, one submesh per draw call).
#pragma once

#include <glm/glm/mat4x4.hpp>

#include <stddef.h>
#include <stdint.h>
#include <string>
#include <string_view>

enum class DrawMode { Triangles, TriangleStrip };

enum VertexAttributeUsage {
    VertexAttributeUsage_Position,
    VertexAttributeUsage_Normal,
    VertexAttributeUsage_TexCoord,
    VertexAttributeUsage_Tangents,
    VertexAttributeUsage_Bitangents,
    VertexAttributeUsage_Colors,
    VertexAttributeUsageCount,
};

struct VertexAttribute {
    VertexAttributeUsage usage = VertexAttributeUsage_Position;
    int buffer = -1;        // which vertex buffer of the provider
    int componentCount = 0; // floats per vertex
    int size = 0;           // bytes per component
    size_t offset = 0;      // byte offset of the first element
    int stride = 0;         // bytes between elements
};

class MeshProvider
{
public:
    struct Submesh {
        int vertexAttributeCount = 0;
        VertexAttribute vertexAttributes[VertexAttributeUsageCount];
        size_t indexBuffer = 0;
        size_t indexOffset = 0;
        size_t elementCount = 0;
        DrawMode drawMode = DrawMode::Triangles;
        int materialIndex = -1;
        glm::mat4 localTransform = glm::mat4(1.0f);
        std::string name;
    };

    explicit MeshProvider(const std::string_view name) : m_name(name) {}
    virtual ~MeshProvider() {}

    virtual size_t GetVertexBufferCount() = 0;
    virtual size_t GetVertexBufferSize(size_t bufferIndex) = 0; // bytes
    virtual void FillVertexBuffer(size_t bufferIndex, uint8_t* buffer) = 0;

    virtual size_t GetIndexBufferCount() = 0;
    virtual size_t GetIndexBufferSize(size_t bufferIndex) = 0; // bytes
    virtual void FillIndexBuffer(size_t bufferIndex, uint8_t* buffer) = 0;

    virtual size_t GetSubmeshCount() = 0;
    virtual Submesh GetSubmesh(size_t submeshIndex) = 0;

    const std::string_view name() { return m_name; }

protected:
    const std::string m_name;
};
