// Standalone builds only: the geometry hand-over interface Scene::addMesh consumes.
// In the Heatray tree the application's own Scene/MeshProvider.h (next to Scene.h) is found first and this
// file is never seen; it exists so that the drop-in layer and its tests build without the application.
// Type, member and method names are the application's (planar vertex buffers addressed by usage, 32-bit
// indices, one submesh per draw call); only named access is relied upon.
#pragma once

#include <glm/glm/mat4x4.hpp>

#include <stddef.h>
#include <stdint.h>
#include <string>
#include <string_view>

enum class DrawMode { Triangles, TriangleStrip };

enum VertexAttributeUsage { // which planar stream an attribute describes
    VertexAttributeUsage_Position, VertexAttributeUsage_Normal, VertexAttributeUsage_TexCoord,
    VertexAttributeUsage_Tangents, VertexAttributeUsage_Bitangents, VertexAttributeUsage_Colors,
    VertexAttributeUsageCount,
};

struct VertexAttribute {
    VertexAttributeUsage usage = VertexAttributeUsage_Position;
    int buffer = -1;                 // which vertex buffer of the provider
    int componentCount = 0, size = 0; // floats per vertex, bytes per component
    int stride = 0;                  // bytes between elements
    size_t offset = 0;               // byte offset of the first element
};

class MeshProvider
{
public:
    struct Submesh {
        std::string name;
        glm::mat4 localTransform = glm::mat4(1.0f);
        DrawMode drawMode = DrawMode::Triangles;
        int materialIndex = -1, vertexAttributeCount = 0;
        size_t indexBuffer = 0, indexOffset = 0, elementCount = 0;
        VertexAttribute vertexAttributes[VertexAttributeUsageCount];
    };

    virtual ~MeshProvider() {}
    explicit MeshProvider(const std::string_view name) : m_name(name) {}
    const std::string_view name() { return m_name; }

    // buffer counts, sizes in bytes, and fills into caller memory; then the draw calls
    virtual size_t GetVertexBufferCount() = 0;
    virtual size_t GetIndexBufferCount() = 0;
    virtual size_t GetSubmeshCount() = 0;
    virtual size_t GetVertexBufferSize(size_t bufferIndex) = 0;
    virtual size_t GetIndexBufferSize(size_t bufferIndex) = 0;
    virtual void FillVertexBuffer(size_t bufferIndex, uint8_t* buffer) = 0;
    virtual void FillIndexBuffer(size_t bufferIndex, uint8_t* buffer) = 0;
    virtual Submesh GetSubmesh(size_t submeshIndex) = 0;

protected:
    const std::string m_name;
};
