// test aid (declaration only): stands in for Source/HeatrayRenderer/Scene/AssimpMeshProvider.h when the viewer translation unit is
// syntax-checked in an image without the assimp submodule.  HeatrayRenderer.cpp includes that header but never names the class
// (Scene::loadFromDisk constructs it); the loader itself is outside the drop-in boundary.
#pragma once

#include "MeshProvider.h"

#include <memory>
#include <string_view>

class Lighting;

class AssimpMeshProvider : public MeshProvider
{
public:
    explicit AssimpMeshProvider(const std::string_view filename, bool convertToMeters, std::shared_ptr<Lighting> lighting);
    size_t GetVertexBufferCount() override;
    size_t GetVertexBufferSize(size_t bufferIndex) override;
    void FillVertexBuffer(size_t bufferIndex, uint8_t *buffer) override;
    size_t GetIndexBufferCount() override;
    size_t GetIndexBufferSize(size_t bufferIndex) override;
    void FillIndexBuffer(size_t bufferIndex, uint8_t *buffer) override;
    size_t GetSubmeshCount() override;
    Submesh GetSubmesh(size_t submeshIndex) override;
};
