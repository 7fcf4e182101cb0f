// test aid: stands in for Source/HeatrayRenderer/Scene/AssimpMeshProvider.h when the viewer translation unit is syntax-checked in an
// image without the assimp submodule.  HeatrayRenderer.cpp includes that header but never names the class (Scene::loadFromDisk
// constructs it, and the loader is outside the drop-in boundary: INTEGRATION.md), so a forward declaration and the headers the real
// file pulls in for its includers are all the viewer needs from it.
#pragma once

#include "MeshProvider.h"

#include <memory>
#include <string>
#include <string_view>
#include <vector>

class Lighting;
class AssimpMeshProvider;
