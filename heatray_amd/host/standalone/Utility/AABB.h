// Standalone builds only: the scene bounding box type the drop-in layer exposes through Scene::aabb().
// In the Heatray tree the application's own Utility/AABB.h is used instead (this directory is not on its include path).
#pragma once

#include <glm/glm/glm.hpp>

#include <limits>

namespace util {

struct AABB {
    AABB() : min(std::numeric_limits<float>::max()), max(-std::numeric_limits<float>::max()) {}
    void expand(const glm::vec3& v)
    {
        min = glm::min(min, v);
        max = glm::max(max, v);
    }
    glm::vec3 center() const { return (min + max) * 0.5f; }
    float radius() const { return glm::length(max - min); }
    bool valid() const { return min.x < max.x || min.y < max.y || min.z < max.z; }

    glm::vec3 min;
    glm::vec3 max;
    glm::mat4 transform = glm::mat4(1.0f);
};

} // namespace util
