// Material.h (heatray_amd host layer)
// Interface the viewer and the mesh loader program against: a named material with three life-cycle calls.  Source-compatible
// with /root/reference/Source/HeatrayRenderer/Materials/Material.h:19-62; here a material is one row of libhrcore's
// material table (hr_material_set) instead of an RLSL program plus a uniform block.
#pragma once

#include <RLWrapper/HrContext.h>

#include <memory>
#include <string>
#include <string_view>

namespace openrl { class Texture; }
using TexturePtr = std::shared_ptr<openrl::Texture>; // every texture slot of the parameter structs

class Material
{
public:
    enum class Type { PBR, Glass };

    virtual ~Material() = default;
    explicit Material(const std::string_view name, Type type) : m_name(name), m_type(type) {}

    Type type() const { return m_type; }
    const std::string_view name() const { return m_name; }
    int tableIndex() const { return m_tableIndex; } // row in libhrcore's table; -1 until build()
    void enableVertexColors() { m_enableVertexColors = true; }

    virtual void build() = 0;   // allocate the row, upload the parameters
    virtual void rebuild() = 0; // start over (the reference recompiles its shader here)
    virtual void modify() = 0;  // re-upload after an edit

protected:
    static int allocateTableIndex() // ids are handed out once per process and never reused, like RL object names
    {
        static int next = 0;
        return next++;
    }

    const std::string m_name;
    Type m_type;
    int m_tableIndex = -1;
    bool m_enableVertexColors = false;
};
