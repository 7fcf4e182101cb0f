//
//  Material.h
//  heatray_amd host layer
//
//  Base class of the materials the viewer and the scene loader create.  Same public surface as
//  /root/reference/Source/HeatrayRenderer/Materials/Material.h:14-63, minus the OpenRL program /
//  uniform-block handles: a material is one row of libhrcore's material table (hr_material_set),
//  and its shader permutation is the flag word of that row.
//

#pragma once

#include <RLWrapper/HrContext.h>

#include <memory>
#include <string>
#include <string_view>

class Material
{
public:
    enum class Type {
        PBR,
        Glass
    };

    explicit Material(const std::string_view name, Type type) : m_name(name), m_type(type) {}
    virtual ~Material() = default;

    const std::string_view name() const { return m_name; }
    Type type() const { return m_type; }

    // Allocate the table row and upload the parameters.
    virtual void build() = 0;
    // Throw the row's contents away and build again (the reference recompiles its shader here).
    virtual void rebuild() = 0;
    // Re-upload the parameters after an edit.
    virtual void modify() = 0;

    void enableVertexColors() { m_enableVertexColors = true; }

    // Row of this material in libhrcore's table; -1 until build().
    int tableIndex() const { return m_tableIndex; }

protected:
    // Material ids are handed out once per process and never reused, like RL object names.
    static int allocateTableIndex()
    {
        static int next = 0;
        return next++;
    }

    bool m_enableVertexColors = false;
    int m_tableIndex = -1;

    const std::string m_name;
    Type m_type;
};
