//
//  HrContext.h
//  heatray_amd host layer
//
//  The "current context" of the drop-in layer.  OpenRL keeps one implicit current context per
//  thread (OpenRLSetCurrentContext, /root/reference/Source/HeatrayRenderer/PassGenerator.cpp:164-165);
//  every RLWrapper object uses it without naming it.  Here PassGenerator creates the hr_ctx on
//  its worker thread and publishes it through this accessor, so that openrl::Texture,
//  materials, lights and meshes reach libhrcore the same way.
//

#pragma once

#include <hrcore.h>

#include <Utility/Log.h>

#include <assert.h>
#include <stdio.h>

namespace openrl {

// The context every wrapper object talks to (one per process, owned by PassGenerator).
inline hr_ctx*& currentContext()
{
    static hr_ctx* ctx = nullptr;
    return ctx;
}

// Counterpart of RLFunc()/checkError (/root/reference/Source/RLWrapper/Error.h:18-41):
// log the library's message and assert in debug builds.
inline bool checkStatus(int status, const char* call)
{
    if (status != HR_OK) {
        // (the application's logger when one is installed — the viewer's ImGuiLog — else stderr: errors are never silent)
        if (util::Log::instance())
            LOG_ERROR("libhrcore error %d in %s: %s", status, call, hr_last_error(currentContext()));
        else
            fprintf(stderr, "libhrcore error %d in %s: %s\n", status, call, hr_last_error(currentContext()));
        assert(0 && "libhrcore call failed");
        return false;
    }
    return true;
}

} // namespace openrl

#define HRFunc(call) ::openrl::checkStatus((call), #call)
