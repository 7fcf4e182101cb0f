//
//  MultiScatterUtil.h
//  heatray_amd host layer
//
//  The specular multiscatter LUT (128x128 R32F, (1-E)/E over NdotV x roughness).  The reference integrates
//  it on the CPU and ships the result as a TIFF (/root/reference/Source/HeatrayRenderer/Materials/
//  MultiScatterUtil.cpp:91-139, 141-150); here the same integral runs as a HIP kernel whenever it is needed.
//

#pragma once

#include <RLWrapper/Texture.h>

#include <memory>

// Regenerate the LUT (the viewer's developer button).  The next material build picks it up.
void generateMultiScatterTexture();

// The shared LUT texture; generated on the device on first use.
std::shared_ptr<openrl::Texture> loadMultiscatterTexture();
