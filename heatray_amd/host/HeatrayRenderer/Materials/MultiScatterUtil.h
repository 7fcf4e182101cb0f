// MultiScatterUtil.h (heatray_amd host layer): the two entry points of
// /root/reference/Source/HeatrayRenderer/Materials/MultiScatterUtil.h; the LUT is integrated by libhrcore on the device.
#pragma once

#include <RLWrapper/Texture.h>

#include <memory>

std::shared_ptr<openrl::Texture> loadMultiscatterTexture(); // 128 x 128 energy-compensation table as a texture
void generateMultiScatterTexture();                          // re-integrate it (the reference writes a TIFF here)
