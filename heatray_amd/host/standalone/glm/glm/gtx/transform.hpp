// mini glm forwarding header (standalone builds only), see glm.hpp
#pragma once
#include "../glm.hpp"
