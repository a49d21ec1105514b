// ORACLE / TEST INFRASTRUCTURE ONLY: see ../glm.hpp (the GLM subset the reference kernels need)
#pragma once
#include "../glm.hpp"
