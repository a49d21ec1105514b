// ORACLE / TEST INFRASTRUCTURE ONLY: stands in for the CUDA-only torch header of the same name (core/camera.cpp includes it for CUDAStreamGuard)
#pragma once
#include "CUDAStream.h"
