// ORACLE / TEST INFRASTRUCTURE ONLY: stands in for the CUDA-only torch header of the same name when the reference kernels are compiled as host code
#pragma once
#include "cuda_emul.h"
