// ORACLE/_ref - TEST INFRASTRUCTURE ONLY: see CUDAStream.h beside this file.
#pragma once
#include "CUDAStream.h"
