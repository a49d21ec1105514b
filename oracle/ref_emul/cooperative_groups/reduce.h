// ORACLE / TEST INFRASTRUCTURE ONLY: cg::reduce lives in ../cooperative_groups.h
#pragma once
#include "../cooperative_groups.h"
