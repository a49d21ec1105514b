// ORACLE / TEST INFRASTRUCTURE ONLY: stands in for the CUDA-only torch header of the same name (IntersectTile.cu includes it for CUB_WRAPPER, whose call is in
// the launcher part that is not compiled here)
#pragma once
