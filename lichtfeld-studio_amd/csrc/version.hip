// Library identification. LFS_SRC_HASH = sha1 over every file under csrc/ and include/lfs_gsplat.h at build time (build.py): profiles/traffic.json is
// stamped with the hash of the library its counters were measured on, and bench.py reports roofline.traffic only while that is the library it runs.
#include "../../include/lfs_gsplat.h"
#ifndef LFS_SRC_HASH
#define LFS_SRC_HASH "unknown"
#endif
extern "C" const char* lfs_version(void) { return "lfs_gsplat gfx950 abi-2 src-" LFS_SRC_HASH; }
