// the pieces of liblfs_gsplat the emulated rasterizer links against (tests/emul: host build of csrc/raster.hip)
#include <hip/hip_runtime.h>
namespace lfs {
int prof_begin(const char*, hipStream_t) { return -1; }
void prof_end(int, hipStream_t) {}
} // namespace lfs
// the profiler's C entry points (csrc/prof.hip is not part of an emulated build): nothing is ever recorded
extern "C" {
int lfs_profile_enable(int) { return 0; }
int lfs_profile_filter(const char*) { return 0; }
int lfs_profile_collect(int, char*, float*, int*) { return 0; }
}
