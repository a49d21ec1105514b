// Optional per-kernel timing with HIP events on the launch stream (off by default, zero cost
// when off). bench.py uses it to time the dominant kernel inside the timed region; the
// numbers must agree with rocprofv3 --kernel-trace. Not part of the reference API.
#pragma once
#include <hip/hip_runtime.h>

namespace lfs {
int prof_begin(const char* name, hipStream_t s); // returns a token (-1 when profiling is off)
void prof_end(int token, hipStream_t s);
struct ProfScope {
    int tok; hipStream_t s;
    ProfScope(const char* name, hipStream_t st) : tok(prof_begin(name, st)), s(st) {}
    ~ProfScope() { if (tok >= 0) prof_end(tok, s); }
};
} // namespace lfs
