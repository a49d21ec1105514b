// Optional per-kernel timing with HIP events on the launch stream (off by default, zero cost
// when off). bench.py uses it to time the dominant kernel inside the timed region; the
// numbers must agree with rocprofv3 --kernel-trace. Not part of the reference API.
#pragma once
#include <hip/hip_runtime.h>

namespace lfs {
int prof_begin(const char* name, hipStream_t s); // returns a token (-1 when profiling is off)
void prof_end(int token, hipStream_t s);
// One-kernel scope without the two event-record packets: when profiling is on (and the filter lets `name` through) hands out a pair of events for
// hipExtLaunchKernelGGL, which stamps them from the dispatch packet's own completion signal - the kernel's start and end, nothing queued in front of or behind it.
// (Two hipEventRecord around a launch are two barrier packets: 5.6 us each around raster_bwd in the trace of the driver's command, profiles/r06/lease31.) false: launch as usual.
bool prof_kernel_events(const char* name, hipEvent_t* start, hipEvent_t* stop);
struct ProfScope {
    int tok; hipStream_t s;
    ProfScope(const char* name, hipStream_t st) : tok(prof_begin(name, st)), s(st) {}
    ~ProfScope() { if (tok >= 0) prof_end(tok, s); }
};
} // namespace lfs
