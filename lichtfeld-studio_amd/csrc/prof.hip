// Event-based kernel timer behind lfs_profile_* (see lfs_prof.h / include/lfs_gsplat.h).
#include "lfs_prof.h"
#include "../../include/lfs_gsplat.h"
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace lfs {
namespace {
struct Pending { std::string name; hipEvent_t e0, e1; bool closed; };
std::mutex g_mu;
bool g_on = false;
std::string g_filter; // empty: every scope; otherwise only scopes with exactly this name
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool; // recycled events: creating two per scope costs more than recording them
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
} // namespace

int prof_begin(const char* name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || !name || !*name) return -1;   // ("" = a scope whose one kernel is timed through prof_kernel_events)
    if (!g_filter.empty() && g_filter != name) return -1;
    Pending p{name, get_event(), get_event(), false};
    if (!p.e0 || !p.e1) return -1;
    (void)hipEventRecord(p.e0, s);
    g_pending.push_back(p);
    return int(g_pending.size()) - 1;
}
bool prof_kernel_events(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return false;
    if (!g_filter.empty() && g_filter != name) return false;
    Pending p{name, get_event(), get_event(), true};   // (closed: the launch itself records both)
    if (!p.e0 || !p.e1) return false;
    g_pending.push_back(p);
    *start = p.e0; *stop = p.e1;
    return true;
}
void prof_end(int token, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < 0 || token >= int(g_pending.size())) return;
    (void)hipEventRecord(g_pending[token].e1, s);
    g_pending[token].closed = true;
}
} // namespace lfs

extern "C" int lfs_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(lfs::g_mu);
    lfs::g_on = on != 0;
    return LFS_OK;
}

// Restrict timing to the scopes called `name` (NULL or "" = all): keeps the event overhead out of a timed region
// that only needs its dominant kernel.
extern "C" int lfs_profile_filter(const char* name) {
    std::lock_guard<std::mutex> lk(lfs::g_mu);
    lfs::g_filter = name ? name : "";
    return LFS_OK;
}

// Waits for all recorded events, aggregates by kernel name, clears the log.
// names: max_entries x 64 chars; total_ms / counts: max_entries. Returns the number of names.
extern "C" int lfs_profile_collect(int max_entries, char* names, float* total_ms, int* counts) {
    std::lock_guard<std::mutex> lk(lfs::g_mu);
    int n = 0;
    for (auto& p : lfs::g_pending) {
        if (p.closed && hipEventSynchronize(p.e1) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
                int k = 0;
                for (; k < n; ++k) if (std::strncmp(names + 64 * k, p.name.c_str(), 63) == 0) break;
                if (k == n && n < max_entries) { std::strncpy(names + 64 * n, p.name.c_str(), 63); names[64 * n + 63] = 0; total_ms[n] = 0.f; counts[n] = 0; ++n; }
                if (k < n) { total_ms[k] += ms; counts[k] += 1; }
            }
        }
        lfs::g_pool.push_back(p.e0); lfs::g_pool.push_back(p.e1);
    }
    lfs::g_pending.clear();
    return n;
}
