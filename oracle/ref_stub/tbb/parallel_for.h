// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. oneTBB is not in this image; the reference's PLY reader uses tbb::parallel_for over tbb::blocked_range for independent
// per-vertex work: run serially here.
#pragma once
#include <cstddef>
namespace tbb {
    template <class T> class blocked_range {
    public:
        blocked_range(T b, T e, std::size_t grain = 1) : b_(b), e_(e) { (void)grain; }
        T begin() const { return b_; }
        T end() const { return e_; }
    private:
        T b_, e_;
    };
    template <class Range, class F> void parallel_for(const Range& r, const F& f) { f(r); }
} // namespace tbb
