// stand-in for <tbb/parallel_reduce.h> — test infrastructure only (oracle/_ref).
// The imperative form the reference uses (map_eval.cpp:1716): the range is halved while it is divisible, the right half
// runs on a Body made by the splitting constructor, and the halves are joined left-to-right.  The top levels of the
// recursion run on std::threads (as many leaves as hardware threads); the split/join TREE is fixed by the range and the
// grain size, so the result is deterministic — real TBB's depends on work stealing.
#pragma once
#include <thread>

#include "blocked_range.h"
namespace tbb {
namespace standin_detail {
template <typename Range, typename Body>
void reduce(Range &range, Body &body, int par_levels) {
    if (!range.is_divisible()) {
        body(range);
        return;
    }
    Range right(range, split());
    Body rbody(body, split());
    if (par_levels > 0) {
        std::thread t([&] { reduce(right, rbody, par_levels - 1); });
        reduce(range, body, par_levels - 1);
        t.join();
    } else {
        reduce(range, body, 0);
        reduce(right, rbody, 0);
    }
    body.join(rbody);
}
}  // namespace standin_detail
template <typename Range, typename Body>
void parallel_reduce(const Range &range, Body &body) {
    unsigned hw = std::thread::hardware_concurrency();
    int levels = 0;
    while ((1u << levels) < (hw ? hw : 1u) && levels < 8) ++levels;
    Range r(range);
    standin_detail::reduce(r, body, levels);
}
}  // namespace tbb
