// stand-in for <tbb/blocked_range.h> — test infrastructure only (oracle/_ref); TBB headers are absent from this image.
#pragma once
#include <cstddef>
namespace tbb {
struct split {};
template <typename Value>
class blocked_range {
    Value b_, e_;
    std::size_t grain_;

  public:
    typedef Value const_iterator;
    blocked_range(Value b, Value e, std::size_t grainsize = 1) : b_(b), e_(e), grain_(grainsize ? grainsize : 1) {}
    // (the split point is rounded down to a multiple of 64 where that leaves both halves non-empty: the reference's functor sets
    //  bits of a std::vector<bool> from every worker — a data race on the words two ranges share; aligned ranges share none)
    blocked_range(blocked_range &r, split) : b_(r.b_ + (r.e_ - r.b_) / 2), e_(r.e_), grain_(r.grain_) {
        const Value aligned = b_ - (b_ % 64);
        if (aligned > r.b_) b_ = aligned;
        r.e_ = b_;
    }
    Value begin() const { return b_; }
    Value end() const { return e_; }
    std::size_t size() const { return (std::size_t)(e_ - b_); }
    std::size_t grainsize() const { return grain_; }
    bool empty() const { return !(b_ < e_); }
    bool is_divisible() const { return grain_ < size(); }
};
}  // namespace tbb
