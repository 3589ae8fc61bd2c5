// plan_cache.hpp — parking lot for solver plans.  The reference constructs a solver per frame (dyn_fusion.cpp:193); a
// plan is 30-40 device allocations, so the plans of destroyed solvers are parked (at most two per kind) and handed to the
// next solver whose problem fits.
#pragma once
#include <mutex>
#include <vector>

namespace dfa {

template <class Plan, void (*Destroy)(Plan*)>
class PlanCache {
    struct Entry {
        Plan* plan;
        int max_D, max_N, k;
    };
    std::mutex mu_;
    std::vector<Entry> idle_;

public:
    ~PlanCache() {
        for (auto& e : idle_) Destroy(e.plan);
    }
    // a parked plan for k neighbours with room for D nodes and N vertices (its capacity in max_D / max_N), or nullptr
    Plan* take(int D, int N, int k, int& max_D, int& max_N) {
        std::lock_guard<std::mutex> lock(mu_);
        for (size_t i = 0; i < idle_.size(); ++i)
            if (idle_[i].k == k && idle_[i].max_D >= D && idle_[i].max_N >= N) {
                const Entry e = idle_[i];
                idle_.erase(idle_.begin() + (long)i);
                max_D = e.max_D, max_N = e.max_N;
                return e.plan;
            }
        return nullptr;
    }
    void park(Plan* plan, int max_D, int max_N, int k) {
        if (!plan) return;
        std::lock_guard<std::mutex> lock(mu_);
        if (idle_.size() >= 2) {
            Destroy(idle_.front().plan);
            idle_.erase(idle_.begin());
        }
        idle_.push_back({plan, max_D, max_N, k});
    }
};

}  // namespace dfa
