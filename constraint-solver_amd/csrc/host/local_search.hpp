// host/local_search.hpp — host-side mirror of the reference's solver core, generic over a
// problem policy like the reference is over its traits (local-search/src/local_search.rs:16-90,
// iterated_local_search.rs:76-88).  Same names, same control flow; the one change is the seam
// this project exists for: lines :315-325 of LocalSearch::execute (enumerate -> filter tabu ->
// score -> take(window) -> sort -> first) are ONE call to the problem's neighbourhood evaluator,
// which runs on the GPU through the C ABI.
//
// A Problem policy provides
//   using Solution (totally ordered, ==), using Score (totally ordered), static bool is_best(Score)
//   Solution generate_initial_solution(Rng&)                       InitialSolutionGenerator
//   Score    score(const Solution&)                                 SolutionScoreCalculator
//   Solution propose_new_starting_solution(current, is_best_in_history, Rng&)   Perturbation
//   bool     best_of_neighbourhood(current, Rng& ls_rng, window, ScoredSolution& out, const History& tabu, bool continued)
//            == MoveProposer + tabu filter + score + take + sort + first; `tabu` is LocalSearch's own History: its
//            set is {current} for every configuration the reference ships (the GPU's built-in rule) and anything
//            else it holds is handed to the GPU as an exclusion list
#pragma once

#include <cstdint>
#include <deque>
#include <memory>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include "rng.hpp"

namespace lshost {

// ScoredSolution: field order (score, solution) IS the ordering (local_search.rs:29-37).
template <class P> struct ScoredSolution {
    typename P::Score score;
    typename P::Solution solution;
    bool operator<(const ScoredSolution &o) const
    {
        if (score < o.score) return true;
        if (o.score < score) return false;
        return solution < o.solution;
    }
    bool operator==(const ScoredSolution &o) const { return !(score < o.score) && !(o.score < score) && solution == o.solution; }
};

// History (local_search.rs:105-250): tabu bookkeeping + bounded best set.
template <class P> class History {
public:
    using SS = ScoredSolution<P>;
    History(size_t best_solutions_capacity, size_t all_solutions_capacity, uint64_t all_solution_iteration_expiry)
        : best_cap_(best_solutions_capacity), all_cap_(all_solutions_capacity), expiry_(all_solution_iteration_expiry) {}

    // :155-162
    void seen_solution(const SS &s)
    {
        iteration_count++;
        if (expiry_ >= 1) {
            // Every configuration the reference ships (expiry 1000 / 10000): _pop_solution_for_age evicts every entry that is
            // YOUNGER than the expiry (:182-195), i.e. all of them, and the solution just seen is added -- the set is {s} after
            // every call, whatever it held before.  Nothing reads its contents while it has one element (the evaluators' built-in
            // tabu rule IS "candidate == current"; tabu_size() is what callers ask), so the 32 KB copy, its allocation and the
            // previous entry's release are skipped: only the fact is recorded.
            single_ = true;
            return;
        }
        pop_solution_for_age();
        if (lookup_.find(s.solution) != lookup_.end()) return;
        add_solution(s);
    }
    // :197-199
    bool is_solution_tabu(const typename P::Solution &s) const
    {
        if (single_) throw std::logic_error("History: the single-entry set is not materialised (callers test tabu_size() == 1 first)");
        return lookup_.find(s) != lookup_.end();
    }
    // :201-203
    bool is_best_solution(const SS &s) const { return best_.count(s) != 0; }
    // :205-218
    void local_search_chose_solution(const SS &s)
    {
        if (best_.size() < best_cap_) { best_.insert(s); return; }
        const SS worst = *best_.rbegin();
        if (!(worst.score < s.score)) { // solution.score <= worst.score
            best_.erase(worst);
            best_.insert(s);
        }
    }
    // :220-229 (consumes one gen_index draw when non-empty)
    template <class R> bool get_random_best_solution(R &rng, SS &out) const
    {
        if (best_.empty()) return false;
        std::vector<const SS *> v;
        v.reserve(best_.size());
        for (const SS &s : best_) v.push_back(&s);
        out = *v[choose_index(rng, v.size())];
        return true;
    }
    // :239-244
    bool get_best(SS &out) const
    {
        if (best_.empty()) return false;
        out = *best_.begin();
        return true;
    }
    size_t tabu_size() const { return single_ ? 1 : lookup_.size(); }
    // visit every tabu solution (general History configurations hand them to the GPU as exclusions)
    template <class F> void for_each_tabu(F &&f) const
    {
        if (single_) throw std::logic_error("History: the single-entry set is not materialised");
        for (const auto &sp : lookup_) f(*sp);
    }
    uint64_t iteration_count = 0;

private:
    // One heap copy of a visited solution is shared by the deque and the lookup set (the reference clones it twice,
    // local_search.rs:164-171; at n = 4096 a board is 32 KB and this runs once per LS iteration).
    using SolPtr = std::shared_ptr<const typename P::Solution>;
    struct PtrLess {
        using is_transparent = void;
        bool operator()(const SolPtr &a, const SolPtr &b) const { return *a < *b; }
        bool operator()(const SolPtr &a, const typename P::Solution &b) const { return *a < b; }
        bool operator()(const typename P::Solution &a, const SolPtr &b) const { return a < *b; }
    };
    struct Entry { SolPtr solution; uint64_t iteration; };
    void add_solution(const SS &s) // :164-171
    {
        while (all_.size() > all_cap_) { // _pop_solution_for_size :173-180
            lookup_.erase(all_.back().solution);
            all_.pop_back();
        }
        SolPtr sp = std::make_shared<const typename P::Solution>(s.solution);
        all_.push_front(Entry{sp, iteration_count});
        lookup_.insert(std::move(sp));
    }
    void pop_solution_for_age() // :182-195 — evicts entries YOUNGER than the expiry (sic)
    {
        while (!all_.empty()) {
            const Entry &e = all_.back();
            if (e.iteration + expiry_ >= iteration_count) {
                lookup_.erase(e.solution);
                all_.pop_back();
                continue;
            }
            break;
        }
    }
    std::set<SS> best_;
    size_t best_cap_, all_cap_;
    std::deque<Entry> all_;
    std::set<SolPtr, PtrLess> lookup_;
    uint64_t expiry_;
    bool single_ = false; // expiry >= 1: the set is {the solution last seen}, not materialised
};

// LocalSearch (local_search.rs:253-342).
template <class P, class Rng = ChaCha20Rng> class LocalSearch {
public:
    using SS = ScoredSolution<P>;
    LocalSearch(P *problem, uint64_t max_iterations, size_t window_size, size_t best_solutions_capacity,
                size_t all_solutions_capacity, uint64_t all_solution_iteration_expiry, Rng rng)
        : p_(problem), max_iterations_(max_iterations), window_size_(window_size),
          history_(best_solutions_capacity, all_solutions_capacity, all_solution_iteration_expiry), rng_(rng) {}

    // :301-342
    SS execute(const typename P::Solution &start, uint64_t allow_no_improvement_for)
    {
        SS current{p_->score(start), start};
        // `best` of the reference (:303) is kept lazily: along a descent it IS the current solution, and copying a
        // 32 KB board per improving iteration buys nothing -- it is copied out when the walk first moves off it.
        SS best;
        bool best_is_current = true;
        uint64_t no_improvement_for = 0;
        bool continued = false; // `current` is the winner the previous iteration's best_of_neighbourhood produced (nothing else touched it)
        for (uint64_t it = 0; it < max_iterations_; it++) {
            history_.seen_solution(current);
            if (P::is_best(current.score)) return current;
            // :315-325 — the GPU seam.  The evaluator's built-in tabu rule is "candidate == current"; seen_solution has
            // just made sure the current solution IS in the set (it was there, or it has been added), whatever else a
            // general History keeps.
            SS &nb = nb_; // reused across iterations: the winner's board is copy-ASSIGNED into it (no 32 KB allocation per iteration)
            ls_iterations++;
            if (p_->best_of_neighbourhood(current, rng_, window_size_, nb, history_, continued)) {
                if (nb.score < current.score) {
                    best_is_current = true; // best = nb, which becomes current below
                    no_improvement_for = 0;
                } else {
                    if (best_is_current) { best = current; best_is_current = false; }
                    no_improvement_for++;
                    if (no_improvement_for >= allow_no_improvement_for) break;
                }
                std::swap(current, nb); // current = nb; the old current's storage becomes the next iteration's nb
                continued = true;
            } else {
                break;
            }
        }
        return best_is_current ? current : best;
    }
    uint64_t ls_iterations = 0;

private:
    P *p_;
    uint64_t max_iterations_;
    size_t window_size_;
    History<P> history_;
    Rng rng_;
    SS nb_;
};

// AcceptanceCriterion::choose (iterated_local_search.rs:51-71): weighted pick among
// existing x1, new x5 and (if any) a random member of the best set x1.  The weights are i32
// literals in the reference => 32-bit draws.
template <class P, class R>
inline ScoredSolution<P> acceptance_choose(const ScoredSolution<P> &existing, const ScoredSolution<P> &fresh,
                                           const History<P> &history, R &rng)
{
    ScoredSolution<P> random_best;
    const bool have = history.get_random_best_solution(rng, random_best);
    std::vector<uint64_t> w = have ? std::vector<uint64_t>{1, 5, 1} : std::vector<uint64_t>{1, 5};
    const size_t i = choose_weighted_index(rng, w, /*wide=*/false);
    return i == 0 ? existing : (i == 1 ? fresh : random_best);
}

struct IterationInfo { uint64_t current, total; }; // iterated_local_search.rs:90-94

// IteratedLocalSearch (iterated_local_search.rs:96-202).
template <class P, class Rng = ChaCha20Rng> class IteratedLocalSearch {
public:
    using SS = ScoredSolution<P>;
    // :130-156
    IteratedLocalSearch(P *problem, LocalSearch<P, Rng> local_search, History<P> history, uint64_t max_iterations,
                        uint64_t max_allow_no_improvement_for, Rng rng)
        : p_(problem), local_search_(std::move(local_search)), history_(std::move(history)),
          max_iterations_(max_iterations), max_allow_no_improvement_for_(max_allow_no_improvement_for), rng_(rng)
    {
        typename P::Solution s = p_->generate_initial_solution(rng_);
        current_ = SS{p_->score(s), s};
    }
    IterationInfo get_iteration_info() const { return IterationInfo{iteration_, max_iterations_}; } // :158-163
    bool get_best_solution(SS &out) const { return history_.get_best(out); }                        // :165-167
    bool is_finished() const { return iteration_ >= max_iterations_; }                              // :169-171
    const SS &current() const { return current_; }
    uint64_t ls_iterations() const { return local_search_.ls_iterations; }

    // :173-202
    void execute_round()
    {
        iteration_++;
        SS best;
        if (history_.get_best(best) && P::is_best(best.score)) return;
        if (iteration_ > 0 && iteration_ % 50 == 0) {
            typename P::Solution s = p_->generate_initial_solution(rng_);
            current_ = SS{p_->score(s), s};
        }
        const typename P::Solution perturbed =
            p_->propose_new_starting_solution(current_, history_.is_best_solution(current_), rng_);
        const SS fresh = local_search_.execute(perturbed, max_allow_no_improvement_for_);
        history_.local_search_chose_solution(fresh);
        current_ = acceptance_choose(current_, fresh, history_, rng_);
    }

private:
    P *p_;
    LocalSearch<P, Rng> local_search_;
    History<P> history_;
    uint64_t iteration_ = 0, max_iterations_, max_allow_no_improvement_for_;
    Rng rng_;
    SS current_;
};

} // namespace lshost
