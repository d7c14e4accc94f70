// host/plugins.hpp — the two problem plug-ins of the reference as policies for
// host/local_search.hpp.  Move proposal, perturbation and initial solutions are host code
// (restated from examples/nqueens/src/lib.rs and examples/employee-scheduling/src/lib.rs);
// every score and every neighbourhood goes through the C ABI to the HIP kernels.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/ls_amd.h"
#include "local_search.hpp"

namespace lshost {

inline void check_rc(ls_ctx *ctx, int rc, const char *what)
{
    if (rc != LS_OK) throw std::runtime_error(std::string(what) + ": " + ls_last_error(ctx));
}

// Lockstep batching of independent restarts: every restart runs the unchanged
// driver on its own host thread; its GPU requests park here until every live restart has one
// pending, then ONE batched launch serves them all.  Results are independent of the batching.
struct NqRequest {
    int kind; // 0 = column scores (+ total), 1 = neighbourhood
    const int32_t *rows;
    int64_t cur_score;
    const int32_t *cols;
    int32_t n_cols;
    int64_t window;
    int64_t *col_scores_out; // kind 0: column scores of rows; kind 1: of the WINNER board (same round trip)
    int64_t *total_out;
    ls_nq_result *result_out;
    std::atomic<bool> done;
};

class NqBatcher {
public:
    NqBatcher(ls_ctx *ctx, int32_t n, int live) : ctx_(ctx), n_(n), live_(live) {}
    void submit(NqRequest *r)
    {
        std::unique_lock<std::mutex> lk(m_);
        if (!error_.empty()) throw std::runtime_error(error_); // a restart already failed: nobody parks any more
        r->done.store(false, std::memory_order_relaxed);
        pending_.push_back(r);
        if ((int)pending_.size() == live_) {
            flush_locked();
        } else {
            // a lockstep step lasts a few hundred microseconds: spin briefly before sleeping, a futex
            // wake-up per restart per step would otherwise dominate
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (!r->done.load(std::memory_order_acquire) && !failed_.load(std::memory_order_relaxed) &&
                   std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            lk.lock();
            cv_.wait(lk, [&] { return r->done.load(std::memory_order_acquire) || !error_.empty(); });
        }
        if (!error_.empty()) {
            // `r` lives on this thread's stack, which unwinds now: it must not stay parked (a later flush would
            // write r->done through a dangling pointer)
            for (size_t i = 0; i < pending_.size(); i++)
                if (pending_[i] == r) { pending_.erase(pending_.begin() + (long)i); break; }
            throw std::runtime_error(error_);
        }
    }
    void thread_done()
    {
        std::unique_lock<std::mutex> lk(m_);
        live_--;
        if (live_ > 0 && (int)pending_.size() == live_) flush_locked();
    }
    void fail(const std::string &msg)
    {
        std::unique_lock<std::mutex> lk(m_);
        if (error_.empty()) error_ = msg;
        failed_.store(true);
        // release every parked restart; each finds error_ set, takes itself off the list and throws
        for (NqRequest *r : pending_) r->done.store(true, std::memory_order_release);
        pending_.clear();
        cv_.notify_all();
    }
    uint64_t launches = 0;
    double t_flush = 0, t_call = 0;
    uint64_t n_flush = 0, sum_R = 0;
    ~NqBatcher() { if (getenv("LS_AMD_TRACE")) fprintf(stderr, "batcher: %llu flushes %.3f s (ABI calls %.3f s), mean restarts per launch %.2f\n", (unsigned long long)n_flush, t_flush, t_call, (double)sum_R / (launches ? launches : 1)); }

private:
    void flush_locked()
    {
        const auto TF0 = std::chrono::steady_clock::now();
        struct Tm { NqBatcher *b; std::chrono::steady_clock::time_point t0; ~Tm() { b->t_flush += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); b->n_flush++; } } tm{this, TF0};
        try {
            if (!error_.empty()) throw std::runtime_error(error_); // a restart failed: release everyone
            std::vector<NqRequest *> cs, ev;
            for (NqRequest *r : pending_) (r->kind == 0 ? cs : ev).push_back(r);
            if (!cs.empty()) {
                const size_t R = cs.size();
                rows_.resize(R * n_);
                colscores_.resize(R * n_);
                totals_.resize(R);
                for (size_t i = 0; i < R; i++) std::copy(cs[i]->rows, cs[i]->rows + n_, rows_.begin() + i * n_);
                check_rc(ctx_, ls_nqueens_col_scores(ctx_, (int32_t)R, n_, rows_.data(), colscores_.data(), totals_.data()),
                         "ls_nqueens_col_scores");
                launches++;
                for (size_t i = 0; i < R; i++) {
                    if (cs[i]->col_scores_out)
                        std::copy(colscores_.begin() + i * n_, colscores_.begin() + (i + 1) * n_, cs[i]->col_scores_out);
                    if (cs[i]->total_out) *cs[i]->total_out = totals_[i];
                }
            }
            if (!ev.empty()) {
                const size_t R = ev.size();
                int32_t max_cols = 1;
                for (NqRequest *r : ev) max_cols = r->n_cols > max_cols ? r->n_cols : max_cols;
                rows_.resize(R * n_);
                cols_.assign(R * max_cols, 0);
                ncols_.resize(R);
                cur_.resize(R);
                results_.resize(R);
                colscores_.resize(R * n_);
                for (size_t i = 0; i < R; i++) {
                    std::copy(ev[i]->rows, ev[i]->rows + n_, rows_.begin() + i * n_);
                    std::copy(ev[i]->cols, ev[i]->cols + ev[i]->n_cols, cols_.begin() + i * max_cols);
                    ncols_[i] = ev[i]->n_cols;
                    cur_[i] = ev[i]->cur_score;
                    if (ev[i]->window != ev[0]->window) throw std::runtime_error("restarts of one batch must share the window size");
                }
                const auto TC0 = std::chrono::steady_clock::now();
                check_rc(ctx_, ls_nqueens_eval_batch_next(ctx_, (int32_t)R, n_, rows_.data(), cur_.data(), cols_.data(),
                                                          ncols_.data(), max_cols, ev[0]->window, results_.data(),
                                                          colscores_.data()),
                         "ls_nqueens_eval_batch_next");
                t_call += std::chrono::duration<double>(std::chrono::steady_clock::now() - TC0).count();
                sum_R += R;
                launches++;
                for (size_t i = 0; i < R; i++) {
                    *ev[i]->result_out = results_[i];
                    if (ev[i]->col_scores_out)
                        std::copy(colscores_.begin() + i * n_, colscores_.begin() + (i + 1) * n_, ev[i]->col_scores_out);
                }
            }
        } catch (const std::exception &e) {
            if (error_.empty()) error_ = e.what();
            failed_.store(true);
        }
        for (NqRequest *r : pending_) r->done.store(true, std::memory_order_release);
        pending_.clear();
        cv_.notify_all();
    }
    ls_ctx *ctx_;
    int32_t n_;
    int live_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<NqRequest *> pending_;
    std::string error_;
    std::atomic<bool> failed_{false};
    std::vector<int32_t> rows_, cols_, ncols_;
    std::vector<int64_t> cur_, colscores_, totals_;
    std::vector<ls_nq_result> results_;
};

// =================================================================== N-queens
struct NQueensProblem {
    using Solution = std::vector<int64_t>; // rows[col] = row, Integer = i64 (lib.rs:13-21)
    using Score = int64_t;                 // NQueensScore (lib.rs:63-71)
    using SS = ScoredSolution<NQueensProblem>;
    static bool is_best(Score s) { return s == 0; }

    ls_ctx *ctx;
    size_t board_size;
    NqBatcher *batcher = nullptr; // set when this restart is one of a lockstep batch
    // statistics
    uint64_t neighbours_scored = 0, gpu_evals = 0, gpu_col_score_calls = 0;
    double t_prop = 0, t_gpu = 0, t_samp = 0;
    uint64_t n_conf = 0;
    ~NQueensProblem() { if (getenv("LS_AMD_TRACE")) fprintf(stderr, "nq: evals %llu prop %.3f s (weighted sample %.3f s, %.1f conflict cols/iter) gpu %.3f s\n", (unsigned long long)gpu_evals, t_prop, t_samp, (double)n_conf / (gpu_evals ? gpu_evals : 1), t_gpu); }
    // column scores of the board the loop will expand next, delivered by the previous neighbourhood's
    // round trip (or by score()): saves one GPU round trip per LS iteration
    mutable Solution cached_board;
    mutable std::vector<int32_t> cached_rows32;  // cached_board packed for the ABI, kept in step with it
    mutable std::vector<int64_t> cached_cs;
    mutable int64_t cached_total = 0;
    mutable bool cache_valid = false;
    mutable bool last_lookup_hit = false;        // the board of the last col_scores() call IS cached_board
    mutable bool cache_is_last_winner = false;   // cached_board is the winner board the last best_of_neighbourhood produced
    // per-iteration scratch, reused (at n = 4096 a fresh 32 KB vector per LS iteration costs microseconds)
    std::vector<int64_t> spare_cs;
    mutable std::vector<uint32_t> conflict_cols_buf;
    mutable std::vector<double> weights_buf;

    NQueensProblem(ls_ctx *c, size_t n, NqBatcher *b = nullptr) : ctx(c), board_size(n), batcher(b) {}
    // Optional, set by the solver entry (which knows the context's internals): registers work the library may do
    // while it waits for a launch of this context (here: computing the generator's key stream ahead); fn returns false
    // once there is nothing left to do.  nullptr arguments clear it.
    void (*set_idle_work)(ls_ctx *, bool (*fn)(void *), void *arg) = nullptr;
    int64_t last_move_col = -1; // the column the previous best_of_neighbourhood call moved (fast route), -1: none

    static const int64_t *rows_dummy() { static const int64_t z = 0; return &z; }
    std::vector<int32_t> pack(const Solution &s) const
    {
        std::vector<int32_t> r(s.size());
        for (size_t i = 0; i < s.size(); i++) r[i] = (int32_t)s[i];
        return r;
    }

    // NQueensInitialSolutionGenerator (lib.rs:152-161): identity permutation, shuffled
    template <class R> Solution generate_initial_solution(R &rng) const
    {
        Solution rows(board_size);
        for (size_t i = 0; i < board_size; i++) rows[i] = (int64_t)i;
        shuffle(rng, rows);
        return rows;
    }

    // get_col_scores (lib.rs:74-87) on the GPU; also returns the sum (lib.rs:130-139)
    // (returns a reference to the cache: valid until the next call)
    // `trusted`: the caller vouches that `s` IS the cached board (LocalSearch continuing from the winner the previous call
    // produced: local_search.hpp `continued`): the 2 x 32 KB compare is skipped
    const std::vector<int64_t> &col_scores(const Solution &s, int64_t *total, bool trusted = false) const
    {
        if (cache_valid && ((trusted && cache_is_last_winner && s.size() == cached_board.size()) || s == cached_board)) {
            last_lookup_hit = true;
            if (total) *total = cached_total;
            return cached_cs;
        }
        last_lookup_hit = false;
        const_cast<NQueensProblem *>(this)->gpu_col_score_calls++;
        const std::vector<int32_t> rows = pack(s);
        std::vector<int64_t> cs(s.size());
        int64_t tot = 0;
        if (batcher) {
            NqRequest rq;
            rq.kind = 0; rq.rows = rows.data(); rq.cur_score = 0; rq.cols = nullptr; rq.n_cols = 0; rq.window = 0;
            rq.col_scores_out = cs.data(); rq.total_out = &tot; rq.result_out = nullptr;
            batcher->submit(&rq);
        } else
            check_rc(ctx, ls_nqueens_col_scores(ctx, 1, (int32_t)s.size(), rows.data(), cs.data(), &tot), "ls_nqueens_col_scores");
        {   // fail loudly on an impossible answer: the column scores are non-negative and sum to the (even) total
            int64_t sum = 0;
            bool ok = true;
            for (int64_t v : cs) { sum += v; ok = ok && v >= 0; }
            if (!ok || sum != tot || (tot & 1))
                throw std::runtime_error("ls_nqueens_col_scores returned inconsistent column scores (sum " + std::to_string(sum) +
                                         ", total " + std::to_string(tot) + ")");
        }
        if (total) *total = tot;
        cached_board = s; cached_rows32 = rows; cached_cs.swap(cs); cached_total = tot; cache_valid = true;
        cache_is_last_winner = false;
        last_lookup_hit = true;
        return cached_cs;
    }
    Score score(const Solution &s) const
    {
        int64_t tot = 0;
        (void)col_scores(s, &tot);
        return tot;
    }

    // the columns with a non-zero score and their sampling weights score + 1e-4 (lib.rs:184-190), in column order
#if defined(__x86_64__)
    __attribute__((target("avx512f,avx512dq,avx512vl"))) static size_t nonzero_columns_avx512(const int64_t *cs, size_t n, uint32_t *cols, double *weights, int64_t *sum)
    {
        const __m512d eps = _mm512_set1_pd(1e-4);
        const __m512i zero = _mm512_setzero_si512();
        __m512i acc = _mm512_setzero_si512();
        __m256i cur = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
        const __m256i eight = _mm256_set1_epi32(8);
        size_t k = 0, c = 0;
        for (; c + 8 <= n; c += 8) {
            const __m512i v = _mm512_loadu_si512(cs + c);
            acc = _mm512_add_epi64(acc, v);
            const __mmask8 m = _mm512_cmpneq_epi64_mask(v, zero);
            _mm256_mask_compressstoreu_epi32(cols + k, m, cur);
            _mm512_mask_compressstoreu_pd(weights + k, m, _mm512_add_pd(_mm512_cvtepi64_pd(v), eps));
            k += (size_t)__builtin_popcount((unsigned)m);
            cur = _mm256_add_epi32(cur, eight);
        }
        int64_t total = _mm512_reduce_add_epi64(acc);
        for (; c < n; c++) {
            total += cs[c];
            if (cs[c] != 0) { cols[k] = (uint32_t)c; weights[k] = (double)cs[c] + 1e-4; k++; }
        }
        *sum = total;
        return k;
    }
#endif
    // (also the sum of the scores: the consistency check "column scores sum to the score" rides on this pass)
    static size_t nonzero_columns(const int64_t *cs, size_t n, uint32_t *cols, double *weights, int64_t *sum)
    {
#if defined(__x86_64__)
        if (weighted_have_avx512()) return nonzero_columns_avx512(cs, n, cols, weights, sum);
#endif
        size_t k = 0;
        int64_t total = 0;
        for (size_t c = 0; c < n; c++) {
            total += cs[c];
            if (cs[c] != 0) { cols[k] = (uint32_t)c; weights[k] = (double)cs[c] + 1e-4; k++; }
        }
        *sum = total;
        return k;
    }

    // NQueensMoveProposer::iter_local_moves, the part before the iterator (lib.rs:182-205):
    // columns with conflicts, Efraimidis-Spirakis weighted sample of `amount`, then a random
    // number of them in choose_multiple order.  Empty result <=> `random_cols == None`.
    template <class R> std::vector<int32_t> propose_columns(const Solution &start, R &rng, bool trusted = false) const
    {
        int64_t want_total = 0;
        const std::vector<int64_t> &cs = col_scores(start, &want_total, trusted);
        std::vector<uint32_t> &conflict_cols = conflict_cols_buf; // n entries; the first n_conflict count (never shrunk: growing a
        std::vector<double> &weights = weights_buf;               // vector back to n would zero-fill ~36 KB per LS iteration)
        if (conflict_cols.size() < cs.size()) conflict_cols.resize(cs.size());
        if (weights.size() < cs.size()) weights.resize(cs.size());
        int64_t sum = 0;
        const size_t n_conflict = nonzero_columns(cs.data(), cs.size(), conflict_cols.data(), weights.data(), &sum);
        if (sum != want_total) // fail loudly on impossible column scores (they sum to the board's score) instead of walking on from them
            throw std::runtime_error("column scores of the current board sum to " + std::to_string(sum) + ", its score is " + std::to_string(want_total));
        if (n_conflict == 0) return {};
        size_t amount = start.size() / 20;
        if (amount < 1) amount = 1;
        if (amount > n_conflict) amount = n_conflict;
        const auto TA = std::chrono::steady_clock::now();
        const_cast<NQueensProblem *>(this)->n_conf += n_conflict;
        const std::vector<uint32_t> picked = index_sample_weighted(rng, weights.data(), (uint32_t)n_conflict, (uint32_t)amount);
        const_cast<NQueensProblem *>(this)->t_samp += std::chrono::duration<double>(std::chrono::steady_clock::now() - TA).count();
        const uint64_t num_cols = gen_range_u64_incl(rng, 1, picked.size());
        const std::vector<uint32_t> order = index_sample(rng, (uint32_t)picked.size(), (uint32_t)num_cols);
        std::vector<int32_t> out(order.size());
        for (size_t k = 0; k < order.size(); k++) out[k] = (int32_t)conflict_cols[picked[order[k]]]; // cols[i] = conflict_cols[picked[i]]
        return out;
    }

    // local_search.rs:315-325 for this plug-in: one GPU launch
    // General History (all_solution_iteration_expiry == 0: the tabu set keeps every visited board, local_search.rs:155-199):
    // the candidates that equal a tabu board other than the current one, as unfiltered positions k = slot * n + value.
    // A candidate differs from the current board in exactly one column, so only tabu boards at Hamming distance 1 matter.
    std::vector<int64_t> tabu_positions(const Solution &cur, const std::vector<int32_t> &cols, const History<NQueensProblem> &tabu) const
    {
        std::vector<int64_t> out;
        const size_t n = cur.size();
        tabu.for_each_tabu([&](const Solution &t) {
            if (t.size() != n) return;
            size_t diff_col = 0, diffs = 0;
            for (size_t j = 0; j < n && diffs < 2; j++)
                if (t[j] != cur[j]) { diff_col = j; diffs++; }
            if (diffs != 1) return; // the current board itself (no-ops: always tabu on the device) or too far away
            for (size_t s = 0; s < cols.size(); s++)
                if ((size_t)cols[s] == diff_col) out.push_back((int64_t)(s * n) + t[diff_col]);
        });
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
        return out;
    }

    template <class R> bool best_of_neighbourhood(const SS &current, R &rng, size_t window, SS &out, const History<NQueensProblem> &tabu,
                                                  bool continued = false)
    {
        const auto T0 = std::chrono::steady_clock::now();
        const std::vector<int32_t> cols = propose_columns(current.solution, rng, continued);
        cache_is_last_winner = false; // until this call has put its winner into the cache (the general-tabu route does not)
        const int64_t prev_move_col = last_move_col;
        last_move_col = -1;           // until this call has moved a queen on the fast route below
        if (cols.empty()) return false;
        // propose_columns has just looked the current board up: it is the cached one (already packed) in the steady state
        const bool cur_is_cached = last_lookup_hit && cached_rows32.size() == current.solution.size();
        std::vector<int32_t> packed_now;
        if (!cur_is_cached) packed_now = pack(current.solution);
        const std::vector<int32_t> &rows = cur_is_cached ? cached_rows32 : packed_now;
        if (tabu.tabu_size() != 1) { // more than {current} is tabu: exclusion list, one-restart launch (never batched)
            const std::vector<int64_t> excl = tabu_positions(current.solution, cols, tabu);
            ls_nq_result res;
            check_rc(ctx, ls_nqueens_eval_tabu(ctx, (int32_t)rows.size(), rows.data(), current.score, cols.data(), (int32_t)cols.size(),
                                               (int64_t)window, excl.empty() ? rows_dummy() : excl.data(), (int32_t)excl.size(), &res, nullptr),
                     "ls_nqueens_eval_tabu");
            gpu_evals++;
            neighbours_scored += (uint64_t)res.n_scored;
            if (res.best_index < 0) return false;
            out.solution = current.solution;
            out.solution[(size_t)res.best_col] = res.best_value;
            out.score = res.best_score;
            return true; // the winner's column scores are fetched by the next col_scores() call (a cache miss)
        }
        const auto T1 = std::chrono::steady_clock::now();
        t_prop += std::chrono::duration<double>(T1 - T0).count();
        ls_nq_result res;
        std::vector<int64_t> &next_cs = spare_cs;
        next_cs.resize(rows.size());
        if (batcher) {
            NqRequest rq;
            rq.kind = 1; rq.rows = rows.data(); rq.cur_score = current.score; rq.cols = cols.data();
            rq.n_cols = (int32_t)cols.size(); rq.window = (int64_t)window;
            rq.col_scores_out = next_cs.data(); rq.total_out = nullptr; rq.result_out = &res;
            batcher->submit(&rq);
        } else {
            const int32_t nc = (int32_t)cols.size();
            // while the launch is in flight the waiting thread computes the next iteration's key stream (the weighted
            // sample draws one f64 per conflicting column: about 1100 per iteration at n = 4096)
            struct IdleWork {
                NQueensProblem *p;
                IdleWork(NQueensProblem *q, R &g) : p(q)
                {
                    if constexpr (std::is_same<R, ChaCha20Rng>::value) {
                        if (p->set_idle_work) p->set_idle_work(p->ctx, [](void *x) { return static_cast<ChaCha20Rng *>(x)->prefetch(); }, &g);
                    }
                }
                ~IdleWork() { if (p->set_idle_work) p->set_idle_work(p->ctx, nullptr, nullptr); }
            } idle(this, rng);
            check_rc(ctx, ls_nqueens_eval_batch_next(ctx, 1, (int32_t)rows.size(), rows.data(), &current.score, cols.data(),
                                                     &nc, nc, (int64_t)window, &res, next_cs.data()),
                     "ls_nqueens_eval_batch_next");
        }
        t_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - T1).count();
        gpu_evals++;
        neighbours_scored += (uint64_t)res.n_scored;
        if (res.best_index < 0) return false;
        {   // fail loudly on an impossible record instead of walking on from it
            const int64_t n = (int64_t)rows.size();
            const int64_t sum = res.best_score; // (the column scores' sum is checked by the next iteration's scan of them: propose_columns)
            const bool ok = res.best_score >= 0 && !(res.best_score & 1) && res.best_col >= 0 && res.best_col < n &&
                            res.best_value >= 0 && res.best_value < n && res.best_index / n < (int64_t)cols.size() &&
                            cols[(size_t)(res.best_index / n)] == res.best_col && res.best_index % n == res.best_value &&
                            res.n_scored > 0 && res.n_scored <= (int64_t)window && sum == res.best_score;
            if (!ok)
                throw std::runtime_error("neighbourhood launch returned an inconsistent record: n_scored " + std::to_string(res.n_scored) +
                                         " best_index " + std::to_string(res.best_index) + " best_score " + std::to_string(res.best_score) +
                                         " best_col " + std::to_string(res.best_col) + " best_value " + std::to_string(res.best_value) +
                                         " cur " + std::to_string(current.score) + " sum(next col scores) " + std::to_string(sum));
        }
        // `continued`: LocalSearch vouches that `out` still holds the board `current` was derived from (it swaps the two after
        // every accepted move, local_search.hpp) -- they differ in the one column of the previous move, so one entry is
        // repaired instead of 8n bytes copied
        if (continued && prev_move_col >= 0 && out.solution.size() == current.solution.size())
            out.solution[(size_t)prev_move_col] = current.solution[(size_t)prev_move_col];
        else
            out.solution = current.solution;
        out.solution[(size_t)res.best_col] = res.best_value; // MoveIterator, lib.rs:226-228
        out.score = res.best_score;
        last_move_col = res.best_col;
        // the loop continues from this board (local_search.rs:335): its column scores came back with the result
        if (cur_is_cached) { // cached_board == current.solution: apply the move in place
            cached_board[(size_t)res.best_col] = res.best_value;
            cached_rows32[(size_t)res.best_col] = res.best_value;
        } else {
            cached_board = out.solution;
            cached_rows32 = packed_now;
            cached_rows32[(size_t)res.best_col] = res.best_value;
        }
        cached_cs.swap(next_cs); cached_total = res.best_score; cache_valid = true;
        cache_is_last_winner = true;
        return true;
    }

    // NQueensPerturbation (lib.rs:285-320), default strategy table (:274-283)
    template <class R> Solution propose_new_starting_solution(const SS &current, bool current_is_in_best_set, R &rng) const
    {
        const size_t strategy = choose_weighted_index(rng, {100, 10}, /*wide=*/true); // 0 ChangeSubset, 1 DoNothing
        Solution next = current.solution;
        if (strategy == 1) return next;
        const uint64_t n = current.solution.size();
        std::vector<uint64_t> rows(n);
        for (uint64_t i = 0; i < n; i++) rows[i] = i;
        shuffle(rng, rows);
        auto clamp1 = [&](uint64_t x) { return x < 1 ? (uint64_t)1 : (x > n ? n : x); };
        const uint64_t k = current_is_in_best_set ? gen_range_u64_incl(rng, 1, clamp1(n / 20))
                                                  : gen_range_u64_incl(rng, 1, clamp1(n / 2));
        for (uint64_t i = 0; i < k; i++) next[(size_t)rows[i]] = (int64_t)gen_range_u64(rng, 0, n);
        return next;
    }
};

// ========================================================= employee scheduling
struct SchedProblem {
    using Solution = std::vector<int32_t>;          // date_to_employee as employee indices (lib.rs:140)
    using Score = std::pair<int64_t, int64_t>;      // ScheduleScore {hard, soft} (lib.rs:239-243)
    using SS = ScoredSolution<SchedProblem>;
    static bool is_best(const Score &s) { return s.first == 0 && s.second == 0; }

    ls_ctx *ctx;
    int32_t n_days_scored, n_days_stored, n_employees;
    uint64_t neighbours_scored = 0, gpu_evals = 0;
    // ScheduleRandomMoveProposer's stream (lib.rs:444-490).  The iterator owns rng.clone()
    // (:488) and nothing else ever advances LocalSearch's rng, so every LS iteration replays
    // the SAME draw stream; it is generated once, lazily, from a clone.
    std::vector<ls_sched_move> stream;
    ChaCha20Rng stream_rng;
    bool stream_started = false;
    size_t resident = 0; // entries of `stream` already on the device (ls_sched_moves_upload)

    SchedProblem(ls_ctx *c, int32_t d, int32_t s, int32_t e) : ctx(c), n_days_scored(d), n_days_stored(s), n_employees(e) {}
    // How many moves of the stream an LS iteration asks the GPU to consider for a window.  No-ops do not consume window and about
    // one move in `n_employees` is one (the example's seven employees: 14 %, and a restart whose FIXED stream starts badly fails
    // again and again), so: a quarter of slack for small rosters, an eighth from 64 employees on; a call that still cannot fill
    // the window doubles the prefix -- the result is that of the reference's lazy take(window) either way.
    static long long list_prefix(uint64_t window, int32_t n_employees)
    {
        return (long long)window + (long long)(window / (n_employees < 64 ? 4 : 8)) + 16;
    }

    // ScheduleInitialSolutionGenerator (lib.rs:400-420): pushes D + 1 entries
    template <class R> Solution generate_initial_solution(R &rng) const
    {
        Solution s;
        for (int32_t i = 0; i < n_days_stored; i++) s.push_back((int32_t)choose_index(rng, (size_t)n_employees));
        return s;
    }

    Score score(const Solution &s) const
    {
        int64_t h = 0, so = 0;
        check_rc(ctx, ls_sched_score(ctx, 1, s.data(), &h, &so, nullptr), "ls_sched_score");
        return Score(h, so);
    }

    void extend_stream(const ChaCha20Rng &ls_rng, size_t want)
    {
        if (!stream_started) { stream_rng = ls_rng; stream_started = true; }
        if (stream.size() >= want) return;
        stream.reserve(want);
        // The draws of lib.rs:459-478, move by move, without the per-move allocations of the generic helpers (a window of 65 536
        // asks for > 80 000 moves before the first LS iteration): the same values in the same order --
        //   choose_weighted(ChangeDay: 1, SwapDays: 4) = WeightedIndex<u64>: Uniform<u64>(0, 5) against the cumulative weights [1]
        //   days.choose / employees.choose           = gen_index (choose_index)
        //   days.choose_multiple(rng, 2)              = index::sample(len, 2): Floyd's, amount < 50 (index_sample; checked against it
        //                                              by tests/test_host_rng.py through the solver trajectories and the oracle driver)
        const bool floyd2 = n_days_scored >= 2 && index_sample_uses_floyd((uint32_t)n_days_scored, 2);
        while (stream.size() < want) {
            const size_t t = uniform_u64_below(stream_rng, 5) < 1 ? 0 : 1; // ChangeDay:1, SwapDays:4 (:435)
            ls_sched_move m;
            if (t == 0) {
                m.type = 0;
                m.a = (int32_t)choose_index(stream_rng, (size_t)n_days_scored); // days_to_employees.choose (:467)
                m.b = (int32_t)choose_index(stream_rng, (size_t)n_employees);   // employees.choose (:468)
            } else if (floyd2) {
                const uint32_t len = (uint32_t)n_days_scored;
                const uint32_t t0 = gen_range_u32_incl(stream_rng, 0, len - 2);
                const uint32_t t1 = gen_range_u32_incl(stream_rng, 0, len - 1);
                m.type = 1;
                if (t1 == t0) { m.a = (int32_t)(len - 1); m.b = (int32_t)t0; } // "insert j before t"
                else { m.a = (int32_t)t0; m.b = (int32_t)t1; }
            } else {
                const std::vector<uint32_t> xs = index_sample(stream_rng, (uint32_t)n_days_scored, 2); // :473
                m.type = 1;
                m.a = (int32_t)xs[0];
                m.b = (int32_t)xs[1];
            }
            stream.push_back(m);
        }
    }

    static Solution apply_move(const Solution &base, const ls_sched_move &m)
    {
        Solution s = base;
        if (m.type == 0) s[(size_t)m.a] = m.b;
        else {
            s[(size_t)m.a] = base[(size_t)m.b];
            s[(size_t)m.b] = base[(size_t)m.a];
        }
        return s;
    }

    // General History: moves of the stream prefix that lead to a tabu solution other than the current one are marked
    // LS_SCHED_MOVE_SKIP in a per-call copy of the list (the resident list stays as drawn).
    bool best_of_neighbourhood_general(const SS &current, ChaCha20Rng &rng, size_t window, SS &out, const History<SchedProblem> &tabu)
    {
        size_t want = window + window / 4 + 16;
        for (;;) {
            extend_stream(rng, want);
            std::vector<ls_sched_move> moves(stream.begin(), stream.begin() + (long)want);
            for (ls_sched_move &m : moves) {
                const Solution cand = apply_move(current.solution, m);
                if (cand != current.solution && tabu.is_solution_tabu(cand)) m.type = LS_SCHED_MOVE_SKIP;
            }
            ls_sched_result res;
            check_rc(ctx, ls_sched_eval(ctx, current.solution.data(), moves.data(), (int64_t)moves.size(), (int64_t)window, &res, nullptr, nullptr),
                     "ls_sched_eval");
            gpu_evals++;
            if ((size_t)res.n_scored == window) {
                neighbours_scored += (uint64_t)res.n_scored;
                out.solution = apply_move(current.solution, stream[(size_t)res.best_index]);
                out.score = Score(res.best_hard, res.best_soft);
                return true;
            }
            if (want > 64 * window + (1u << 16))
                throw std::runtime_error("the random move stream cannot fill the window (the reference would not terminate)");
            want *= 2;
        }
    }

    bool best_of_neighbourhood(const SS &current, ChaCha20Rng &rng, size_t window, SS &out, const History<SchedProblem> &tabu, bool = false)
    {
        if (window == 0) return false;
        if (tabu.tabu_size() != 1) return best_of_neighbourhood_general(current, rng, window, out, tabu);
        size_t want = (size_t)list_prefix(window, n_employees);
        for (;;) {
            extend_stream(rng, want);
            if (resident < want) { // the stream is the same for the whole run: hand it over once (a little beyond what this call asks for)
                extend_stream(rng, want + want / 16);
                check_rc(ctx, ls_sched_moves_upload(ctx, stream.data(), (int64_t)stream.size()), "ls_sched_moves_upload");
                resident = stream.size();
            }
            ls_sched_result res;
            check_rc(ctx, ls_sched_eval_resident(ctx, current.solution.data(), (int64_t)want, (int64_t)window, &res),
                     "ls_sched_eval_resident");
            gpu_evals++;
            if ((size_t)res.n_scored == window) {
                neighbours_scored += (uint64_t)res.n_scored;
                const ls_sched_move m = stream[(size_t)res.best_index];
                out.solution = current.solution;
                if (m.type == 0) out.solution[(size_t)m.a] = m.b;
                else {
                    out.solution[(size_t)m.a] = current.solution[(size_t)m.b];
                    out.solution[(size_t)m.b] = current.solution[(size_t)m.a];
                }
                out.score = Score(res.best_hard, res.best_soft);
                return true;
            }
            if (want > 64 * window + (1u << 16))
                throw std::runtime_error("the random move stream cannot fill the window (the reference would not terminate)");
            want *= 2;
        }
    }

    // SchedulePerturbation (lib.rs:582-612), default table (:572-579): DoNothing 10, ChangeDaysSubsetRandomly 100
    template <class R> Solution propose_new_starting_solution(const SS &current, bool current_is_in_best_set, R &rng) const
    {
        const size_t strategy = choose_weighted_index(rng, {10, 100}, /*wide=*/true);
        Solution next = current.solution;
        if (strategy == 0) return next;
        const uint64_t total_days = next.size();
        auto clamp1 = [&](uint64_t x) { return x < 1 ? (uint64_t)1 : (x > total_days ? total_days : x); };
        const uint64_t k = current_is_in_best_set ? gen_range_u64_incl(rng, 1, clamp1(total_days / 20))
                                                  : gen_range_u64_incl(rng, 1, clamp1(total_days / 2));
        std::vector<uint64_t> idx(total_days);
        for (uint64_t i = 0; i < total_days; i++) idx[i] = i;
        shuffle(rng, idx);
        for (uint64_t i = 0; i < k; i++) next[(size_t)idx[i]] = (int32_t)choose_index(rng, (size_t)n_employees);
        return next;
    }
};

} // namespace lshost
