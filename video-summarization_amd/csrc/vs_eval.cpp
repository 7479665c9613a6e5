// vs_eval.cpp — host C++ port of the reference keyshot evaluation (include/vs_eval.h).
// Follows reference src/evaluation/{compute_metrics,generate_summary,knapsack_implementation,
// evaluation_metrics,compute_correlation}.py; arithmetic that decides a selection is reproduced bit for
// bit (numpy's float32 pairwise sum for shot means, Python-float (double) knapsack table).
#include "vs_eval.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>

#include "vs_scorer.h"

int vs_fail_msg(int code, const char *msg);     // vs_scorer.cpp: sets the thread-local error text

namespace {

int bad(const char *msg) { return vs_fail_msg(VS_ERR_INVALID, msg); }

// numpy's pairwise summation of a contiguous float32 run (numpy/core/src/umath/loops_utils.h.src,
// pairwise_sum: <8 plain loop; <=128 eight strided partial sums; else split at a multiple of 8).
float np_pairwise_sum_f32(const float *a, ptrdiff_t n) {
    if (n < 8) {
        float res = 0.f;
        for (ptrdiff_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        ptrdiff_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    ptrdiff_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}

int upsample(const float *scores, int n_scores, const int32_t *positions, int n_positions, int n_frames,
             float *out) {
    if (n_frames < 0 || n_positions < 1 || n_scores < 0) return bad("upsample: bad sizes");
    std::fill(out, out + n_frames, 0.f);
    // positions (+ n_frames when the last entry differs): compute_metrics.py:30-32
    const int np_ = n_positions + (positions[n_positions - 1] != n_frames ? 1 : 0);
    auto pos = [&](int i) { return i < n_positions ? positions[i] : n_frames; };
    for (int i = 0; i + 1 < np_; ++i) {
        int lo = pos(i), hi = pos(i + 1);
        lo = std::max(0, std::min(lo, n_frames));
        hi = std::max(0, std::min(hi, n_frames));           // numpy slices clip silently
        if (i > n_scores) return bad("upsample: more segments than scores + 1");     // IndexError in the reference
        const float v = (i == n_scores) ? 0.f : scores[i];  // :34-37
        for (int f = lo; f < hi; ++f) out[f] = v;
    }
    return VS_OK;
}

// returns false where the reference raises IndexError (NaN values can walk the capacity below -(W+1))
bool knapsack(int W, const int32_t *wt, const double *val, int n, std::vector<int32_t> &sel) {
    // K[i][w] exactly as knapsack_implementation.py:11-21 (double == Python float)
    std::vector<double> K((size_t)(n + 1) * (W + 1), 0.0);
    for (int i = 1; i <= n; ++i) {
        const double *prev = &K[(size_t)(i - 1) * (W + 1)];
        double *cur = &K[(size_t)i * (W + 1)];
        const int w_i = wt[i - 1];
        const double v_i = val[i - 1];
        for (int w = 1; w <= W; ++w) {
            if (w_i <= w) {
                const double take = v_i + prev[w - w_i];
                cur[w] = prev[w] > take ? prev[w] : take;      // Python max(a, b): a unless b > a - also when either is NaN (a shot past n_frames has a NaN mean)
            } else {
                cur[w] = prev[w];
            }
        }
    }
    sel.clear();
    int w = W;
    for (int i = n; i > 0; --i) {                               // :23-28
        // A NaN entry differs from everything, so the reference can "select" a shot that does not fit and carry a
        // NEGATIVE capacity on; its K[i][w] then indexes from the end of the row (Python list semantics).
        if (w < -(W + 1)) return false;
        const int col = w < 0 ? w + W + 1 : w;
        if (K[(size_t)i * (W + 1) + col] != K[(size_t)(i - 1) * (W + 1) + col]) {
            sel.push_back(i - 1);
            w -= wt[i - 1];
        }
    }
    std::reverse(sel.begin(), sel.end());
    return true;
}

// ---- rank correlations on RUN-LENGTH-COMPRESSED frame vectors ----
// Both inputs of evaluate_scores are piecewise constant: the prediction is up-sampled from one score per pick (15 frames,
// compute_metrics.py:30-37) and the users' importance scores are per-shot integers (TVSum: 1..5 over 2-second shots).  A
// frame vector is therefore held as runs (start, value), ranks are taken over run VALUES with the run lengths as weights,
// and Kendall's pair counts / Spearman's sums become weighted sums over the joint runs of the two vectors.  Every pair
// count is an exact integer and every Spearman sum an exact multiple of 1/4 (far below 2^53), so the results are the
// ones of the per-frame computation (scipy.stats.rankdata(-x, 'average'), kendalltau variant 'b', np.corrcoef) bit for
// bit; vectors without runs cost what the per-frame form costs.
struct Ranked {
    int n = 0;                          // frames
    std::vector<int> start;             // run r covers frames [start[r], start[r+1]); start.back() == n
    std::vector<int> dense;             // per run: index of its value among the distinct values, largest value first (rank order)
    std::vector<long long> gweight;     // per distinct value: frames holding it
    std::vector<double> grank;          // per distinct value: the average rank of those frames
};

// per-thread work vectors: they keep their capacity from task to task (a fresh std::vector per task grows its thread's
// malloc arena by system calls, which serialise on the process' address-space lock: measured, no speed-up at all from
// 8 threads before this)
struct Scratch {
    std::vector<double> val;
    std::vector<int> order, cnt, cx;
    std::vector<long long> bit;
    struct Seg { long long w; int x, y; };
    std::vector<Seg> seg, tmp;
};

template <class T>
void rank_runs(const T *x, int n, Ranked &R, Scratch &W) {
    R.n = n;
    R.start.clear();
    std::vector<double> &val = W.val;
    val.clear();
    for (int i = 0; i < n; ++i)
        if (i == 0 || !((double)x[i] == (double)x[i - 1])) { R.start.push_back(i); val.push_back((double)x[i]); }
    const int m = (int)val.size();
    R.start.push_back(n);
    std::vector<int> &order = W.order;
    order.resize(m);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return val[a] > val[b]; });      // rankdata(-x): largest first
    R.dense.assign(m, 0);
    R.gweight.clear();
    for (int k = 0; k < m; ++k) {
        const int r = order[k];
        if (k == 0 || !(val[r] == val[order[k - 1]])) R.gweight.push_back(0);
        R.dense[r] = (int)R.gweight.size() - 1;
        R.gweight.back() += R.start[r + 1] - R.start[r];
    }
    R.grank.resize(R.gweight.size());
    long long before = 0;
    for (size_t g = 0; g < R.gweight.size(); ++g) {
        R.grank[g] = (double)before + 0.5 * (double)(R.gweight[g] + 1);          // average of before+1 .. before+c
        before += R.gweight[g];
    }
}

// mean Kendall tau-b and Spearman rho ingredients for one (prediction, user) pair
void correlate(const Ranked &X, const Ranked &Y, double &tau, double &rho, Scratch &W) {
    const int n = X.n;
    const int gx = (int)X.gweight.size(), gy = (int)Y.gweight.size();
    // joint runs: (weight, x group, y group)
    typedef Scratch::Seg Seg;
    std::vector<Seg> &seg = W.seg, &tmp = W.tmp;
    seg.clear();
    for (size_t rx = 0, ry = 0; rx < X.dense.size() && ry < Y.dense.size();) {
        const int lo = std::max(X.start[rx], Y.start[ry]), hi = std::min(X.start[rx + 1], Y.start[ry + 1]);
        if (hi > lo) seg.push_back(Seg{hi - lo, X.dense[rx], Y.dense[ry]});
        if (X.start[rx + 1] <= Y.start[ry + 1]) ++rx; else ++ry;
    }
    const int m = (int)seg.size();
    // stable counting sorts: by y group, then by x group -> x ascending, y ascending inside an x group
    tmp.resize(m);
    {
        std::vector<int> &cnt = W.cnt, &cx = W.cx;
        cnt.assign(gy + 1, 0);
        for (const Seg &q : seg) ++cnt[q.y + 1];
        for (int g = 0; g < gy; ++g) cnt[g + 1] += cnt[g];
        for (const Seg &q : seg) tmp[cnt[q.y]++] = q;
        cx.assign(gx + 1, 0);
        for (const Seg &q : tmp) ++cx[q.x + 1];
        for (int g = 0; g < gx; ++g) cx[g + 1] += cx[g];
        for (const Seg &q : tmp) seg[cx[q.x]++] = q;
    }
    auto pairs = [](long long c) { return c * (c - 1) / 2; };
    const long long tot = pairs(n);
    long long xtie = 0, ytie = 0, ntie = 0, dis = 0;
    for (long long c : X.gweight) xtie += pairs(c);
    for (long long c : Y.gweight) ytie += pairs(c);
    for (int i = 0; i < m;) {
        int k = i;
        long long c = 0;
        while (k < m && seg[k].x == seg[i].x && seg[k].y == seg[i].y) c += seg[k++].w;
        ntie += pairs(c);
        i = k;
    }
    // discordant pairs: x groups in ascending order; a frame pair is discordant when the later x group holds the smaller y
    // group.  Fenwick tree over y groups holding the weight seen so far.
    std::vector<long long> &bit = W.bit;
    bit.assign(gy + 1, 0);
    long long seen = 0;
    auto add = [&](int y, long long w) { for (int i = y + 1; i <= gy; i += i & -i) bit[i] += w; };
    auto upto = [&](int y) { long long t = 0; for (int i = y + 1; i > 0; i -= i & -i) t += bit[i]; return t; };     // weight with group <= y
    for (int i = 0; i < m;) {
        int k = i;
        while (k < m && seg[k].x == seg[i].x) { dis += seg[k].w * (seen - upto(seg[k].y)); ++k; }
        for (int q = i; q < k; ++q) { add(seg[q].y, seg[q].w); seen += seg[q].w; }
        i = k;
    }
    if (xtie == tot || ytie == tot) tau = NAN;
    else {
        const double con_minus_dis = (double)(tot - xtie - ytie + ntie - 2 * dis);
        tau = std::min(1.0, std::max(-1.0, con_minus_dis / std::sqrt((double)(tot - xtie)) / std::sqrt((double)(tot - ytie))));
    }
    // Spearman = Pearson of the average ranks (np.corrcoef): both means are (n + 1) / 2
    const double mean = 0.5 * (double)(n + 1);
    double sab = 0, saa = 0, sbb = 0;
    for (const Seg &q : seg) sab += (double)q.w * (X.grank[q.x] - mean) * (Y.grank[q.y] - mean);
    for (int g = 0; g < gx; ++g) saa += (double)X.gweight[g] * (X.grank[g] - mean) * (X.grank[g] - mean);
    for (int g = 0; g < gy; ++g) sbb += (double)Y.gweight[g] * (Y.grank[g] - mean) * (Y.grank[g] - mean);
    rho = sab / std::sqrt(saa * sbb);
}

}  // namespace

extern "C" {

int vs_eval_upsample(const float *scores, int32_t n_scores, const int32_t *positions, int32_t n_positions,
                     int32_t n_frames, float *frame_scores) {
    if (!scores || !positions || !frame_scores) return bad("NULL pointer");
    return upsample(scores, n_scores, positions, n_positions, n_frames, frame_scores);
}

int vs_eval_knapsack(int32_t W, const int32_t *wt, const double *val, int32_t n, int32_t *selected,
                     int32_t *n_selected) {
    if (!wt || !val || !selected || !n_selected || W < 0 || n < 0) return bad("knapsack: bad arguments");
    for (int i = 0; i < n; ++i) if (wt[i] < 0) return bad("knapsack: negative weight");
    std::vector<int32_t> sel;
    if (!knapsack(W, wt, val, n, sel)) return bad("knapsack: capacity index out of range (IndexError in the reference)");
    std::copy(sel.begin(), sel.end(), selected);
    *n_selected = (int32_t)sel.size();
    return VS_OK;
}

int vs_eval_generate_summary(const float *scores, int32_t n_scores, const int32_t *positions,
                             int32_t n_positions, int32_t n_frames, const int32_t *change_points,
                             int32_t n_shots, int8_t *summary, int32_t summary_len) {
    if (!scores || !positions || !change_points || !summary || n_shots < 1) return bad("generate_summary: bad arguments");
    const int last_end = change_points[2 * (n_shots - 1) + 1];
    if (summary_len != last_end + 1) return bad("generate_summary: summary_len must be last_shot_end + 1");
    std::vector<float> fs((size_t)std::max(n_frames, 0));
    if (int rc = upsample(scores, n_scores, positions, n_positions, n_frames, fs.data())) return rc;
    std::vector<int32_t> len(n_shots);
    std::vector<double> imp(n_shots);
    for (int s = 0; s < n_shots; ++s) {
        const int a = change_points[2 * s], b = change_points[2 * s + 1];
        len[s] = b - a + 1;                                                   // generate_summary.py:41
        const int lo = std::max(0, std::min(a, n_frames)), hi = std::max(lo, std::min(b + 1, n_frames));
        // float32 mean as numpy: pairwise float32 sum, float32 divide (empty slice -> nan)   :42
        const float m = hi > lo ? np_pairwise_sum_f32(fs.data() + lo, hi - lo) / (float)(hi - lo) : NAN;
        imp[s] = (double)m;
        if (len[s] < 0) return bad("generate_summary: shot with negative length");
    }
    const int W = (int)((double)(last_end + 1) * 0.15);                       // :46
    std::vector<int32_t> sel;
    if (!knapsack(W, len.data(), imp.data(), n_shots, sel)) return bad("generate_summary: capacity index out of range (IndexError in the reference)");
    std::fill(summary, summary + summary_len, (int8_t)0);
    for (int s : sel) {
        const int a = std::max(0, change_points[2 * s]), b = std::min(summary_len - 1, change_points[2 * s + 1]);
        for (int f = a; f <= b; ++f) summary[f] = 1;
    }
    return VS_OK;
}

int vs_eval_fscore(const int8_t *summary, int32_t summary_len, const int8_t *user_summary, int32_t n_users,
                   int32_t user_len, int32_t use_max, double *f_score) {
    if (!summary || !user_summary || !f_score || n_users < 1 || summary_len < 0 || user_len < 0) return bad("fscore: bad arguments");
    const int L = std::max(summary_len, user_len);                           // evaluation_metrics.py:12
    long long sumS = 0;
    for (int i = 0; i < summary_len; ++i) sumS += summary[i];
    double acc = 0.0, best = -INFINITY;
    for (int u = 0; u < n_users; ++u) {
        const int8_t *g = user_summary + (size_t)u * user_len;
        long long sumG = 0, ov = 0;
        for (int i = 0; i < user_len; ++i) sumG += g[i];
        for (int i = 0; i < std::min(summary_len, user_len); ++i) ov += (summary[i] & g[i]);
        (void)L;
        const double precision = (double)ov / (double)sumS, recall = (double)ov / (double)sumG;   // :23-24
        const double f = (precision + recall == 0) ? 0.0 : 2 * precision * recall * 100 / (precision + recall);
        acc += f;
        best = std::max(best, f);
    }
    *f_score = use_max ? best : acc / n_users;                                // :30-33
    return VS_OK;
}

int vs_eval_rank_correlation(const float *frame_scores, int32_t n, const double *user_scores, int32_t n_users,
                             double *kendall, double *spearman) {
    if (!frame_scores || !user_scores || !kendall || !spearman || n < 2 || n_users < 1) return bad("rank_correlation: bad arguments");
    Ranked X, Y;
    Scratch W;
    rank_runs(frame_scores, n, X, W);
    double ksum = 0, ssum = 0;
    for (int u = 0; u < n_users; ++u) {          // users in order: the sums do not depend on any scheduling (vs_eval_corpus: the parallel form)
        double kt, sp;
        rank_runs(user_scores + (size_t)u * n, n, Y, W);
        correlate(X, Y, kt, sp, W);                  // compute_correlation.py:9-14
        ksum += kt; ssum += sp;
    }
    *kendall = ksum / n_users;
    *spearman = ssum / n_users;
    return VS_OK;
}

int vs_eval_corpus(const vs_eval_video *videos, int32_t n_videos, int32_t max_threads, double *f_score, double *kendall,
                   double *spearman) {
    if (n_videos < 0 || (n_videos > 0 && (!videos || !f_score || !kendall || !spearman))) return bad("eval_corpus: bad arguments");
    if (n_videos == 0) return VS_OK;
    for (int v = 0; v < n_videos; ++v) {
        const vs_eval_video &V = videos[v];
        if (!V.scores || !V.positions || !V.change_points || !V.user_summary || V.n_shots < 1 || V.n_users < 1 || V.n_frames < 0)
            return bad("eval_corpus: a video record has a NULL pointer or an empty field");
        if (V.user_scores && (V.n_score_users < 1 || V.n_frames < 2)) return bad("eval_corpus: user_scores needs n_score_users >= 1 and n_frames >= 2");
    }
    const int hw = (int)std::thread::hardware_concurrency();
    int nth = max_threads > 0 ? max_threads : std::min(hw > 0 ? hw : 1, 32);
    // phase 2 tasks: (video, user) pairs, heaviest videos first (a long video's users should not start last)
    struct Task { int v, u; };
    std::vector<Task> tasks;
    std::vector<int> order(n_videos);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return videos[a].n_frames > videos[b].n_frames; });
    std::vector<size_t> uoff(n_videos + 1, 0);
    for (int v = 0; v < n_videos; ++v) uoff[v + 1] = uoff[v] + (videos[v].user_scores ? (size_t)videos[v].n_score_users : 0);
    for (int v : order)
        if (videos[v].user_scores)
            for (int u = 0; u < videos[v].n_score_users; ++u) tasks.push_back(Task{v, u});
    nth = std::max(1, std::min<int>(nth, (int)std::max<size_t>(tasks.size(), (size_t)n_videos)));
    std::vector<Ranked> pr(n_videos);                            // prediction ranks per video (phase 1 -> phase 2)
    std::vector<double> kt(uoff[n_videos]), sp(uoff[n_videos]);
    std::vector<int> rc(n_videos, VS_OK);
    std::atomic<int> next1{0}, next2{0};
    auto phase1 = [&]() {
        Scratch W;
        std::vector<float> fs;
        std::vector<int8_t> summary;
        for (;;) {
            const int i = next1.fetch_add(1);
            if (i >= n_videos) break;
            const int v = order[i];
            const vs_eval_video &V = videos[v];
            const int last_end = V.change_points[2 * (V.n_shots - 1) + 1];
            if (last_end < 0) { rc[v] = VS_ERR_INVALID; continue; }
            summary.resize((size_t)last_end + 1);
            rc[v] = vs_eval_generate_summary(V.scores, V.n_scores, V.positions, V.n_positions, V.n_frames, V.change_points, V.n_shots,
                                             summary.data(), last_end + 1);
            if (rc[v] == VS_OK)
                rc[v] = vs_eval_fscore(summary.data(), last_end + 1, V.user_summary, V.n_users, V.user_len, V.use_max, &f_score[v]);
            if (rc[v] == VS_OK && V.user_scores) {
                fs.resize((size_t)V.n_frames);
                rc[v] = upsample(V.scores, V.n_scores, V.positions, V.n_positions, V.n_frames, fs.data());
                if (rc[v] == VS_OK) rank_runs(fs.data(), V.n_frames, pr[v], W);
            }
        }
    };
    auto phase2 = [&]() {
        Ranked ur;
        Scratch W;
        for (;;) {
            const int i = next2.fetch_add(1);
            if (i >= (int)tasks.size()) break;
            const Task t = tasks[i];
            if (rc[t.v] != VS_OK) continue;
            const vs_eval_video &V = videos[t.v];
            if (V.user_scores_f32) rank_runs((const float *)V.user_scores + (size_t)t.u * V.n_frames, V.n_frames, ur, W);
            else rank_runs((const double *)V.user_scores + (size_t)t.u * V.n_frames, V.n_frames, ur, W);
            correlate(pr[t.v], ur, kt[uoff[t.v] + t.u], sp[uoff[t.v] + t.u], W);
        }
    };
    auto run = [&](auto &fn) {
        if (nth == 1) { fn(); return; }
        std::vector<std::thread> pool;
        for (int t = 0; t < nth; ++t) pool.emplace_back(fn);
        for (auto &th : pool) th.join();
    };
    auto T0 = std::chrono::steady_clock::now();
    run(phase1);
    auto T1 = std::chrono::steady_clock::now();
    for (int v = 0; v < n_videos; ++v)
        if (rc[v] != VS_OK) return bad("eval_corpus: a video failed in generate_summary / evaluate_summary (see the per-video entry points for the cause)");
    run(phase2);
    auto T2 = std::chrono::steady_clock::now();
    if (getenv("VS_EVAL_DEBUG")) fprintf(stderr, "phase1 %.2f ms phase2 %.2f ms nth %d tasks %zu\n", std::chrono::duration<double, std::milli>(T1 - T0).count(), std::chrono::duration<double, std::milli>(T2 - T1).count(), nth, tasks.size());
    for (int v = 0; v < n_videos; ++v) {
        if (!videos[v].user_scores) { kendall[v] = NAN; spearman[v] = NAN; continue; }
        double ks = 0, ss = 0;
        for (int u = 0; u < videos[v].n_score_users; ++u) { ks += kt[uoff[v] + u]; ss += sp[uoff[v] + u]; }
        kendall[v] = ks / videos[v].n_score_users;
        spearman[v] = ss / videos[v].n_score_users;
    }
    return VS_OK;
}

}  // extern "C"
