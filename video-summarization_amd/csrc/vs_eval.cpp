// vs_eval.cpp — host C++ port of the reference keyshot evaluation (include/vs_eval.h).
// Follows reference src/evaluation/{compute_metrics,generate_summary,knapsack_implementation,
// evaluation_metrics,compute_correlation}.py; arithmetic that decides a selection is reproduced bit for
// bit (numpy's float32 pairwise sum for shot means, Python-float (double) knapsack table).
#include "vs_eval.h"

#include <algorithm>
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

// scipy.stats.rankdata(-x) with method 'average', as doubles
template <class T>
void rank_neg_average(const T *x, int n, std::vector<double> &rk) {
    std::vector<std::pair<double, int>> kv(n);          // (key, index) pairs sort ~3x faster than indirect compares
    for (int i = 0; i < n; ++i) kv[i] = {-(double)x[i], i};
    std::sort(kv.begin(), kv.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
    rk.assign(n, 0.0);
    for (int i = 0; i < n;) {
        int j = i;
        while (j + 1 < n && kv[j + 1].first == kv[i].first) ++j;
        const double r = 0.5 * ((i + 1) + (j + 1));     // average rank of the tie group (order inside it is irrelevant)
        for (int k = i; k <= j; ++k) rk[kv[k].second] = r;
        i = j + 1;
    }
}

// Pearson r of two rank vectors, the way np.corrcoef computes it (means, centred products, in double)
double pearson(const std::vector<double> &a, const std::vector<double> &b) {
    const int n = (int)a.size();
    double ma = 0, mb = 0;
    for (int i = 0; i < n; ++i) { ma += a[i]; mb += b[i]; }
    ma /= n; mb /= n;
    double sab = 0, saa = 0, sbb = 0;
    for (int i = 0; i < n; ++i) {
        const double da = a[i] - ma, db = b[i] - mb;
        sab += da * db; saa += da * da; sbb += db * db;
    }
    return sab / std::sqrt(saa * sbb);
}

// number of discordant pairs of (x, y), x sorted ascending with ties broken by y (merge-sort inversion count)
long long count_discordant(std::vector<int> &y) {
    const int n = (int)y.size();
    std::vector<int> tmp(n);
    long long inv = 0;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            const int mid = std::min(lo + width, n), hi = std::min(lo + 2 * width, n);
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (y[j] < y[i]) { tmp[k++] = y[j++]; inv += mid - i; }
                else tmp[k++] = y[i++];
            }
            while (i < mid) tmp[k++] = y[i++];
            while (j < hi) tmp[k++] = y[j++];
            std::copy(tmp.begin() + lo, tmp.begin() + hi, y.begin() + lo);
        }
    }
    return inv;
}

// scipy.stats.kendalltau (variant 'b') on two rank vectors
double kendall_tau_b(const std::vector<double> &xr, const std::vector<double> &yr) {
    const int n = (int)xr.size();
    if (n < 2) return NAN;
    // one sort by (x, y) gives x ascending with ties in x ordered by y (what scipy's two stable sorts build);
    // ranks are multiples of 0.5, so 2*rank is an exact integer key
    std::vector<std::pair<long long, int>> kv(n);
    for (int i = 0; i < n; ++i) kv[i] = {(long long)(2.0 * xr[i]) * (4LL * n + 4) + (long long)(2.0 * yr[i]), i};
    std::sort(kv.begin(), kv.end());
    std::vector<int> xs(n), ys(n);
    {
        std::vector<std::pair<long long, int>> ykv(n);
        for (int i = 0; i < n; ++i) ykv[i] = {(long long)(2.0 * yr[i]), i};
        std::sort(ykv.begin(), ykv.end());
        std::vector<int> ydense(n);
        int d = 0;
        for (int i = 0; i < n; ++i) { if (i > 0 && ykv[i].first != ykv[i - 1].first) ++d; ydense[ykv[i].second] = d; }
        d = 0;
        for (int i = 0; i < n; ++i) {
            if (i > 0 && xr[kv[i].second] != xr[kv[i - 1].second]) ++d;
            xs[i] = d; ys[i] = ydense[kv[i].second];
        }
    }
    auto tie_pairs = [&](const std::vector<int> &v_sorted) {
        long long t = 0;
        for (int i = 0; i < n;) { int j = i; while (j + 1 < n && v_sorted[j + 1] == v_sorted[i]) ++j; const long long c = j - i + 1; t += c * (c - 1) / 2; i = j + 1; }
        return t;
    };
    long long ntie = 0;     // joint ties: runs equal in both x and y (adjacent after the double sort)
    for (int i = 0; i < n;) { int j = i; while (j + 1 < n && xs[j + 1] == xs[i] && ys[j + 1] == ys[i]) ++j; const long long c = j - i + 1; ntie += c * (c - 1) / 2; i = j + 1; }
    const long long xtie = tie_pairs(xs);
    std::vector<int> ysorted(ys);
    std::sort(ysorted.begin(), ysorted.end());
    const long long ytie = tie_pairs(ysorted);
    std::vector<int> ycopy(ys);
    const long long dis = count_discordant(ycopy);
    const long long tot = (long long)n * (n - 1) / 2;
    if (xtie == tot || ytie == tot) return NAN;
    const double con_minus_dis = (double)(tot - xtie - ytie + ntie - 2 * dis);
    double tau = con_minus_dis / std::sqrt((double)(tot - xtie)) / std::sqrt((double)(tot - ytie));
    return std::min(1.0, std::max(-1.0, tau));
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
    std::vector<double> pr;
    rank_neg_average(frame_scores, n, pr);
    // users are independent: a few host threads, results summed in user order (deterministic)
    std::vector<double> kt(n_users), sp(n_users);
    auto work = [&](int u0, int u1) {
        std::vector<double> ur;
        for (int u = u0; u < u1; ++u) {
            rank_neg_average(user_scores + (size_t)u * n, n, ur);
            sp[u] = pearson(pr, ur);                                          // compute_correlation.py:9-11
            kt[u] = kendall_tau_b(pr, ur);                                    // :12-14
        }
    };
    const int hw = (int)std::thread::hardware_concurrency();
    const int nth = std::max(1, std::min({n_users, hw > 0 ? hw : 1, 8}));
    if (nth == 1 || (size_t)n * n_users < 20000) {
        work(0, n_users);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nth; ++t) pool.emplace_back(work, (int)((long long)n_users * t / nth), (int)((long long)n_users * (t + 1) / nth));
        for (auto &th : pool) th.join();
    }
    double ksum = 0, ssum = 0;
    for (int u = 0; u < n_users; ++u) { ksum += kt[u]; ssum += sp[u]; }
    *kendall = ksum / n_users;
    *spearman = ssum / n_users;
    return VS_OK;
}

}  // extern "C"
