/*
 * vs_eval.h — C ABI of the keyshot evaluation that consumes the scorer's output (host code, no GPU).
 * the step right after the scorer in the reference's val_step
 * (reference src/train.py:150 -> src/evaluation/compute_metrics.py:42 eval_metrics).
 * All pointers are HOST pointers.  Every function returns 0 or a VS_ERR_* status (vs_scorer.h) and sets
 * vs_last_error().  Arithmetic follows the reference bit for bit where it decides a selection
 * (float32 pairwise shot means as numpy computes them, double knapsack table as Python floats).
 */
#ifndef VS_EVAL_H
#define VS_EVAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces: upsample() (compute_metrics.py:19-39; same loop in generate_summary.py:25-35).
 * frame_scores[n_frames] = scores[i] on [positions[i], positions[i+1]), n_frames appended to positions
 * when its last entry differs; the segment after the last score is 0. */
int vs_eval_upsample(const float *scores, int32_t n_scores, const int32_t *positions, int32_t n_positions,
                     int32_t n_frames, float *frame_scores);

/* Replaces: knapSack(W, wt, val, n) (knapsack_implementation.py:1-30): 0/1 knapsack over shots, table
 * in double, ties resolved like the reference (take when val + K[i-1][w-wt] >= K[i-1][w]; back-track on
 * K[i][w] != K[i-1][w]).  selected: out, capacity n; n_selected: out.  Indices ascending. */
int vs_eval_knapsack(int32_t W, const int32_t *wt, const double *val, int32_t n,
                     int32_t *selected, int32_t *n_selected);

/* Replaces: generate_summary() for one video (generate_summary.py:17-55): upsample, float32 shot means,
 * budget int((last_shot_end+1)*0.15), knapsack, binary summary of length last_shot_end+1.
 * change_points [n_shots][2] inclusive ends.  summary: out int8 [last_shot_end+1]. */
int vs_eval_generate_summary(const float *scores, int32_t n_scores, const int32_t *positions,
                             int32_t n_positions, int32_t n_frames, const int32_t *change_points,
                             int32_t n_shots, int8_t *summary, int32_t summary_len);

/* Replaces: evaluate_summary(pred, user_summary, eval_method) (evaluation_metrics.py:4-33).
 * user_summary [n_users][user_len] 0/1; use_max != 0 -> 'max' (SumMe) else 'avg' (TVSum). */
int vs_eval_fscore(const int8_t *summary, int32_t summary_len, const int8_t *user_summary,
                   int32_t n_users, int32_t user_len, int32_t use_max, double *f_score);

/* Replaces: evaluate_scores(frame_scores, user_scores) (compute_correlation.py:4-15): mean over users of
 * Kendall tau-b and Spearman rho between rankdata(-frame_scores) and rankdata(-user_scores[u]).
 * user_scores [n_users][n]. */
int vs_eval_rank_correlation(const float *frame_scores, int32_t n, const double *user_scores,
                             int32_t n_users, double *kendall, double *spearman);

/* One video's inputs to the evaluation: the scorer's output and the user record of the reference's
 * UserSummaries (data/dataset.py:146-154).  user_scores may be NULL (no rank correlation for that video: NaN). */
typedef struct vs_eval_video {
    const float *scores;          /* [n_scores] per sub-sampled frame */
    const int32_t *positions;     /* [n_positions] picks */
    const int32_t *change_points; /* [n_shots][2] inclusive ends */
    const int8_t *user_summary;   /* [n_users][user_len] 0/1 */
    const void *user_scores;      /* [n_score_users][n_frames] double (or float, see below), or NULL */
    int32_t n_scores, n_positions, n_frames, n_shots, n_users, user_len, n_score_users;
    int32_t use_max;              /* != 0: 'max' protocol (SumMe), else 'avg' (TVSum) */
    int32_t user_scores_f32;      /* != 0: user_scores is float (the datasets' own type: no widened copy, half the bytes) */
} vs_eval_video;

/* Replaces: the loop of eval_metrics(data, user_dict) (compute_metrics.py:42-92) over a shard's videos: per video
 * generate_summary -> evaluate_summary, upsample -> evaluate_scores.  All videos run on ONE pool of at most max_threads
 * host threads (<= 0: the hardware's count, capped at 32) that first takes the per-video work (summary, F-score,
 * prediction ranks) and then the (video, user) rank correlations - no nested pools, no per-video thread start.  Per-user
 * results are summed in user order: the outputs do not depend on the scheduling.
 * f_score, kendall, spearman: out, [n_videos] each.  A video whose inputs are invalid fails the whole call. */
int vs_eval_corpus(const vs_eval_video *videos, int32_t n_videos, int32_t max_threads,
                   double *f_score, double *kendall, double *spearman);

#ifdef __cplusplus
}
#endif
#endif /* VS_EVAL_H */
