/* train_demo.c — one training step through include/vs_train.h from plain C: no Python, no torch.
 * Build like score_demo.c (tests/test_cabi_c.py does it and runs it on the GPU box).
 * Random small model (d 128, 2 heads of 64, 2 layers), padded batch + key mask, then:
 *   1. vs_train_forward without dropout == vs_scorer_forward (logits within 1e-4: different kernels, same function);
 *   2. masked-MSE loss forward/backward (vs_mse_mask_loss_*), vs_train_backward: every gradient finite,
 *      d final_layer.bias == sum of d_scores (exactly what the chain rule says), d_x of padded frames == 0;
 *   3. a central-difference check of the loss along the gradient direction of final_layer.weight (one parameter
 *      tensor perturbed through vs_weights_update): (L(w + h g) - L(w - h g)) / 2h ~= |g|^2 within 2 %;
 *   4. the same step with dropout 0.3 run twice: bit-identical gradients (fixed seed, ordered reductions).
 * Prints "OK". */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vs_train.h"

static uint32_t rng_state = 2468u;
static float rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 8388608.0f - 1.0f; }
static float *dev_random(size_t n, float scale, float offset) {
    float *h = (float *)malloc(n * sizeof(float)), *d = NULL;
    for (size_t i = 0; i < n; ++i) h[i] = rnd() * scale + offset;
    if (hipMalloc((void **)&d, n * sizeof(float)) != hipSuccess) exit(2);
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    free(h);
    return d;
}
static float *dev_alloc(size_t n) { float *d = NULL; if (hipMalloc((void **)&d, n * sizeof(float)) != hipSuccess) exit(2); return d; }
static float *to_host(const float *d, size_t n) { float *h = (float *)malloc(n * 4); hipDeviceSynchronize(); hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost); return h; }
#define CHECK(call) do { int rc_ = (call); if (rc_ != VS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vs_last_error()); return 1; } } while (0)
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

enum { D = 128, H = 2, L = 2, DIN = 1024, MAXLEN = 2000, B = 2, T = 90 };

int main(void) {
    const int lengths[B] = {90, 57};
    vs_layer_params layers[L];
    vs_layer_grads glayers[L];
    for (int l = 0; l < L; ++l) {
        const float s = 1.0f / 11.0f;
        layers[l].wq = dev_random(D * D, s, 0); layers[l].bq = dev_random(D, 0.1f, 0);
        layers[l].wk = dev_random(D * D, s, 0); layers[l].bk = dev_random(D, 0.1f, 0);
        layers[l].wv = dev_random(D * D, s, 0); layers[l].bv = dev_random(D, 0.1f, 0);
        layers[l].wo = dev_random(D * D, s, 0); layers[l].bo = dev_random(D, 0.1f, 0);
        layers[l].ln1_g = dev_random(D, 0.1f, 1.0f); layers[l].ln1_b = dev_random(D, 0.1f, 0);
        layers[l].w1 = dev_random(4 * D * D, s, 0); layers[l].b1 = dev_random(4 * D, 0.1f, 0);
        layers[l].w2 = dev_random(4 * D * D, s / 2, 0); layers[l].b2 = dev_random(D, 0.1f, 0);
        layers[l].ln2_g = dev_random(D, 0.1f, 1.0f); layers[l].ln2_b = dev_random(D, 0.1f, 0);
        glayers[l].wq = dev_alloc(D * D); glayers[l].bq = dev_alloc(D); glayers[l].wk = dev_alloc(D * D); glayers[l].bk = dev_alloc(D);
        glayers[l].wv = dev_alloc(D * D); glayers[l].bv = dev_alloc(D); glayers[l].wo = dev_alloc(D * D); glayers[l].bo = dev_alloc(D);
        glayers[l].ln1_g = dev_alloc(D); glayers[l].ln1_b = dev_alloc(D); glayers[l].w1 = dev_alloc(4 * D * D); glayers[l].b1 = dev_alloc(4 * D);
        glayers[l].w2 = dev_alloc(4 * D * D); glayers[l].b2 = dev_alloc(D); glayers[l].ln2_g = dev_alloc(D); glayers[l].ln2_b = dev_alloc(D);
    }
    vs_model_params P;
    P.embed_w = dev_random((size_t)D * DIN, 1.0f / 32.0f, 0); P.embed_b = dev_random(D, 0.1f, 0);
    P.pos_embedding = dev_random((size_t)MAXLEN * D, 1.0f, 0);
    P.layers = layers;
    float *final_w = dev_random(D, 1.0f / 11.0f, 0);
    P.final_w = final_w; P.final_b = dev_random(1, 0.1f, 0);
    vs_model_grads G;
    G.embed_w = dev_alloc((size_t)D * DIN); G.embed_b = dev_alloc(D); G.layers = glayers; G.final_w = dev_alloc(D); G.final_b = dev_alloc(1);
    vs_model_desc desc = {D, H, L, DIN, MAXLEN, 1};
    vs_weights *w = NULL;
    CHECK(vs_weights_pack(&desc, &P, NULL, &w));

    float *hx = (float *)malloc((size_t)B * T * DIN * 4), htgt[B * T];
    uint8_t hmask[B * T];
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            hmask[b * T + t] = t >= lengths[b];
            htgt[b * T + t] = 0.5f + 0.5f * rnd();
            for (int c = 0; c < DIN; ++c) hx[((size_t)b * T + t) * DIN + c] = t < lengths[b] ? fabsf(rnd()) * 0.5f : 1000.0f;
        }
    float *x = dev_alloc((size_t)B * T * DIN), *tgt = dev_alloc(B * T), *scores = dev_alloc(B * T), *hidden = dev_alloc((size_t)B * T * D);
    float *ref = dev_alloc(B * T), *dsc = dev_alloc(B * T), *dx = dev_alloc((size_t)B * T * DIN), *loss = dev_alloc(1), *scratch = dev_alloc(256), *one = dev_alloc(1);
    uint8_t *mask = NULL; hipMalloc((void **)&mask, B * T);
    hipMemcpy(x, hx, (size_t)B * T * DIN * 4, hipMemcpyHostToDevice); hipMemcpy(mask, hmask, B * T, hipMemcpyHostToDevice);
    hipMemcpy(tgt, htgt, sizeof htgt, hipMemcpyHostToDevice);
    const float onef = 1.0f; hipMemcpy(one, &onef, 4, hipMemcpyHostToDevice);
    void *saved = NULL, *ws = NULL, *ws_s = NULL;
    const size_t nsaved = vs_train_saved_bytes(w, B, T), nws = vs_train_workspace_bytes(w, B, T), nws_s = vs_scorer_workspace_bytes(w, B, T);
    hipMalloc(&saved, nsaved); hipMalloc(&ws, nws); hipMalloc(&ws_s, nws_s);

    /* 1. train forward (no dropout) vs the scoring kernels */
    CHECK(vs_train_forward(w, x, mask, B, T, NULL, scores, hidden, saved, nsaved, ws, nws, NULL));
    CHECK(vs_scorer_forward(w, x, mask, B, T, 0, ref, NULL, ws_s, nws_s, NULL));
    float *hs = to_host(scores, B * T), *hr = to_host(ref, B * T);
    for (int i = 0; i < B * T; ++i)
        if (!hmask[i] && !(fabsf(hs[i] - hr[i]) < 1e-4f)) FAIL("train forward %g vs scoring forward %g at %d", hs[i], hr[i], i);

    /* 2. loss + backward */
    CHECK(vs_mse_mask_loss_forward(scores, tgt, mask, B * T, 1, scratch, loss, NULL));
    CHECK(vs_mse_mask_loss_backward(scores, tgt, mask, one, B * T, 1, dsc, NULL));
    CHECK(vs_train_backward(w, x, mask, B, T, NULL, dsc, NULL, saved, nsaved, &G, dx, ws, nws, NULL));
    float *hl = to_host(loss, 1), *hd = to_host(dsc, B * T), *gfb = to_host(G.final_b, 1), *gfw = to_host(G.final_w, D);
    double sum_d = 0.0, want_loss = 0.0, g2 = 0.0;
    for (int i = 0; i < B * T; ++i) { sum_d += hd[i]; if (!hmask[i]) want_loss += (double)(hs[i] - htgt[i]) * (hs[i] - htgt[i]); }
    want_loss /= B * T;
    if (!(fabs(hl[0] - want_loss) < 1e-5 * (1 + want_loss))) FAIL("loss %g, expected %g", hl[0], want_loss);
    if (!(fabs(gfb[0] - sum_d) < 1e-5 * (1 + fabs(sum_d)))) FAIL("d final_b %g != sum d_scores %g", gfb[0], sum_d);
    float *hdx = to_host(dx, (size_t)B * T * DIN), *gew = to_host(G.embed_w, (size_t)D * DIN), *gw1 = to_host(glayers[0].w1, 4 * D * D);
    for (size_t i = 0; i < (size_t)D * DIN; ++i) if (!isfinite(gew[i])) FAIL("d embed_w not finite");
    for (size_t i = 0; i < (size_t)4 * D * D; ++i) if (!isfinite(gw1[i])) FAIL("d fc1.weight not finite");
    for (int b = 0; b < B; ++b)
        for (int t = lengths[b]; t < T; ++t)
            for (int c = 0; c < DIN; ++c) if (hdx[((size_t)b * T + t) * DIN + c] != 0.0f) FAIL("input gradient of a padded frame is not zero");
    for (int i = 0; i < D; ++i) g2 += (double)gfw[i] * gfw[i];

    /* 3. central difference along the gradient of final_layer.weight (the loss is quadratic in that tensor) */
    {
        const float h = 1e-2f / (float)sqrt(g2 + 1e-30);
        float *hw = to_host(final_w, D), lpm[2];
        for (int sgn = 0; sgn < 2; ++sgn) {
            float hw2[D];
            for (int i = 0; i < D; ++i) hw2[i] = hw[i] + (sgn ? -h : h) * gfw[i];
            hipMemcpy(final_w, hw2, D * 4, hipMemcpyHostToDevice);
            CHECK(vs_weights_update(w, &P, NULL));
            CHECK(vs_train_forward(w, x, mask, B, T, NULL, scores, NULL, saved, nsaved, ws, nws, NULL));
            CHECK(vs_mse_mask_loss_forward(scores, tgt, mask, B * T, 1, scratch, loss, NULL));
            float *hl2 = to_host(loss, 1);
            lpm[sgn] = hl2[0];
        }
        hipMemcpy(final_w, hw, D * 4, hipMemcpyHostToDevice);
        CHECK(vs_weights_update(w, &P, NULL));
        const double fd = ((double)lpm[0] - lpm[1]) / (2.0 * h);
        if (!(fabs(fd - g2) < 0.02 * g2 + 1e-7)) FAIL("central difference %g vs |g|^2 %g", fd, g2);
    }

    /* 4. dropout: two identical steps give identical bits */
    {
        vs_dropout_cfg drop = {0.0f, 0.3f, 0x0123456789abcdefull};
        float *first = NULL;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(vs_train_forward(w, x, mask, B, T, &drop, scores, hidden, saved, nsaved, ws, nws, NULL));
            CHECK(vs_mse_mask_loss_backward(scores, tgt, mask, one, B * T, 1, dsc, NULL));
            CHECK(vs_train_backward(w, x, mask, B, T, &drop, dsc, NULL, saved, nsaved, &G, NULL, ws, nws, NULL));
            float *g = to_host(glayers[1].wq, D * D);
            if (rep == 0) first = g;
            else if (memcmp(first, g, D * D * 4)) FAIL("dropout step is not reproducible");
        }
        float *hs2 = to_host(scores, B * T);
        int differs = 0;
        for (int i = 0; i < B * T; ++i) differs += !hmask[i] && fabsf(hs2[i] - hs[i]) > 1e-3f;
        if (!differs) FAIL("dropout 0.3 changed nothing");
    }
    /* error path */
    vs_dropout_cfg bad = {0.0f, 1.0f, 1};
    if (vs_train_forward(w, x, mask, B, T, &bad, scores, hidden, saved, nsaved, ws, nws, NULL) != VS_ERR_INVALID) FAIL("p = 1 accepted");
    vs_weights_free(w);
    printf("OK loss %.6f, |d final_w|^2 %.4e, %zu bytes of activations kept\n", hl[0], g2, nsaved);
    return 0;
}
