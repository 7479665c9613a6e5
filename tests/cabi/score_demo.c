/* score_demo.c — the scorer used from plain C through include/vs_scorer.h: no Python, no torch.
 * Build: gcc -std=gnu99 score_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L<pkg> -lvsscore
 *        -L/opt/rocm/lib -lamdhip64 -lm   (tests/test_cabi_c.py does this and runs it on the GPU box)
 * It packs a random model (M-A shape, 2 layers), scores a padded batch twice (same bits expected), scores the same
 * videos as a PACKED batch (same bits on the valid frames expected), and checks one error path.  Prints "OK". */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vs_scorer.h"

static uint32_t rng_state = 12345u;
static float rnd(void) {                               /* uniform in (-1, 1) */
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) / 8388608.0f - 1.0f;
}
static float *dev_random(size_t n, float scale, float offset) {
    float *h = (float *)malloc(n * sizeof(float)), *d = NULL;
    for (size_t i = 0; i < n; ++i) h[i] = rnd() * scale + offset;
    if (hipMalloc((void **)&d, n * sizeof(float)) != hipSuccess) exit(2);
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) exit(2);
    free(h);
    return d;
}
#define CHECK(call) do { int rc_ = (call); if (rc_ != VS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vs_last_error()); return 1; } } while (0)

int main(void) {
    enum { D = 256, H = 4, L = 2, DIN = 1024, MAXLEN = 2000, B = 3, T = 200 };
    const int lengths[B] = {200, 137, 64};
    if (vs_abi_version() != VS_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    vs_layer_params layers[L];
    for (int l = 0; l < L; ++l) {
        const float s = 1.0f / 16.0f;
        layers[l].wq = dev_random(D * D, s, 0); layers[l].bq = dev_random(D, 0.1f, 0);
        layers[l].wk = dev_random(D * D, s, 0); layers[l].bk = dev_random(D, 0.1f, 0);
        layers[l].wv = dev_random(D * D, s, 0); layers[l].bv = dev_random(D, 0.1f, 0);
        layers[l].wo = dev_random(D * D, s, 0); layers[l].bo = dev_random(D, 0.1f, 0);
        layers[l].ln1_g = dev_random(D, 0.1f, 1.0f); layers[l].ln1_b = dev_random(D, 0.1f, 0);
        layers[l].w1 = dev_random(4 * D * D, s, 0); layers[l].b1 = dev_random(4 * D, 0.1f, 0);
        layers[l].w2 = dev_random(4 * D * D, s / 2, 0); layers[l].b2 = dev_random(D, 0.1f, 0);
        layers[l].ln2_g = dev_random(D, 0.1f, 1.0f); layers[l].ln2_b = dev_random(D, 0.1f, 0);
    }
    vs_model_params P;
    P.embed_w = dev_random((size_t)D * DIN, 1.0f / 32.0f, 0); P.embed_b = dev_random(D, 0.1f, 0);
    P.pos_embedding = dev_random((size_t)MAXLEN * D, 1.0f, 0);
    P.layers = layers;
    P.final_w = dev_random(D, 1.0f / 16.0f, 0); P.final_b = dev_random(1, 0.1f, 0);
    vs_model_desc desc = {D, H, L, DIN, MAXLEN, 1};
    vs_weights *w = NULL;
    CHECK(vs_weights_pack(&desc, &P, NULL, &w));

    /* padded batch: 1000.0 in every feature of the padded frames + key mask, like collate_fn_train */
    float *hx = (float *)malloc((size_t)B * T * DIN * sizeof(float));
    uint8_t hmask[B * T];
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            hmask[b * T + t] = t >= lengths[b];
            for (int c = 0; c < DIN; ++c) hx[((size_t)b * T + t) * DIN + c] = t < lengths[b] ? fabsf(rnd()) * 0.5f : 1000.0f;
        }
    float *x = NULL, *scores = NULL, *scores2 = NULL; uint8_t *mask = NULL; void *ws = NULL;
    hipMalloc((void **)&x, (size_t)B * T * DIN * 4); hipMemcpy(x, hx, (size_t)B * T * DIN * 4, hipMemcpyHostToDevice);
    hipMalloc((void **)&mask, B * T); hipMemcpy(mask, hmask, B * T, hipMemcpyHostToDevice);
    hipMalloc((void **)&scores, B * T * 4); hipMalloc((void **)&scores2, B * T * 4);
    const size_t need = vs_scorer_workspace_bytes(w, B, T);
    hipMalloc(&ws, need);
    CHECK(vs_scorer_forward(w, x, mask, B, T, VS_FLAG_SIGMOID, scores, NULL, ws, need, NULL));
    CHECK(vs_scorer_forward(w, x, mask, B, T, VS_FLAG_SIGMOID, scores2, NULL, ws, need, NULL));
    float h1[B * T], h2[B * T];
    hipDeviceSynchronize();
    hipMemcpy(h1, scores, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(h2, scores2, sizeof h2, hipMemcpyDeviceToHost);
    for (int i = 0; i < B * T; ++i) {
        if (!hmask[i] && !(h1[i] > 0.0f && h1[i] < 1.0f)) { fprintf(stderr, "score %d = %g\n", i, h1[i]); return 1; }
        if (memcmp(&h1[i], &h2[i], 4)) { fprintf(stderr, "run-to-run difference at %d\n", i); return 1; }
    }

    /* the same videos packed: frames concatenated, no padding, no mask */
    int mtot = 0; for (int b = 0; b < B; ++b) mtot += lengths[b];
    float *hxp = (float *)malloc((size_t)mtot * DIN * sizeof(float)), *xp = NULL, *sp = NULL; int32_t *dl = NULL; void *wsp = NULL;
    for (int b = 0, r = 0; b < B; ++b)
        for (int t = 0; t < lengths[b]; ++t, ++r) memcpy(hxp + (size_t)r * DIN, hx + ((size_t)b * T + t) * DIN, DIN * 4);
    hipMalloc((void **)&xp, (size_t)mtot * DIN * 4); hipMemcpy(xp, hxp, (size_t)mtot * DIN * 4, hipMemcpyHostToDevice);
    hipMalloc((void **)&sp, mtot * 4); hipMalloc((void **)&dl, sizeof lengths); hipMemcpy(dl, lengths, sizeof lengths, hipMemcpyHostToDevice);
    const size_t needp = vs_scorer_workspace_bytes_packed(w, lengths, B);
    hipMalloc(&wsp, needp);
    CHECK(vs_scorer_forward_packed(w, xp, lengths, dl, B, VS_FLAG_SIGMOID, sp, NULL, wsp, needp, NULL));
    float *hp = (float *)malloc(mtot * 4);
    hipDeviceSynchronize();
    hipMemcpy(hp, sp, mtot * 4, hipMemcpyDeviceToHost);
    for (int b = 0, r = 0; b < B; ++b)
        for (int t = 0; t < lengths[b]; ++t, ++r)
            if (memcmp(&hp[r], &h1[b * T + t], 4)) { fprintf(stderr, "packed != padded at video %d frame %d\n", b, t); return 1; }

    /* error path: T beyond the positional table */
    if (vs_scorer_forward(w, x, NULL, 1, MAXLEN + 1, 0, scores, NULL, ws, need, NULL) != VS_ERR_INVALID || !strstr(vs_last_error(), "positional")) {
        fprintf(stderr, "expected VS_ERR_INVALID for T > max_len\n"); return 1;
    }
    vs_weights_free(w);
    printf("OK %d padded frames, %d packed frames, first scores %.6f %.6f\n", B * T, mtot, h1[0], h1[1]);
    return 0;
}
