/*
 * orc_freq.c -- CPU oracle (TEST INFRASTRUCTURE, see radnerf_oracle.h) for
 * freqencoder/src/freqencoder.cu:30-94 (NeRF positional encoding).
 * The reference is built with -use_fast_math (__sinf); the oracle uses libm
 * sinf, so GPU parity is a tolerance, not bit equality.
 */
#include "radnerf_oracle.h"

#include <math.h>

#define ORC_PI 3.141592653589793f /* freqencoder.cu:21 */

/* freqencoder.cu:30-58: one output element per work item */
void orc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *outputs) {
    (void)deg;
#pragma omp parallel for schedule(static)
    for (int64_t tt = 0; tt < (int64_t)B * C; tt++) {
        const uint32_t t = (uint32_t)tt;
        const uint32_t b = t / C;
        const uint32_t c = t - b * C;
        const float *in = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = in[c];
        } else {
            const uint32_t col = c / D - 1;
            const uint32_t d = c % D;
            const uint32_t freq = col / 2;
            const float phase_shift = (float)(col % 2) * (ORC_PI / 2);
            outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
    }
}

/* freqencoder.cu:63-94 */
void orc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                              uint32_t deg, uint32_t C, float *grad_inputs) {
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D;
        const uint32_t d = t - b * D;
        const float *g = grad + (size_t)b * C;
        const float *o = outputs + (size_t)b * C;
        float result = g[d];
        g += D;
        o += D;
        for (uint32_t f = 0; f < deg; f++) {
            result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D;
            o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}
