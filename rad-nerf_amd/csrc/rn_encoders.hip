// rn_encoders.hip -- spherical-harmonics and frequency encoders for gfx950.
//
// Behaviour: shencoder/src/shencoder.cu:28-382 and freqencoder/src/freqencoder.cu:30-94 of the
// reference.  Both are streaming kernels (12 B in / 64 B out per sample for the degree-4 SH the
// model uses): one sample per lane, the whole output row built in registers and written with
// 16-byte stores.
#include "rn_sh_dev.h"

namespace rn {

constexpr int kBlockE = 256;

// Jacobian rows d/dx, d/dy, d/dz (shencoder.cu:130-350).  AXIS: 0 = x, 1 = y, 2 = z.
template <uint32_t C, int AXIS>
__device__ __forceinline__ void sh_jac(float x, float y, float z, float *d) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
    (void)xy; (void)xz; (void)yz; (void)x6; (void)y6; (void)z6;
#define SH3(i, ex, ey, ez) d[i] = (AXIS == 0) ? (ex) : (AXIS == 1) ? (ey) : (ez)
    SH3(0, 0.0f, 0.0f, 0.0f);
    if constexpr (C > 1) {
        SH3(1, 0.0f, -0.48860251190291992f, 0.0f);
        SH3(2, 0.0f, 0.0f, 0.48860251190291992f);
        SH3(3, -0.48860251190291992f, 0.0f, 0.0f);
    }
    if constexpr (C > 2) {
        SH3(4, 1.0925484305920792f * y, 1.0925484305920792f * x, 0.0f);
        SH3(5, 0.0f, -1.0925484305920792f * z, -1.0925484305920792f * y);
        SH3(6, 0.0f, 0.0f, 1.8923493915151202f * z);
        SH3(7, -1.0925484305920792f * z, 0.0f, -1.0925484305920792f * x);
        SH3(8, 1.0925484305920792f * x, -1.0925484305920792f * y, 0.0f);
    }
    if constexpr (C > 3) {
        SH3(9, -3.5402615395598609f * xy, -1.7701307697799304f * x2 + 1.7701307697799304f * y2, 0.0f);
        SH3(10, 2.8906114426405538f * yz, 2.8906114426405538f * xz, 2.8906114426405538f * xy);
        SH3(11, 0.0f, 0.45704579946446572f - 2.2852289973223288f * z2, -4.5704579946446566f * yz);
        SH3(12, 0.0f, 0.0f, 5.597644988851731f * z2 - 1.1195289977703462f);
        SH3(13, 0.45704579946446572f - 2.2852289973223288f * z2, 0.0f, -4.5704579946446566f * xz);
        SH3(14, 2.8906114426405538f * xz, -2.8906114426405538f * yz, 1.4453057213202769f * x2 - 1.4453057213202769f * y2);
        SH3(15, -1.7701307697799304f * x2 + 1.7701307697799304f * y2, 3.5402615395598609f * xy, 0.0f);
    }
    if constexpr (C > 4) {
        SH3(16, 2.5033429417967046f * y * (3.0f * x2 - y2), 2.5033429417967046f * x * (x2 - 3.0f * y2), 0.0f);
        SH3(17, -10.620784618679583f * xy * z, 5.3103923093397913f * z * (-x2 + y2), 1.7701307697799304f * y * (-3.0f * x2 + y2));
        SH3(18, 0.94617469575756008f * y * (7.0f * z2 - 1.0f), 0.94617469575756008f * x * (7.0f * z2 - 1.0f), 13.246445740605839f * xy * z);
        SH3(19, 0.0f, 0.66904654355728921f * z * (3.0f - 7.0f * z2), 2.0071396306718676f * y * (1.0f - 7.0f * z2));
        SH3(20, 0.0f, 0.0f, 14.809976568128603f * (z2 * z) - 6.3471328149122579f * z);
        SH3(21, 0.66904654355728921f * z * (3.0f - 7.0f * z2), 0.0f, 2.0071396306718676f * x * (1.0f - 7.0f * z2));
        SH3(22, 0.94617469575756008f * x * (7.0f * z2 - 1.0f), 0.94617469575756008f * y * (1.0f - 7.0f * z2), 6.6232228703029197f * z * (x2 - y2));
        SH3(23, 5.3103923093397913f * z * (-x2 + y2), 10.620784618679583f * xy * z, 1.7701307697799304f * x * (-x2 + 3.0f * y2));
        SH3(24, 2.5033429417967046f * x * (x2 - 3.0f * y2), 2.5033429417967046f * y * (-3.0f * x2 + y2), 0.0f);
    }
    if constexpr (C > 5) {
        SH3(25, 13.127641136803401f * xy * (-x2 + y2), 19.6914617052051f * x2 * y2 - 3.2819102842008503f * x4 - 3.2819102842008503f * y4, 0.0f);
        SH3(26, 8.3026492595241645f * yz * (3.0f * x2 - y2), 8.3026492595241645f * xz * (x2 - 3.0f * y2), 8.3026492595241645f * xy * (x2 - y2));
        SH3(27, 2.9354297966115022f * xy * (1.0f - 9.0f * z2), -1.4677148983057511f * (x2 - y2) * (9.0f * z2 - 1.0f), 8.8062893898345074f * yz * (-3.0f * x2 + y2));
        SH3(28, 4.7935367849733241f * yz * (3.0f * z2 - 1.0f), 4.7935367849733241f * xz * (3.0f * z2 - 1.0f), 4.7935367849733241f * xy * (9.0f * z2 - 1.0f));
        SH3(29, 0.0f, 6.3412531167397574f * z2 - 9.5118796751096362f * z4 - 0.45294665119569694f, 12.682506233479513f * yz * (1.0f - 3.0f * z2));
        SH3(30, 0.0f, 0.0f, -24.559567715218954f * z2 + 36.839351572828434f * z4 + 1.754254836801354f);
        SH3(31, 6.3412531167397574f * z2 - 9.5118796751096362f * z4 - 0.45294665119569694f, 0.0f, 12.682506233479513f * xz * (1.0f - 3.0f * z2));
        SH3(32, 4.7935367849733241f * xz * (3.0f * z2 - 1.0f), 4.7935367849733241f * yz * (1.0f - 3.0f * z2), 2.3967683924866621f * (x2 - y2) * (9.0f * z2 - 1.0f));
        SH3(33, -13.209434084751759f * x2 * z2 + 1.4677148983057511f * x2 + 13.209434084751759f * y2 * z2 - 1.4677148983057511f * y2, 2.9354297966115022f * xy * (9.0f * z2 - 1.0f), 8.8062893898345074f * xz * (-x2 + 3.0f * y2));
        SH3(34, 8.3026492595241645f * xz * (x2 - 3.0f * y2), 8.3026492595241645f * yz * (-3.0f * x2 + y2), -12.453973889286246f * x2 * y2 + 2.0756623148810411f * x4 + 2.0756623148810411f * y4);
        SH3(35, 19.6914617052051f * x2 * y2 - 3.2819102842008503f * x4 - 3.2819102842008503f * y4, 13.127641136803401f * xy * (x2 - y2), 0.0f);
    }
    if constexpr (C > 6) {
        SH3(36, 4.0991046311514854f * y * (-10.0f * x2 * y2 + 5.0f * x4 + y4), 4.0991046311514854f * x * (-10.0f * x2 * y2 + x4 + 5.0f * y4), 0.0f);
        SH3(37, 47.332383244635047f * xy * z * (-x2 + y2), 11.833095811158762f * z * (6.0f * x2 * y2 - x4 - y4), 2.3666191622317521f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4));
        SH3(38, 2.0182596029148963f * y * (3.0f * x2 - y2) * (11.0f * z2 - 1.0f), 2.0182596029148963f * x * (x2 - 3.0f * y2) * (11.0f * z2 - 1.0f), 44.401711264127719f * xy * z * (x2 - y2));
        SH3(39, 5.5272315570895412f * xy * z * (3.0f - 11.0f * z2), -2.7636157785447706f * z * (x2 - y2) * (11.0f * z2 - 3.0f), -2.7636157785447706f * y * (3.0f * x2 - y2) * (11.0f * z2 - 1.0f));
        SH3(40, 0.92120525951492349f * y * (-18.0f * z2 + 33.0f * z4 + 1.0f), 0.92120525951492349f * x * (-18.0f * z2 + 33.0f * z4 + 1.0f), 11.054463114179082f * xy * z * (11.0f * z2 - 3.0f));
        SH3(41, 0.0f, 0.58262136251873131f * z * (30.0f * z2 - 33.0f * z4 - 5.0f), 2.9131068125936568f * y * (18.0f * z2 - 33.0f * z4 - 1.0f));
        SH3(42, 0.0f, 0.0f, 2.6699064952403937f * z * (-30.0f * z2 + 33.0f * z4 + 5.0f));
        SH3(43, 0.58262136251873131f * z * (30.0f * z2 - 33.0f * z4 - 5.0f), 0.0f, 2.9131068125936568f * x * (18.0f * z2 - 33.0f * z4 - 1.0f));
        SH3(44, 0.92120525951492349f * x * (-18.0f * z2 + 33.0f * z4 + 1.0f), 0.92120525951492349f * y * (18.0f * z2 - 33.0f * z4 - 1.0f), 5.5272315570895412f * z * (x2 - y2) * (11.0f * z2 - 3.0f));
        SH3(45, -2.7636157785447706f * z * (x2 - y2) * (11.0f * z2 - 3.0f), 5.5272315570895412f * xy * z * (11.0f * z2 - 3.0f), -2.7636157785447706f * x * (x2 - 3.0f * y2) * (11.0f * z2 - 1.0f));
        SH3(46, 2.0182596029148963f * x * (x2 - 3.0f * y2) * (11.0f * z2 - 1.0f), -2.0182596029148963f * y * (3.0f * x2 - y2) * (11.0f * z2 - 1.0f), 11.10042781603193f * z * (-6.0f * x2 * y2 + x4 + y4));
        SH3(47, 11.833095811158762f * z * (6.0f * x2 * y2 - x4 - y4), 47.332383244635047f * xy * z * (x2 - y2), 2.3666191622317521f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4));
        SH3(48, 4.0991046311514854f * x * (-10.0f * x2 * y2 + x4 + 5.0f * y4), 4.0991046311514854f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4), 0.0f);
    }
    if constexpr (C > 7) {
        SH3(49, 9.9002782553443485f * xy * (10.0f * x2 * y2 - 3.0f * x4 - 3.0f * y4), -74.252086915082614f * x2 * y4 + 74.252086915082614f * x4 * y2 - 4.9501391276721742f * x6 + 4.9501391276721742f * y6, 0.0f);
        SH3(50, 15.875763970811402f * yz * (-10.0f * x2 * y2 + 5.0f * x4 + y4), 15.875763970811402f * xz * (-10.0f * x2 * y2 + x4 + 5.0f * y4), 5.2919213236038001f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4));
        SH3(51, -10.378311574405206f * xy * (x2 - y2) * (13.0f * z2 - 1.0f), 0.51891557872026028f * (13.0f * z2 - 1.0f) * (10.0f * x2 * y2 - 5.0f * x4 + 4.0f * y2 * (5.0f * x2 - y2) - y4), 13.491805046726766f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4));
        SH3(52, 4.1513246297620823f * yz * (3.0f * x2 - y2) * (13.0f * z2 - 3.0f), 4.1513246297620823f * xz * (x2 - 3.0f * y2) * (13.0f * z2 - 3.0f), 12.453973889286248f * xy * (x2 - y2) * (13.0f * z2 - 1.0f));
        SH3(53, 0.93875360317376422f * xy * (66.0f * z2 - 143.0f * z4 - 3.0f), -0.46937680158688211f * (x2 - y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f), -6.8841930899409371f * yz * (3.0f * x2 - y2) * (13.0f * z2 - 3.0f));
        SH3(54, 0.44253269244498261f * yz * (-110.0f * z2 + 143.0f * z4 + 15.0f), 0.44253269244498261f * xz * (-110.0f * z2 + 143.0f * z4 + 15.0f), 2.2126634622249131f * xy * (-66.0f * z2 + 143.0f * z4 + 3.0f));
        SH3(55, 0.0f, -12.194767023639836f * z2 + 44.714145753346067f * z4 - 38.752259652899923f * z6 + 0.45165803791258652f, 1.6259689364853116f * yz * (110.0f * z2 - 143.0f * z4 - 15.0f));
        SH3(56, 0.0f, 0.0f, 64.528641681844675f * z2 - 236.60501950009714f * z4 + 205.05768356675085f * z6 - 2.3899496919201733f);
        SH3(57, -12.194767023639836f * z2 + 44.714145753346067f * z4 - 38.752259652899923f * z6 + 0.45165803791258652f, 0.0f, 1.6259689364853116f * xz * (110.0f * z2 - 143.0f * z4 - 15.0f));
        SH3(58, 0.44253269244498261f * xz * (-110.0f * z2 + 143.0f * z4 + 15.0f), 0.44253269244498261f * yz * (110.0f * z2 - 143.0f * z4 - 15.0f), 0.07375544874083044f * (x2 - y2) * (143.0f * z2 * (3.0f * z2 - 1.0f) + 132.0f * z2 * (13.0f * z2 - 5.0f) - 187.0f * z2 + 45.0f));
        SH3(59, 30.97886890473422f * x2 * z2 - 67.120882626924143f * x2 * z4 - 1.4081304047606462f * x2 - 30.97886890473422f * y2 * z2 + 67.120882626924143f * y2 * z4 + 1.4081304047606462f * y2, 0.93875360317376422f * xy * (-66.0f * z2 + 143.0f * z4 + 3.0f), -6.8841930899409371f * xz * (x2 - 3.0f * y2) * (13.0f * z2 - 3.0f));
        SH3(60, 4.1513246297620823f * xz * (x2 - 3.0f * y2) * (13.0f * z2 - 3.0f), -4.1513246297620823f * yz * (3.0f * x2 - y2) * (13.0f * z2 - 3.0f), 3.1134934723215619f * (13.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4));
        SH3(61, -0.51891557872026028f * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + 4.0f * x2 * (x2 - 5.0f * y2) + x4 + 5.0f * y4), 10.378311574405206f * xy * (x2 - y2) * (13.0f * z2 - 1.0f), 13.491805046726766f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4));
        SH3(62, 15.875763970811402f * xz * (-10.0f * x2 * y2 + x4 + 5.0f * y4), 15.875763970811402f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4), 39.6894099270285f * x2 * y4 - 39.6894099270285f * x4 * y2 + 2.6459606618019f * x6 - 2.6459606618019f * y6);
        SH3(63, -74.252086915082614f * x2 * y4 + 74.252086915082614f * x4 * y2 - 4.9501391276721742f * x6 + 4.9501391276721742f * y6, 9.9002782553443485f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4), 0.0f);
    }
#undef SH3
}

template <uint32_t N>
__device__ __forceinline__ void store_f32_row(float *dst, const float *v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (uint32_t i = 0; i < N / 4; i++)
            reinterpret_cast<float4 *>(dst)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < N; i++) dst[i] = v[i];
    }
}

// shencoder.cu:28-355
template <uint32_t C>
__global__ void __launch_bounds__(kBlockE)
k_sh_forward(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B, uint32_t D,
             float *__restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * kBlockE + threadIdx.x;
    if (b >= B) return;
    constexpr uint32_t C2 = C * C;
    const float x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
    float o[C2];
    sh_basis<C>(x, y, z, o);
    store_f32_row<C2>(outputs + (size_t)b * C2, o);
    if (dy_dx) {
        float *g = dy_dx + (size_t)b * D * C2;  // rows dx | dy | dz  (shencoder.cu:126-128)
        sh_jac<C, 0>(x, y, z, o); store_f32_row<C2>(g, o);
        sh_jac<C, 1>(x, y, z, o); store_f32_row<C2>(g + C2, o);
        sh_jac<C, 2>(x, y, z, o); store_f32_row<C2>(g + 2 * C2, o);
    }
}

// shencoder.cu:359-382: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]
__global__ void __launch_bounds__(kBlockE)
k_sh_backward(const float *__restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
              const float *__restrict__ dy_dx, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kBlockE + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D;
    const float *g = grad + (size_t)b * C2;
    const float *dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) acc += g[ch] * dd[ch];
    grad_inputs[t] = acc;
}

// freqencoder.cu:30-58.  One SAMPLE per lane (the reference uses one output element per thread):
// the lane builds its C = D + 2*D*deg outputs and stores them contiguously.
__global__ void __launch_bounds__(kBlockE)
k_freq_forward(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
               float *__restrict__ outputs) {
    const uint32_t b = blockIdx.x * kBlockE + threadIdx.x;
    if (b >= B) return;
    const float *in = inputs + (size_t)b * D;
    float *out = outputs + (size_t)b * C;
    constexpr float kHalfPi = 3.141592653589793f / 2;
    for (uint32_t d = 0; d < D; d++) {
        const float x = in[d];
        out[d] = x;
        for (uint32_t f = 0; f < deg; f++) {
            const float a = scalbnf(x, (int)f);
            out[D + (2 * f) * D + d] = sinf(a);                // col = 2f,   phase 0
            out[D + (2 * f + 1) * D + d] = sinf(a + kHalfPi);  // col = 2f+1, phase pi/2
        }
    }
}

// freqencoder.cu:63-94
__global__ void __launch_bounds__(kBlockE)
k_freq_backward(const float *__restrict__ grad, const float *__restrict__ outputs, uint32_t B, uint32_t D,
                uint32_t deg, uint32_t C, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kBlockE + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
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

}  // namespace rn

using namespace rn;

extern "C" {

int rn_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C, float *dy_dx,
                         rn_stream_t stream) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    RN_REQUIRE(D == 3, "SH encoder only support input dim == 3");  // sphere_harmonics.py:69
    RN_REQUIRE(C >= 1 && C <= 8, "SH encoder only supports degree in [1, 8]");
    const dim3 grid(div_up(B, kBlockE)), block(kBlockE);
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 1: hipLaunchKernelGGL(k_sh_forward<1>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 2: hipLaunchKernelGGL(k_sh_forward<2>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 3: hipLaunchKernelGGL(k_sh_forward<3>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 4: hipLaunchKernelGGL(k_sh_forward<4>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 5: hipLaunchKernelGGL(k_sh_forward<5>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 6: hipLaunchKernelGGL(k_sh_forward<6>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 7: hipLaunchKernelGGL(k_sh_forward<7>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
        case 8: hipLaunchKernelGGL(k_sh_forward<8>, grid, block, 0, s, inputs, outputs, B, D, dy_dx); break;
    }
    return check_launch("sh_encode_forward");
}

int rn_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                          const float *dy_dx, float *grad_inputs, rn_stream_t stream) {
    if (B == 0) return RN_OK;
    (void)inputs;
    RN_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    RN_REQUIRE(D == 3 && C >= 1 && C <= 8, "sh_encode_backward: D must be 3 and degree in [1, 8]");
    hipLaunchKernelGGL(k_sh_backward, dim3(div_up(B * D, kBlockE)), dim3(kBlockE), 0, as_stream(stream), grad, B, D,
                       C * C, dy_dx, grad_inputs);
    return check_launch("sh_encode_backward");
}

int rn_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs,
                           rn_stream_t stream) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(inputs && outputs, "freq_encode_forward: null pointer");
    RN_REQUIRE(D >= 1 && C == D + 2 * D * deg, "freq_encode_forward: output_dim must equal D + 2*D*deg");
    hipLaunchKernelGGL(k_freq_forward, dim3(div_up(B, kBlockE)), dim3(kBlockE), 0, as_stream(stream), inputs, B, D,
                       deg, C, outputs);
    return check_launch("freq_encode_forward");
}

int rn_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float *grad_inputs, rn_stream_t stream) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(grad && outputs && grad_inputs, "freq_encode_backward: null pointer");
    RN_REQUIRE(D >= 1 && C == D + 2 * D * deg, "freq_encode_backward: output_dim must equal D + 2*D*deg");
    hipLaunchKernelGGL(k_freq_backward, dim3(div_up(B * D, kBlockE)), dim3(kBlockE), 0, as_stream(stream), grad,
                       outputs, B, D, deg, C, grad_inputs);
    return check_launch("freq_encode_backward");
}

}  // extern "C"
