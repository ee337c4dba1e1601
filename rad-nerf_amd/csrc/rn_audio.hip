// rn_audio.hip -- the per-frame audio code as one kernel (C ABI: include/radnerf_fused.h, "audio code").
//
// What is computed: NeRFNetwork.encode_audio (nerf/network.py:170-185) = AudioNet (:41-67) on the 8 frames of the
// attention window + AudioAttNet (:10-37); then the lip-smoothing EMA of nerf/renderer.py:190-194 (rn_audio_smooth).
// How: AudioNet is independent per frame, so k_audio_frames runs one 256-thread workgroup per (window, frame) -- 8 per
// window -- and k_audio_attend one per window.  Activations live in LDS ([channel][position]); each layer's weights are
// staged into LDS with coalesced loads, then every thread produces outputs in a strided loop.  ~0.6 M MAC per window:
// the point is latency (2 launches instead of ~45) -- which is also what lets a frame-parallel rank advance the
// smoothing state through the frames other ranks render without paying 45 launches for each of them.
#include "rn_common.h"

#include "../../include/radnerf_fused.h"

namespace rn {

constexpr int kAudioThreads = 256;
constexpr int kSeq = RN_AUDIO_SEQ, kWin = RN_AUDIO_WIN;
constexpr int kMaxDimIn = 64;     // dim_in is 29 / 32 / 44 (nerf/network.py:112-117)
constexpr int kMaxWeights = 64 * 64 * 3;  // largest layer: conv 64 -> 64, k = 3

struct AudioW {
    const float *conv_w[4], *conv_b[4], *fc_w[2], *fc_b[2], *att_conv_w[5], *att_conv_b[5], *att_fc_w, *att_fc_b;
    uint32_t dim_in, dim_aud, has_att;
};

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.02f * v; }  // nn.LeakyReLU(0.02)

// How many lanes share one output (a power of two <= 16): layers with few outputs would otherwise leave most of the
// workgroup idle behind a few long dot products.  The partial sums meet in a __shfl_xor tree.
__device__ __forceinline__ int split_for(int total) {
    int split = 1;
    while (split < 16 && total * split * 2 <= kAudioThreads) split *= 2;
    return split;
}
__device__ __forceinline__ float quad_sum(float v, int split) {
    for (int d = 1; d < split; d <<= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Conv1d(kernel 3, padding 1) + LeakyReLU over `frames` independent [cin, len_in] maps held in LDS.
// in: [frames][cin][len_in], out: [frames][cout][len_out], len_out = (len_in - 1) / stride + 1; w: [cout][cin][3] in LDS.
__device__ __forceinline__ void conv3(const float *in, float *out, const float *w, const float *__restrict__ bias, int frames,
                                      int cin, int cout, int len_in, int stride) {
    const int len_out = (len_in - 1) / stride + 1;
    const int total = frames * cout * len_out;
    const int split = split_for(total);
    for (int base = 0; base < total * split; base += kAudioThreads) {   // uniform trip count: the shuffles need whole waves
        const int id = base + (int)threadIdx.x, o = id / split, part = id % split;
        float acc = 0.0f;
        if (o < total) {
            const int pos = o % len_out, co = (o / len_out) % cout, f = o / (len_out * cout);
            const float *x = in + (size_t)f * cin * len_in;
            const float *wr = w + (size_t)co * cin * 3;
            const int c0 = pos * stride - 1;
            for (int ci = part; ci < cin; ci += split) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int p = c0 + k;
                    if (p >= 0 && p < len_in) acc += wr[ci * 3 + k] * x[ci * len_in + p];
                }
            }
        }
        acc = quad_sum(acc, split);
        if (o < total && part == 0) out[o] = leaky(acc + bias[(o / len_out) % cout]);
    }
}

// rows: y[f][o] = act(b[o] + sum_k W[o][k] x[f][k]); w [dout][din] in LDS
__device__ __forceinline__ void linear(const float *in, float *out, const float *w, const float *__restrict__ bias, int frames,
                                       int din, int dout, bool act) {
    const int total = frames * dout;
    const int split = split_for(total);
    for (int base = 0; base < total * split; base += kAudioThreads) {
        const int id = base + (int)threadIdx.x, o = id / split, part = id % split;
        float acc = 0.0f;
        if (o < total) {
            const int f = o / dout, r = o % dout;
            for (int k = part; k < din; k += split) acc += w[r * din + k] * in[f * din + k];
        }
        acc = quad_sum(acc, split);
        if (o < total && part == 0) {
            acc += bias[o % dout];
            out[o] = act ? leaky(acc) : acc;
        }
    }
}

// The next layer's weights travel global -> registers while the current layer computes, then registers -> the other LDS
// buffer: the workgroup waits for one memory round trip per kernel instead of one per layer.
constexpr int kPrefetch = kMaxWeights / kAudioThreads;  // 48 floats per lane
struct Prefetch {
    float v[kPrefetch];
    __device__ __forceinline__ void load(const float *__restrict__ src, int n) {
#pragma unroll
        for (int i = 0; i < kPrefetch; i++) {
            const int at = i * kAudioThreads + (int)threadIdx.x;
            v[i] = at < n ? src[at] : 0.0f;
        }
    }
    __device__ __forceinline__ void store(float *dst, int n) const {  // caller syncs before (dst free) and after (dst ready)
#pragma unroll
        for (int i = 0; i < kPrefetch; i++) {
            const int at = i * kAudioThreads + (int)threadIdx.x;
            if (at < n) dst[at] = v[i];
        }
    }
};

// window source: explicit windows [n][frames][dim_in][16], or cut from a stream [T][dim_in][16] (nerf/utils.py:56-72)
struct Source {
    const float *base;
    uint32_t T, first;
    int from_stream;
};

// AudioNet on ONE frame per workgroup: blockIdx.x = window * frames_per_window + t; codes [n * frames][dim_aud]
// acts (nullable; the training forward): every layer's output of the frame, [n * frames][kActs] = y1 [32][8] | y2 [32][4] |
// y3 [64][2] | y4 [64] | y5 [64], for k_audio_frames_bwd to start from instead of running the forward again
constexpr int kActs = 32 * 8 + 32 * 4 + 64 * 2 + 64 + 64;
__device__ __forceinline__ void keep_acts(float *acts, int at, const float *buf, int n) {   // call after the layer's barrier
    if (!acts) return;
    for (int i = threadIdx.x; i < n; i += kAudioThreads) acts[(size_t)blockIdx.x * kActs + at + i] = buf[i];
}

__global__ void __launch_bounds__(kAudioThreads) k_audio_frames(AudioW w, Source src, float *__restrict__ codes, float *__restrict__ acts) {
    __shared__ float wts0[kMaxWeights], wts1[kMaxWeights];  // weights of the layer computing / of the next one
    __shared__ float bufA[kMaxDimIn * kWin];  // ping
    __shared__ float bufB[32 * 8];            // pong (largest: conv1 output)
    const int frames = w.has_att ? kSeq : 1;
    const int cin0 = (int)w.dim_in, A = (int)w.dim_aud;
    const uint32_t win = blockIdx.x / frames, t = blockIdx.x % frames;
    Prefetch pre;
    pre.load(w.conv_w[0], 32 * cin0 * 3);

    // the frame's input map: x[:, :, 8 - 8 : 8 + 8] of nerf/network.py:62-63 is the whole 16-sample frame
    for (int i = threadIdx.x; i < cin0 * kWin; i += kAudioThreads) {
        float v;
        if (src.from_stream) {
            const int centre = (int)((src.first + win) % src.T);
            const int g = centre - 4 + (int)t;  // frames index-4 .. index+3, zero outside the stream
            v = (g >= 0 && g < (int)src.T) ? src.base[(size_t)g * cin0 * kWin + i] : 0.0f;
        } else {
            v = src.base[(size_t)blockIdx.x * cin0 * kWin + i];
        }
        bufA[i] = v;
    }
    pre.store(wts0, 32 * cin0 * 3);
    __syncthreads();
    // AudioNet.encoder_conv: dim_in -> 32 -> 32 -> 64 -> 64, lengths 16 -> 8 -> 4 -> 2 -> 1; then encoder_fc1:
    // Linear(64, 64) + LeakyReLU, Linear(64, dim_aud).  Layer l computes from wts[l & 1] while layer l + 1's weights load.
    pre.load(w.conv_w[1], 32 * 32 * 3);
    conv3(bufA, bufB, wts0, w.conv_b[0], 1, cin0, 32, 16, 2);
    pre.store(wts1, 32 * 32 * 3);
    __syncthreads();
    keep_acts(acts, 0, bufB, 32 * 8);
    pre.load(w.conv_w[2], 64 * 32 * 3);
    conv3(bufB, bufA, wts1, w.conv_b[1], 1, 32, 32, 8, 2);
    pre.store(wts0, 64 * 32 * 3);
    __syncthreads();
    keep_acts(acts, 256, bufA, 32 * 4);
    pre.load(w.conv_w[3], 64 * 64 * 3);
    conv3(bufA, bufB, wts0, w.conv_b[2], 1, 32, 64, 4, 2);
    pre.store(wts1, 64 * 64 * 3);
    __syncthreads();
    keep_acts(acts, 384, bufB, 64 * 2);
    pre.load(w.fc_w[0], 64 * 64);
    conv3(bufB, bufA, wts1, w.conv_b[3], 1, 64, 64, 2, 2);  // -> [64][1]
    pre.store(wts0, 64 * 64);
    __syncthreads();
    keep_acts(acts, 512, bufA, 64);
    pre.load(w.fc_w[1], A * 64);
    linear(bufA, bufB, wts0, w.fc_b[0], 1, 64, 64, true);
    pre.store(wts1, A * 64);
    __syncthreads();
    keep_acts(acts, 576, bufB, 64);
    linear(bufB, codes + (size_t)blockIdx.x * A, wts1, w.fc_b[1], 1, 64, A, false);
}

// AudioAttNet on one window per workgroup: codes [n][8][A] -> enc [n][A]
__global__ void __launch_bounds__(kAudioThreads) k_audio_attend(AudioW w, const float *__restrict__ codes_all,
                                                                float *__restrict__ enc) {
    __shared__ float wts[64 * 16 * 3 + 16 * 8 * 3 + 8 * 4 * 3 + 4 * 2 * 3 + 2 * 1 * 3];  // all five conv layers
    __shared__ float bufA[64 * kSeq], bufB[16 * kSeq], codes[kSeq * 64], att[kSeq];
    const int A = (int)w.dim_aud;
    const uint32_t win = blockIdx.x;
    const int chans[6] = {A, 16, 8, 4, 2, 1};
    int woff[6];
    woff[0] = 0;
    for (int l = 0; l < 5; l++) woff[l + 1] = woff[l] + chans[l + 1] * chans[l] * 3;
    for (int l = 0; l < 5; l++)   // one round trip for all weights (3.6 k floats) instead of one per layer
        for (int i = threadIdx.x; i < woff[l + 1] - woff[l]; i += kAudioThreads) wts[woff[l] + i] = w.att_conv_w[l][i];
    for (int i = threadIdx.x; i < kSeq * A; i += kAudioThreads) codes[i] = codes_all[(size_t)win * kSeq * A + i];
    __syncthreads();
    // x [8, A] -> permute -> [A channels][8 positions]
    for (int i = threadIdx.x; i < A * kSeq; i += kAudioThreads) bufA[i] = codes[(i % kSeq) * A + i / kSeq];
    __syncthreads();
    float *a = bufA, *b = bufB;
    for (int l = 0; l < 5; l++) {
        conv3(a, b, wts + woff[l], w.att_conv_b[l], 1, chans[l], chans[l + 1], kSeq, 1);
        __syncthreads();
        float *t = a; a = b; b = t;
    }
    __syncthreads();  // a: [1][8] scores
    // attentionNet: Linear(8, 8) + Softmax(dim=1), then sum_t y[t] * x[t, :]
    if (threadIdx.x < kSeq) {
        float acc = w.att_fc_b[threadIdx.x];
        for (int k = 0; k < kSeq; k++) acc += w.att_fc_w[threadIdx.x * kSeq + k] * a[k];
        att[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = att[0];
        for (int t = 1; t < kSeq; t++) m = fmaxf(m, att[t]);
        float e[kSeq], s = 0.0f;
        for (int t = 0; t < kSeq; t++) { e[t] = expf(att[t] - m); s += e[t]; }
        for (int t = 0; t < kSeq; t++) att[t] = e[t] / s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A; i += kAudioThreads) {
        float acc = 0.0f;
        for (int t = 0; t < kSeq; t++) acc += att[t] * codes[t * A + i];
        enc[(size_t)win * A + i] = acc;
    }
}

__global__ void k_audio_smooth(const float *__restrict__ enc, uint32_t n, uint32_t dim, float lambda, float *__restrict__ state,
                               int state_valid, float *__restrict__ seq) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim) return;
    float s = state[i];
    bool valid = state_valid != 0;
    const float keep = lambda, take = (float)(1.0 - (double)lambda);  // Python: _lambda * a + (1 - _lambda) * b
    for (uint32_t f = 0; f < n; f++) {
        const float e = enc[(size_t)f * dim + i];
        s = valid ? keep * s + take * e : e;
        valid = true;
        if (seq) seq[(size_t)f * dim + i] = s;     // rn_audio_smooth_seq: every intermediate state
    }
    state[i] = s;
}

// ==========================================================================================================
// Backward of encode_audio for the training step (nerf/utils.py:718-806 trains audio_net / audio_att_net with lr_net):
// the same two-kernel shape as the forward.  Each workgroup recomputes its forward with every layer's output kept in LDS,
// then walks the layers backwards; weight and bias gradients are ADDED to the caller's buffers with atomics (8 frames
// share AudioNet's weights).  ~50 torch / MIOpen launches (naive conv kernels of 7 - 20 us each) become two.
struct AudioG {
    float *conv_w[4], *conv_b[4], *fc_w[2], *fc_b[2], *att_conv_w[5], *att_conv_b[5], *att_fc_w, *att_fc_b;
};

__device__ __forceinline__ float leaky_grad(float y) { return y > 0.0f ? 1.0f : 0.02f; }  // y = leaky(z) has the sign of z

__device__ __forceinline__ void stage_weights(float *dst, const float *__restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += kAudioThreads) dst[i] = src[i];
}

// Backward of conv3 (+ LeakyReLU): x [cin][len_in] -> y [cout][len_out].  dy is turned into dz in place; gw / gb receive
// atomicAdds; dx (nullable) [cin][len_in] is written.  Ends with a barrier.
__device__ __forceinline__ void conv3_bwd(const float *x, const float *y, float *dy, const float *w, float *__restrict__ gw,
                                          float *__restrict__ gb, float *dx, int cin, int cout, int len_in, int stride) {
    const int len_out = (len_in - 1) / stride + 1;
    for (int o = threadIdx.x; o < cout * len_out; o += kAudioThreads) dy[o] *= leaky_grad(y[o]);
    __syncthreads();
    for (int e = threadIdx.x; e < cout * cin * 3; e += kAudioThreads) {
        const int k = e % 3, ci = (e / 3) % cin, co = e / (3 * cin);
        float acc = 0.0f;
        for (int pos = 0; pos < len_out; pos++) {
            const int p = pos * stride - 1 + k;
            if (p >= 0 && p < len_in) acc += dy[co * len_out + pos] * x[ci * len_in + p];
        }
        atomicAdd(gw + e, acc);
    }
    for (int co = threadIdx.x; co < cout; co += kAudioThreads) {
        float acc = 0.0f;
        for (int pos = 0; pos < len_out; pos++) acc += dy[co * len_out + pos];
        atomicAdd(gb + co, acc);
    }
    if (dx) {
        for (int e = threadIdx.x; e < cin * len_in; e += kAudioThreads) {
            const int p = e % len_in, ci = e / len_in;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int q = p + 1 - k;  // = pos * stride
                if (q < 0 || q % stride) continue;
                const int pos = q / stride;
                if (pos >= len_out) continue;
                for (int co = 0; co < cout; co++) acc += w[(co * cin + ci) * 3 + k] * dy[co * len_out + pos];
            }
            dx[e] = acc;
        }
    }
    __syncthreads();
}

// Backward of linear(): y [dout] = act(W x + b).  dy -> dz in place (act), gw/gb atomics, dx written.  Ends with a barrier.
__device__ __forceinline__ void linear_bwd(const float *x, const float *y, float *dy, const float *w, float *__restrict__ gw,
                                           float *__restrict__ gb, float *dx, int din, int dout, bool act) {
    if (act) {
        for (int o = threadIdx.x; o < dout; o += kAudioThreads) dy[o] *= leaky_grad(y[o]);
        __syncthreads();
    }
    for (int e = threadIdx.x; e < dout * din; e += kAudioThreads) atomicAdd(gw + e, dy[e / din] * x[e % din]);
    for (int o = threadIdx.x; o < dout; o += kAudioThreads) atomicAdd(gb + o, dy[o]);
    if (dx) {
        for (int k = threadIdx.x; k < din; k += kAudioThreads) {
            float acc = 0.0f;
            for (int r = 0; r < dout; r++) acc += w[r * din + k] * dy[r];
            dx[k] = acc;
        }
    }
    __syncthreads();
}

// AudioAttNet backward, one window per workgroup: grad_enc [n][A] + codes [n][8][A] -> grad_codes [n][8][A]
__global__ void __launch_bounds__(kAudioThreads) k_audio_attend_bwd(AudioW w, AudioG g, const float *__restrict__ codes_all,
                                                                    const float *__restrict__ grad_enc, float *__restrict__ grad_codes) {
    __shared__ float wts[64 * 16 * 3 + 16 * 8 * 3 + 8 * 4 * 3 + 4 * 2 * 3 + 2 * 1 * 3];
    __shared__ float act[(64 + 16 + 8 + 4 + 2 + 1) * kSeq];   // a0 (= codes^T) .. a5 (scores)
    __shared__ float grad[2][64 * kSeq];
    __shared__ float codes[kSeq * 64], att[kSeq], logit_g[kSeq], genc[64];
    const int A = (int)w.dim_aud;
    const uint32_t win = blockIdx.x;
    const int chans[6] = {A, 16, 8, 4, 2, 1};
    int woff[6], aoff[7];
    woff[0] = 0; aoff[0] = 0;
    for (int l = 0; l < 5; l++) woff[l + 1] = woff[l] + chans[l + 1] * chans[l] * 3;
    for (int l = 0; l < 6; l++) aoff[l + 1] = aoff[l] + chans[l] * kSeq;
    for (int l = 0; l < 5; l++) stage_weights(wts + woff[l], w.att_conv_w[l], woff[l + 1] - woff[l]);
    for (int i = threadIdx.x; i < kSeq * A; i += kAudioThreads) codes[i] = codes_all[(size_t)win * kSeq * A + i];
    for (int i = threadIdx.x; i < A; i += kAudioThreads) genc[i] = grad_enc[(size_t)win * A + i];
    __syncthreads();
    for (int i = threadIdx.x; i < A * kSeq; i += kAudioThreads) act[i] = codes[(i % kSeq) * A + i / kSeq];
    __syncthreads();
    for (int l = 0; l < 5; l++) {   // forward, every layer kept
        conv3(act + aoff[l], act + aoff[l + 1], wts + woff[l], w.att_conv_b[l], 1, chans[l], chans[l + 1], kSeq, 1);
        __syncthreads();
    }
    const float *score = act + aoff[5];
    if (threadIdx.x < kSeq) {
        float acc = w.att_fc_b[threadIdx.x];
        for (int k = 0; k < kSeq; k++) acc += w.att_fc_w[threadIdx.x * kSeq + k] * score[k];
        att[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = att[0];
        for (int t = 1; t < kSeq; t++) m = fmaxf(m, att[t]);
        float e[kSeq], s = 0.0f;
        for (int t = 0; t < kSeq; t++) { e[t] = expf(att[t] - m); s += e[t]; }
        for (int t = 0; t < kSeq; t++) att[t] = e[t] / s;
    }
    __syncthreads();
    // enc = sum_t att[t] codes[t]: d att[t] = <genc, codes[t]>; softmax backward -> d logits
    if (threadIdx.x < kSeq) {
        float acc = 0.0f;
        for (int i = 0; i < A; i++) acc += genc[i] * codes[threadIdx.x * A + i];
        logit_g[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float dot = 0.0f;
        for (int t = 0; t < kSeq; t++) dot += att[t] * logit_g[t];
        for (int t = 0; t < kSeq; t++) logit_g[t] = att[t] * (logit_g[t] - dot);
    }
    __syncthreads();
    // attentionNet Linear(8, 8): logits = W score + b
    float *ga = grad[0], *gb_ = grad[1];
    if (threadIdx.x < kSeq * kSeq) atomicAdd(g.att_fc_w + threadIdx.x, logit_g[threadIdx.x / kSeq] * score[threadIdx.x % kSeq]);
    if (threadIdx.x < kSeq) {
        atomicAdd(g.att_fc_b + threadIdx.x, logit_g[threadIdx.x]);
        float acc = 0.0f;
        for (int t = 0; t < kSeq; t++) acc += w.att_fc_w[t * kSeq + threadIdx.x] * logit_g[t];
        ga[threadIdx.x] = acc;    // d score
    }
    __syncthreads();
    for (int l = 4; l >= 0; l--) {
        conv3_bwd(act + aoff[l], act + aoff[l + 1], ga, wts + woff[l], g.att_conv_w[l], g.att_conv_b[l], gb_, chans[l], chans[l + 1],
                  kSeq, 1);
        float *t = ga; ga = gb_; gb_ = t;
    }
    // ga: d a0 [A][8]; codes enter twice: through a0 (permuted) and through the weighted sum
    for (int i = threadIdx.x; i < kSeq * A; i += kAudioThreads) {
        const int t = i / A, c = i % A;
        grad_codes[(size_t)win * kSeq * A + i] = ga[c * kSeq + t] + att[t] * genc[c];
    }
}

// AudioNet backward, one frame per workgroup: grad_codes [n * frames][A] -> weight gradients
__global__ void __launch_bounds__(kAudioThreads) k_audio_frames_bwd(AudioW w, AudioG g, Source src, const float *__restrict__ grad_codes,
                                                                    const float *__restrict__ acts) {
    __shared__ float wts[kMaxWeights];
    __shared__ float x0[kMaxDimIn * kWin], y1[32 * 8], y2[32 * 4], y3[64 * 2], y4[64], y5[64];
    __shared__ float ga[32 * 8], gb_[32 * 8];
    const int frames = w.has_att ? kSeq : 1;
    const int cin0 = (int)w.dim_in, A = (int)w.dim_aud;
    const uint32_t win = blockIdx.x / frames, t = blockIdx.x % frames;
    for (int i = threadIdx.x; i < cin0 * kWin; i += kAudioThreads) {
        float v;
        if (src.from_stream) {
            const int centre = (int)((src.first + win) % src.T);
            const int gidx = centre - 4 + (int)t;
            v = (gidx >= 0 && gidx < (int)src.T) ? src.base[(size_t)gidx * cin0 * kWin + i] : 0.0f;
        } else {
            v = src.base[(size_t)blockIdx.x * cin0 * kWin + i];
        }
        x0[i] = v;
    }
    if (acts) {   // the forward kept every layer's output (k_audio_frames): five weight stagings and layer passes less
        const float *a = acts + (size_t)blockIdx.x * kActs;
        for (int i = threadIdx.x; i < kActs; i += kAudioThreads) {
            const float v = a[i];
            if (i < 256) y1[i] = v;
            else if (i < 384) y2[i - 256] = v;
            else if (i < 512) y3[i - 384] = v;
            else if (i < 576) y4[i - 512] = v;
            else y5[i - 576] = v;
        }
        __syncthreads();
    } else {
    // forward with every layer kept (same helpers, same order of operations as k_audio_frames)
    stage_weights(wts, w.conv_w[0], 32 * cin0 * 3); __syncthreads();
    conv3(x0, y1, wts, w.conv_b[0], 1, cin0, 32, 16, 2); __syncthreads();
    stage_weights(wts, w.conv_w[1], 32 * 32 * 3); __syncthreads();
    conv3(y1, y2, wts, w.conv_b[1], 1, 32, 32, 8, 2); __syncthreads();
    stage_weights(wts, w.conv_w[2], 64 * 32 * 3); __syncthreads();
    conv3(y2, y3, wts, w.conv_b[2], 1, 32, 64, 4, 2); __syncthreads();
    stage_weights(wts, w.conv_w[3], 64 * 64 * 3); __syncthreads();
    conv3(y3, y4, wts, w.conv_b[3], 1, 64, 64, 2, 2); __syncthreads();
    stage_weights(wts, w.fc_w[0], 64 * 64); __syncthreads();
    linear(y4, y5, wts, w.fc_b[0], 1, 64, 64, true); __syncthreads();
    }
    // backward
    for (int i = threadIdx.x; i < A; i += kAudioThreads) ga[i] = grad_codes[(size_t)blockIdx.x * A + i];
    stage_weights(wts, w.fc_w[1], A * 64); __syncthreads();
    linear_bwd(y5, nullptr, ga, wts, g.fc_w[1], g.fc_b[1], gb_, 64, A, false);
    stage_weights(wts, w.fc_w[0], 64 * 64); __syncthreads();
    linear_bwd(y4, y5, gb_, wts, g.fc_w[0], g.fc_b[0], ga, 64, 64, true);
    stage_weights(wts, w.conv_w[3], 64 * 64 * 3); __syncthreads();
    conv3_bwd(y3, y4, ga, wts, g.conv_w[3], g.conv_b[3], gb_, 64, 64, 2, 2);
    stage_weights(wts, w.conv_w[2], 64 * 32 * 3); __syncthreads();
    conv3_bwd(y2, y3, gb_, wts, g.conv_w[2], g.conv_b[2], ga, 32, 64, 4, 2);
    stage_weights(wts, w.conv_w[1], 32 * 32 * 3); __syncthreads();
    conv3_bwd(y1, y2, ga, wts, g.conv_w[1], g.conv_b[1], gb_, 32, 32, 8, 2);
    conv3_bwd(x0, y1, gb_, wts, g.conv_w[0], g.conv_b[0], nullptr, cin0, 32, 16, 2);   // no dx: weights not needed
}

static int check_audio(const rn_audio_weights_t *w) {
    RN_REQUIRE(w, "audio: null weights");
    for (int l = 0; l < 4; l++) RN_REQUIRE(w->conv_w[l] && w->conv_b[l], "audio: null AudioNet conv weights");
    RN_REQUIRE(w->fc_w[0] && w->fc_b[0] && w->fc_w[1] && w->fc_b[1], "audio: null AudioNet fc weights");
    RN_REQUIRE(w->dim_in >= 1 && w->dim_in <= (uint32_t)kMaxDimIn && w->dim_aud >= 1 && w->dim_aud <= 64,
               "audio: dim_in must be <= 64 and dim_aud <= 64 (got %u, %u)", w->dim_in, w->dim_aud);
    if (w->has_att) {
        for (int l = 0; l < 5; l++) RN_REQUIRE(w->att_conv_w[l] && w->att_conv_b[l], "audio: null AudioAttNet conv weights");
        RN_REQUIRE(w->att_fc_w && w->att_fc_b, "audio: null AudioAttNet fc weights");
    }
    return RN_OK;
}

static AudioW audio_w(const rn_audio_weights_t *w) {
    AudioW a{};
    for (int l = 0; l < 4; l++) { a.conv_w[l] = w->conv_w[l]; a.conv_b[l] = w->conv_b[l]; }
    for (int l = 0; l < 2; l++) { a.fc_w[l] = w->fc_w[l]; a.fc_b[l] = w->fc_b[l]; }
    for (int l = 0; l < 5; l++) { a.att_conv_w[l] = w->att_conv_w[l]; a.att_conv_b[l] = w->att_conv_b[l]; }
    a.att_fc_w = w->att_fc_w; a.att_fc_b = w->att_fc_b;
    a.dim_in = w->dim_in; a.dim_aud = w->dim_aud; a.has_att = w->has_att ? 1u : 0u;
    return a;
}

}  // namespace rn

using namespace rn;

extern "C" {

static int launch_audio(const rn_audio_weights_t *w, const Source &src, uint32_t n, float *enc, float *workspace, hipStream_t s,
                        const char *what, float *acts = nullptr) {
    const AudioW a = audio_w(w);
    if (!a.has_att) {  // one frame per window: the frame code is the result
        hipLaunchKernelGGL(k_audio_frames, dim3(n), dim3(kAudioThreads), 0, s, a, src, enc, acts);
        return check_launch(what);
    }
    RN_REQUIRE(workspace, "%s: workspace of n * 8 * dim_aud floats is required", what);
    hipLaunchKernelGGL(k_audio_frames, dim3(n * kSeq), dim3(kAudioThreads), 0, s, a, src, workspace, acts);
    hipLaunchKernelGGL(k_audio_attend, dim3(n), dim3(kAudioThreads), 0, s, a, workspace, enc);
    return check_launch(what);
}

int rn_audio_encode_windows(const rn_audio_weights_t *w, const float *auds, uint32_t n, float *enc, float *workspace,
                            rn_stream_t stream) {
    if (n == 0) return RN_OK;
    if (int rc = check_audio(w)) return rc;
    RN_REQUIRE(auds && enc, "audio_encode_windows: null pointer");
    return launch_audio(w, Source{auds, 0u, 0u, 0}, n, enc, workspace, as_stream(stream), "audio_encode_windows");
}

int rn_audio_encode_stream(const rn_audio_weights_t *w, const float *feats, uint32_t T, uint32_t first, uint32_t n,
                           float *enc, float *workspace, rn_stream_t stream) {
    if (n == 0) return RN_OK;
    if (int rc = check_audio(w)) return rc;
    RN_REQUIRE(feats && enc, "audio_encode_stream: null pointer");
    RN_REQUIRE(w->has_att && T >= 8, "audio_encode_stream: needs the attention window (has_att) and a stream of >= 8 frames");
    return launch_audio(w, Source{feats, T, first, 1}, n, enc, workspace, as_stream(stream), "audio_encode_stream");
}

size_t rn_audio_train_acts_floats(uint32_t n, int has_att) { return (size_t)n * (has_att ? kSeq : 1) * kActs; }

int rn_audio_encode_windows_train(const rn_audio_weights_t *w, const float *auds, uint32_t n, float *enc, float *workspace, float *acts,
                                  rn_stream_t stream) {
    if (n == 0) return RN_OK;
    if (int rc = check_audio(w)) return rc;
    RN_REQUIRE(auds && enc && acts, "audio_encode_windows_train: null pointer");
    return launch_audio(w, Source{auds, 0u, 0u, 0}, n, enc, workspace, as_stream(stream), "audio_encode_windows_train", acts);
}

static int audio_backward(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes, const float *grad_enc,
                          const rn_audio_grads_t *grads, float *grad_codes, const float *acts, rn_stream_t stream);

int rn_audio_encode_windows_backward(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes,
                                     const float *grad_enc, const rn_audio_grads_t *grads, float *grad_codes, rn_stream_t stream) {
    return audio_backward(w, auds, n, codes, grad_enc, grads, grad_codes, nullptr, stream);
}

int rn_audio_encode_windows_backward_acts(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes,
                                          const float *grad_enc, const rn_audio_grads_t *grads, float *grad_codes, const float *acts,
                                          rn_stream_t stream) {
    RN_REQUIRE(acts, "audio_encode_windows_backward_acts: the activations kept by rn_audio_encode_windows_train are required");
    return audio_backward(w, auds, n, codes, grad_enc, grads, grad_codes, acts, stream);
}

static int audio_backward(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes, const float *grad_enc,
                          const rn_audio_grads_t *grads, float *grad_codes, const float *acts, rn_stream_t stream) {
    if (n == 0) return RN_OK;
    if (int rc = check_audio(w)) return rc;
    RN_REQUIRE(auds && grad_enc && grads, "audio_encode_windows_backward: null pointer");
    AudioG g{};
    for (int l = 0; l < 4; l++) { g.conv_w[l] = grads->conv_w[l]; g.conv_b[l] = grads->conv_b[l]; RN_REQUIRE(g.conv_w[l] && g.conv_b[l], "audio backward: null AudioNet gradient buffer"); }
    for (int l = 0; l < 2; l++) { g.fc_w[l] = grads->fc_w[l]; g.fc_b[l] = grads->fc_b[l]; RN_REQUIRE(g.fc_w[l] && g.fc_b[l], "audio backward: null AudioNet gradient buffer"); }
    const AudioW a = audio_w(w);
    const Source src{auds, 0u, 0u, 0};
    hipStream_t s = as_stream(stream);
    if (!a.has_att) {
        hipLaunchKernelGGL(k_audio_frames_bwd, dim3(n), dim3(kAudioThreads), 0, s, a, g, src, grad_enc, acts);
        return check_launch("audio_encode_windows_backward");
    }
    RN_REQUIRE(codes && grad_codes, "audio_encode_windows_backward: the forward's per-frame codes and n * 8 * dim_aud floats of scratch are required");
    for (int l = 0; l < 5; l++) { g.att_conv_w[l] = grads->att_conv_w[l]; g.att_conv_b[l] = grads->att_conv_b[l]; RN_REQUIRE(g.att_conv_w[l] && g.att_conv_b[l], "audio backward: null AudioAttNet gradient buffer"); }
    g.att_fc_w = grads->att_fc_w; g.att_fc_b = grads->att_fc_b;
    RN_REQUIRE(g.att_fc_w && g.att_fc_b, "audio backward: null AudioAttNet gradient buffer");
    hipLaunchKernelGGL(k_audio_attend_bwd, dim3(n), dim3(kAudioThreads), 0, s, a, g, codes, grad_enc, grad_codes);
    hipLaunchKernelGGL(k_audio_frames_bwd, dim3(n * kSeq), dim3(kAudioThreads), 0, s, a, g, src, grad_codes, acts);
    return check_launch("audio_encode_windows_backward");
}

int rn_audio_smooth(const float *enc, uint32_t n, uint32_t dim, float lambda, float *state, int state_valid,
                    rn_stream_t stream) {
    if (n == 0) return RN_OK;
    RN_REQUIRE(enc && state && dim >= 1, "audio_smooth: null pointer");
    hipLaunchKernelGGL(k_audio_smooth, dim3(div_up(dim, 64)), dim3(64), 0, as_stream(stream), enc, n, dim, lambda, state,
                       state_valid, static_cast<float *>(nullptr));
    return check_launch("audio_smooth");
}

int rn_audio_smooth_seq(const float *enc, uint32_t n, uint32_t dim, float lambda, float *state, int state_valid, float *out,
                        rn_stream_t stream) {
    if (n == 0) return RN_OK;
    RN_REQUIRE(enc && state && out && dim >= 1, "audio_smooth_seq: null pointer");
    hipLaunchKernelGGL(k_audio_smooth, dim3(div_up(dim, 64)), dim3(64), 0, as_stream(stream), enc, n, dim, lambda, state,
                       state_valid, out);
    return check_launch("audio_smooth_seq");
}

}  // extern "C"
