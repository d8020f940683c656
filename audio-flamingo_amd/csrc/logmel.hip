// Whisper-style log-mel frontend on gfx950 (fp32):  waveform [W, N] -> input_features [W, n_mels, T].
//
// Oracle: WhisperFeatureExtractor._torch_extract_fbank_features, models/whisper/feature_extraction_whisper.py:135-168
//   hann(400, periodic) :141 -> torch.stft(n_fft=400, hop=160, center=True, reflect pad) :149 -> drop last frame,
//   |.|^2 :154 -> mel_filters.T @ mag :157 -> log10(clamp 1e-10) :159 -> per-window max-8 floor :161-162 -> (x+4)/4 :165.
// A 400-point transform is not a power of two, so the DFT is done directly against a precomputed, hann-folded
// basis (host builds it in fp64, stores fp32): cosb/sinb = [n_fft, nbins_pad], bins contiguous so that lane k
// reads bin k coalesced.  One block = 32 frames of one window: the 5360-sample span (reflect-indexed) is staged
// in LDS once and broadcast-read as float4 by all lanes; lane k accumulates re/im of bin k for all 32 frames in
// registers (64 fp32 accumulators), so the basis streams from L2 exactly once per block.  Power spectrum and the
// 128x201 mel contraction stay in LDS; the [mel][frame] tile is written out in 128-byte rows.  The per-window max
// is folded through an order-preserving integer atomicMax; a second tiny kernel applies floor/scale.
// HBM traffic per window: 1.92 MB in, 1.54 MB out (+1.54 MB read/rewrite by the finish pass).
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, FT = 32;  // frames per block
constexpr int SPAN = (FT - 1) * HOP + NFFT;                 // 5360 samples
constexpr int MAXMEL = 128;

__device__ __forceinline__ int float_to_ordered(float f) {
    const int i = __float_as_int(f);
    return (i >= 0) ? i : (i ^ 0x7fffffff);
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float((i >= 0) ? i : (i ^ 0x7fffffff)); }

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, int64_t nsamp, const float* __restrict__ cosb,
                                                     const float* __restrict__ sinb, int nbins_pad, const float* __restrict__ melT,
                                                     int nmel, float* __restrict__ out_raw, int T, int* __restrict__ wmax) {
    __shared__ __attribute__((aligned(16))) float seg[SPAN + 16];
    __shared__ float pw[FT][NBIN + 3];
    __shared__ float lm[MAXMEL][FT + 1];
    __shared__ float red[4];
    const int w = blockIdx.y, f0 = blockIdx.x * FT;
    const int tid = threadIdx.x;
    const float* x = wav + (int64_t)w * nsamp;
    // stage samples: padded index i = f0*HOP + j, source = reflect(i - NFFT/2)
    for (int j = tid; j < SPAN; j += 256) {
        int64_t src = (int64_t)f0 * HOP + j - NFFT / 2;
        if (src < 0) src = -src;
        if (src >= nsamp) src = 2 * (nsamp - 1) - src;
        seg[j] = (src >= 0 && src < nsamp) ? x[src] : 0.f;
    }
    __syncthreads();
    // DFT: lane k owns bin k
    const int k = tid;
    if (k < NBIN) {
        float re[FT], im[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) re[f] = im[f] = 0.f;
        for (int n = 0; n < NFFT; n += 4) {
            float c[4], s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = cosb[(n + j) * nbins_pad + k];
                s[j] = sinb[(n + j) * nbins_pad + k];
            }
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                const f32x4 xv = *(const f32x4*)&seg[f * HOP + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    re[f] = fmaf(xv[j], c[j], re[f]);
                    im[f] = fmaf(xv[j], s[j], im[f]);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < FT; ++f) pw[f][k] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();
    // mel + log10: thread -> (mel m = tid % nmel, frames f = tid / nmel + (256/nmel) * j)
    float lmax = -INFINITY;
    {
        const int m = tid % nmel, fstep = 256 / nmel;
        for (int f = tid / nmel; f < FT; f += fstep) {
            float acc = 0.f;
            for (int kk = 0; kk < NBIN; ++kk) acc = fmaf(melT[kk * nmel + m], pw[f][kk], acc);
            const float v = log10f(fmaxf(acc, 1e-10f));
            lm[m][f] = v;
            if (f0 + f < T) lmax = fmaxf(lmax, v);
        }
    }
    lmax = block_max<4>(lmax, red);
    if (tid == 0) atomicMax(&wmax[w], float_to_ordered(lmax));
    __syncthreads();
    // write [mel][frame] rows: 32 consecutive frames (128 B) per mel row
    for (int idx = tid; idx < nmel * FT; idx += 256) {
        const int m = idx / FT, f = idx % FT;
        if (f0 + f < T) out_raw[((int64_t)w * nmel + m) * T + f0 + f] = lm[m][f];
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void logmel_finish_kernel(const float* raw, const int* __restrict__ wmax,
                                                            OutT* out, int64_t per_window, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i / per_window);
        const float mx = ordered_to_float(wmax[w]);
        out[i] = (OutT)((fmaxf(raw[i], mx - 8.f) + 4.f) * 0.25f);
    }
}

__global__ void fill_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

extern "C" int afk_logmel(const float* wav, int W, int64_t nsamp, const float* cosb, const float* sinb, int nbins_pad,
                          const float* melT, int nmel, float* raw_ws, int* wmax_ws, void* out, int out_is_bf16, void* stream) {
    AFK_REQUIRE(wav && cosb && sinb && melT && raw_ws && wmax_ws && out, "afk_logmel: null pointer");
    AFK_REQUIRE(W > 0 && nsamp >= NFFT && nsamp % HOP == 0, "afk_logmel: nsamp must be a positive multiple of %d", HOP);
    AFK_REQUIRE(nmel > 0 && nmel <= MAXMEL && 256 % nmel == 0, "afk_logmel: n_mels=%d unsupported (must divide 256, <=128)", nmel);
    AFK_REQUIRE(nbins_pad >= NBIN, "afk_logmel: basis must have >= %d bins per row", NBIN);
    AFK_REQUIRE(((uintptr_t)wav % 4 == 0), "afk_logmel: misaligned waveform");
    const int T = (int)(nsamp / HOP);  // frames kept (the centered STFT yields T+1, the last is dropped)
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)afk_cdiv(W, 256)), dim3(256), 0, st, wmax_ws, W, (int)0x80000000);
    dim3 grid((unsigned)afk_cdiv(T, FT), (unsigned)W);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, st, wav, nsamp, cosb, sinb, nbins_pad, melT, nmel, raw_ws, T, wmax_ws);
    const int64_t per_window = (int64_t)nmel * T, total = per_window * W;
    int g = (int)afk_cdiv(total, 256);
    if (g > 4096) g = 4096;
    if (out_is_bf16)
        hipLaunchKernelGGL(logmel_finish_kernel<bf16>, dim3(g), dim3(256), 0, st, raw_ws, wmax_ws, (bf16*)out, per_window, total);
    else
        hipLaunchKernelGGL(logmel_finish_kernel<float>, dim3(g), dim3(256), 0, st, raw_ws, wmax_ws, (float*)out, per_window, total);
    AFK_LAUNCH_CHECK("afk_logmel");
    return AFK_OK;
}
