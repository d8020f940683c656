// LayerNorm / RMSNorm forward + backward for gfx950 (HBM-bound row kernels).
//
// One 64-lane wave owns one row; the row lives in registers as VPL vectors of VW = 8 bf16 per lane (16-byte
// coalesced loads, 1 KiB per wave-instruction; VW = 4 when D % 8 != 0), statistics in fp32 with wave shuffles only - no LDS, no
// barriers - 4 rows per 256-thread block, grid-stride over rows.  Algorithmic traffic: read x + write y
// (fwd), read x,dy + write dx (bwd) - SURVEY.md §8(d).
//
// Oracle semantics reproduced exactly:
//   LayerNorm  (nn.LayerNorm, modeling_audioflamingo3.py:207,212,335; eps 1e-5): fp32 statistics,
//              y = bf16((x-mean)*rstd*w + b).
//   RMSNorm    (Qwen2RMSNorm.forward, modeling_qwen2.py:247-252; eps 1e-6): x32 = float(x);
//              xh = bf16(x32 * rsqrt(mean(x32^2)+eps));  y = bf16(w * xh)   (cast BEFORE the weight multiply).
// Weight/bias gradients are reduced in two stages: each block keeps fp32 partial column sums for the rows it
// visited and writes one partial row to a workspace [nblocks, 2, D]; afk_colsum_partials folds them.
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;

// VW = elements per lane-vector: 8 (16-byte accesses, 1 KiB per wave-instruction) whenever D % 8 == 0, else 4
template <int VPL, bool RMS, int VW>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                       const bf16* __restrict__ b, bf16* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    typedef __attribute__((ext_vector_type(VW))) __bf16 bvec;
    const int nvec = D / VW;
    const float invD = 1.f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
        const bf16* xr = x + row * D;
        float v[VPL][VW];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                const bvec t = *(const bvec*)(xr + VW * vi);
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    v[i][e] = (float)t[e];
                    s += v[i][e];
                    ss += v[i][e] * v[i][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[i][e] = 0.f;
            }
        }
        float mean = 0.f, rstd;
        if (RMS) {
            ss = wave_sum(ss);
            rstd = rsqrtf(ss * invD + eps);
        } else {
            s = wave_sum(s);
            mean = s * invD;
            // two-pass variance on the register copy (matches ATen's numerically careful path)
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int vi = lane + 64 * i;
                if (vi < nvec) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        const float d = v[i][e] - mean;
                        q += d * d;
                    }
                }
            }
            q = wave_sum(q);
            rstd = rsqrtf(q * invD + eps);
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        bf16* yr = y + row * D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                const bvec wv4 = *(const bvec*)(w + VW * vi);
                bvec o;
                if (RMS) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) o[e] = (bf16)((float)wv4[e] * rbf(v[i][e] * rstd));
                } else {
                    const bvec bv4 = *(const bvec*)(b + VW * vi);
#pragma unroll
                    for (int e = 0; e < VW; ++e) o[e] = (bf16)((v[i][e] - mean) * rstd * (float)wv4[e] + (float)bv4[e]);
                }
                *(bvec*)(yr + VW * vi) = o;
            }
        }
    }
}

// backward.  dx for one row; per-block fp32 partial sums of dw (and db) over the rows this block visits.
//   LN : xh=(x-mean)*rstd; g=dy*w;  dx = rstd*(g - mean(g) - xh*mean(g*xh));  dw+=dy*xh; db+=dy
//   RMS: xh=x*rstd;        g=dy*w;  dx = rstd*(g - xh*mean(g*xh));            dw+=dy*bf16(xh)
template <int VPL, bool RMS, int VW>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                       const bf16* __restrict__ dy, const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in, bf16* __restrict__ dx,
                                                       const bf16* __restrict__ dx_add, float* __restrict__ partials,
                                                       int64_t rows, int D) {
    typedef __attribute__((ext_vector_type(VW))) __bf16 bvec;
    typedef __attribute__((ext_vector_type(VW / 2))) unsigned packed_t;
    __shared__ float red[2][VPL * 64 * VW];  // cross-wave fold of the column partials
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nvec = D / VW;
    const float invD = 1.f / (float)D;
    float pw[VPL][VW], pb[VPL][VW];
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < VW; ++e) pw[i][e] = pb[i][e] = 0.f;
    for (int c = threadIdx.x; c < VPL * 64 * VW; c += 256) red[0][c] = red[1][c] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
        const bf16* xr = x + row * D;
        const bf16* dyr = dy + row * D;
        const float mean = RMS ? 0.f : mean_in[row];
        const float rstd = rstd_in[row];
        // the row stays in registers as PACKED bf16 (2 VGPRs per 4 elements): 56 instead of 112 VGPRs at D=3584, which
        // is what lets 3 waves/SIMD be resident and keep enough loads in flight for an HBM-bound kernel
        bvec xv[VPL], dv[VPL];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                xv[i] = *(const bvec*)(xr + VW * vi);
                dv[i] = *(const bvec*)(dyr + VW * vi);
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) xv[i][e] = dv[i][e] = (bf16)0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                const bvec wt = *(const bvec*)(w + VW * vi);
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const float d = (float)dv[i][e];
                    const float xh = ((float)xv[i][e] - mean) * rstd;
                    const float g = d * (float)wt[e];
                    sg += g;
                    sgx += g * xh;
                    pw[i][e] += d * (RMS ? rbf(xh) : xh);
                    if (!RMS) pb[i][e] += d;
                }
            }
        }
        sgx = wave_sum(sgx) * invD;
        sg = RMS ? 0.f : wave_sum(sg) * invD;
        // opaque to the optimiser: the second pass must re-derive xh/g from the packed row instead of keeping the
        // first pass's fp32 values alive (that is what blew the register budget)
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            asm volatile("" : "+v"(*(packed_t*)&xv[i]));
            asm volatile("" : "+v"(*(packed_t*)&dv[i]));
        }
        bf16* dxr = dx + row * D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                const bvec wt = *(const bvec*)(w + VW * vi);
                bvec o;
                float r[VW];
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const float xh = ((float)xv[i][e] - mean) * rstd;
                    const float g = (float)dv[i][e] * (float)wt[e];
                    r[e] = rstd * (g - sg - xh * sgx);
                }
                if (dx_add) {  // fused residual-gradient merge: dx = bf16(norm-branch) + skip-branch
                    const bvec a = *(const bvec*)(dx_add + row * D + VW * vi);
#pragma unroll
                    for (int e = 0; e < VW; ++e) r[e] = rbf(r[e]) + (float)a[e];
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) o[e] = (bf16)r[e];
                *(bvec*)(dxr + VW * vi) = o;
            }
        }
    }
    // fold the 4 waves' partials in a FIXED order (wave 0,1,2,3 take turns: bit-reproducible, unlike LDS float atomics)
    // and write this block's partial row: partials[block][0][D] = dw, [1][D] = db
    __syncthreads();
#pragma unroll
    for (int turn = 0; turn < ROWS_PER_BLOCK; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int i = 0; i < VPL; ++i)
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    red[0][(lane + 64 * i) * VW + e] += pw[i][e];
                    if (!RMS) red[1][(lane + 64 * i) * VW + e] += pb[i][e];
                }
        }
        __syncthreads();
    }
    float* outp = partials + (int64_t)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        outp[c] = red[0][c];
        outp[D + c] = red[1][c];
    }
}

// out[c] (+)= sum_p partials[p][which][c]   -> bf16 grads.
// Block = 64 columns (16 lanes x float4: 256-byte row segments) x 16 part-slices; every thread streams nparts / 16 float4 loads with four
// independent accumulators (the round-2 form - 32 columns x 8 slices of scalar loads, one dependent chain of 64 loads per thread - took 20 us
// for the 7 MB of a decoder RMSNorm: 187 launches = 3.8 ms of the step), then the 16 slices are folded through LDS in a fixed order
// (bit-reproducible).  D % 4 == 0 is guaranteed by the callers.
// gridDim.y = 2 folds the second partial plane (offset D: LayerNorm's db) into out2 in the same launch.
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partials, int nparts, int D,
                                                            int stride, bf16* __restrict__ out, bf16* __restrict__ out2, int accumulate) {
    __shared__ f32x4 red[16][17];
    const int offset = blockIdx.y ? D : 0;
    if (blockIdx.y) out = out2;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = (blockIdx.x * 16 + cl) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < D) {
        const float* base = partials + offset + c;
        int p = sl;
        for (; p + 48 < nparts; p += 64) {
            s0 += *(const f32x4*)(base + (int64_t)p * stride);
            s1 += *(const f32x4*)(base + (int64_t)(p + 16) * stride);
            s2 += *(const f32x4*)(base + (int64_t)(p + 32) * stride);
            s3 += *(const f32x4*)(base + (int64_t)(p + 48) * stride);
        }
        for (; p < nparts; p += 16) s0 += *(const f32x4*)(base + (int64_t)p * stride);
    }
    red[sl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && c < D) {
        f32x4 t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        bf16x4 o;
        if (accumulate) {
            const bf16x4 old = *(const bf16x4*)(out + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += (float)old[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)t[e];
        *(bf16x4*)(out + c) = o;
    }
}

int norm_grid(int64_t rows) {
    int64_t g = afk_cdiv(rows, ROWS_PER_BLOCK);
    if (g > 2048) g = 2048;
    return (int)g;
}

template <bool RMS>
int launch_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows, int D,
               float eps, hipStream_t st) {
    const int g = norm_grid(rows);
#define AFK_NF(V, W)                                                                                                 \
    hipLaunchKernelGGL((norm_fwd_kernel<V, RMS, W>), dim3(g), dim3(256), 0, st, (const bf16*)x, (const bf16*)w,    \
                       (const bf16*)b, (bf16*)y, mean, rstd, rows, D, eps)
    if (D % 8 == 0) {
        if (D <= 512) AFK_NF(1, 8);
        else if (D <= 1536) AFK_NF(3, 8);
        else if (D <= 3584) AFK_NF(7, 8);
        else AFK_NF(16, 8);
    } else {
        if (D <= 512) AFK_NF(2, 4);
        else if (D <= 1280) AFK_NF(5, 4);
        else if (D <= 3584) AFK_NF(14, 4);
        else AFK_NF(32, 4);
    }
#undef AFK_NF
    return AFK_OK;
}

template <bool RMS>
int launch_bwd(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, void* dx,
               const void* dx_add, float* partials, int nblocks, int64_t rows, int D, hipStream_t st) {
#define AFK_NB(V, W)                                                                                                 \
    hipLaunchKernelGGL((norm_bwd_kernel<V, RMS, W>), dim3(nblocks), dim3(256), 0, st, (const bf16*)x, (const bf16*)w, \
                       (const bf16*)dy, mean, rstd, (bf16*)dx, (const bf16*)dx_add, partials, rows, D)
    if (D % 8 == 0) {
        if (D <= 512) AFK_NB(1, 8);
        else if (D <= 1536) AFK_NB(3, 8);
        else AFK_NB(7, 8);
    } else {
        if (D <= 512) AFK_NB(2, 4);
        else if (D <= 1280) AFK_NB(5, 4);
        else AFK_NB(14, 4);
    }
#undef AFK_NB
    return AFK_OK;
}

}  // namespace

extern "C" int afk_norm_bwd_blocks(int64_t rows) {
    int64_t g = afk_cdiv(rows, ROWS_PER_BLOCK);
    if (g > 512) g = 512;
    return (int)g;
}

extern "C" int afk_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                                 int64_t rows, int D, float eps, void* stream) {
    AFK_REQUIRE(x && w && b && y && mean && rstd, "afk_layernorm_fwd: null pointer");
    AFK_REQUIRE(D % 4 == 0 && D <= 8192 && rows > 0, "afk_layernorm_fwd: unsupported D=%d", D);
    launch_fwd<false>(x, w, b, y, mean, rstd, rows, D, eps, (hipStream_t)stream);
    AFK_LAUNCH_CHECK("afk_layernorm_fwd");
    return AFK_OK;
}

extern "C" int afk_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int D, float eps,
                               void* stream) {
    AFK_REQUIRE(x && w && y && rstd, "afk_rmsnorm_fwd: null pointer");
    AFK_REQUIRE(D % 4 == 0 && D <= 8192 && rows > 0, "afk_rmsnorm_fwd: unsupported D=%d", D);
    launch_fwd<true>(x, w, nullptr, y, nullptr, rstd, rows, D, eps, (hipStream_t)stream);
    AFK_LAUNCH_CHECK("afk_rmsnorm_fwd");
    return AFK_OK;
}

extern "C" int afk_layernorm_bwd(const void* x, const void* w, const void* dy, const float* mean, const float* rstd,
                                 void* dx, const void* dx_add, void* dw, void* db, int accumulate, float* workspace,
                                 int64_t rows, int D, void* stream) {
    AFK_REQUIRE(x && w && dy && mean && rstd && dx && dw && db && workspace, "afk_layernorm_bwd: null pointer");
    AFK_REQUIRE(D % 4 == 0 && D <= 3584 && rows > 0, "afk_layernorm_bwd: unsupported D=%d", D);
    const int nb = afk_norm_bwd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    launch_bwd<false>(x, w, dy, mean, rstd, dx, dx_add, workspace, nb, rows, D, st);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(D, 64), 2), dim3(256), 0, st, workspace, nb, D, 2 * D, (bf16*)dw, (bf16*)db,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_layernorm_bwd");
    return AFK_OK;
}

extern "C" int afk_rmsnorm_bwd(const void* x, const void* w, const void* dy, const float* rstd, void* dx,
                               const void* dx_add, void* dw, int accumulate, float* workspace, int64_t rows, int D,
                               void* stream) {
    AFK_REQUIRE(x && w && dy && rstd && dx && dw && workspace, "afk_rmsnorm_bwd: null pointer");
    AFK_REQUIRE(D % 4 == 0 && D <= 3584 && rows > 0, "afk_rmsnorm_bwd: unsupported D=%d", D);
    const int nb = afk_norm_bwd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    launch_bwd<true>(x, w, dy, nullptr, rstd, dx, dx_add, workspace, nb, rows, D, st);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(D, 64), 1), dim3(256), 0, st, workspace, nb, D, 2 * D, (bf16*)dw, (bf16*)nullptr,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_rmsnorm_bwd");
    return AFK_OK;
}
