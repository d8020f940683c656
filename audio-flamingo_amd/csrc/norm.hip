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
// visited and writes one partial row to a workspace [nblocks, 2, D]; fold_partials_kernel folds them.
//
// Backward, round 4 (norm_bwd_cols_kernel, D % 8 == 0): the row-per-wave form above needs the whole row AND fp32 dw/db partials of the
// whole row in every lane's registers (246 VGPRs at D = 3584: two waves per SIMD, one row at a time, two dependent HBM round trips per
// row - 3.2 TB/s).  The column-owned form turns the block by 90 degrees: thread t owns the 8 columns 8t..8t+7 of EVERY row the block
// visits (one 16-byte vector per tensor and row), so the dw/db partials are 8 + 8 registers, a group of R = 4 rows is 12 independent
// 16-byte loads per thread issued back to back (x, dy, dx_add), and the per-row statistics (sum of g, sum of g*xh over D) go through one
// wave reduction + one LDS exchange + ONE barrier per group (double-buffered strip).  No cross-wave fold of the partials at the end: a
// column lives in exactly one thread.
#include <algorithm>
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

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
// gridDim.y = 3 (round 6) folds a third, separately laid out plane [nparts][stride3] into out3 as well: the column sums of the OUTPUT of the column-owned
// LayerNorm backward = the bias gradient of the Linear below it (afk_layernorm_bwd_colsum).
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partials, int nparts, int D,
                                                            int stride, bf16* __restrict__ out, bf16* __restrict__ out2, int accumulate,
                                                            const float* __restrict__ partials3 = nullptr, int stride3 = 0, bf16* __restrict__ out3 = nullptr,
                                                            int accumulate3 = 0) {
    __shared__ f32x4 red[16][17];
    int offset = blockIdx.y ? D : 0;
    if (blockIdx.y) out = out2;
    if (blockIdx.y == 2) {
        partials = partials3, offset = 0, stride = stride3, out = out3, accumulate = accumulate3;
    }
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

// Column-owned backward (see the header).  blockDim.x = D / 8 rounded up to whole waves; R rows per group.
// CS (round 6): the kernel owns its columns over every row anyway, so it also sums the dx it WRITES (the bf16 values, as a colsum pass over dx would read them)
// into cs_partials[block][D]: in a pre-LN block dx = (norm branch) + (skip branch) is the gradient of the residual stream, i.e. of the output of the Linear
// that precedes the norm - its column sums are that Linear's bias gradient (encoder: out_proj below final_layer_norm, the LOWER layer's fc2 below self_attn_layer_norm).
template <bool RMS, bool ADD, int R, bool CS = false>
__global__ __launch_bounds__(512, 3) void norm_bwd_cols_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ dy,
                                                             const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                             bf16* __restrict__ dx, const bf16* __restrict__ dx_add, float* __restrict__ partials,
                                                             int64_t rows, int D, float* __restrict__ cs_partials = nullptr) {
    constexpr int NS = RMS ? R : 2 * R;                      // statistics per group: sum(g * xh) [, sum(g)] per row
    __shared__ float red[2][16][NS];                         // [parity][wave][stat]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int col = threadIdx.x * 8;
    const bool live = col < D;
    const int colc = live ? col : D - 8;
    const float invD = 1.f / (float)D;
    float wf[8], pw[8], pb[8], pc[CS ? 8 : 1];
    {
        const bf16x8 wt = *(const bf16x8*)(w + colc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wf[e] = (float)wt[e];
            pw[e] = pb[e] = 0.f;
            if (CS) pc[e] = 0.f;
        }
    }
    const int64_t ngroups = (rows + R - 1) / R;
    int par = 0;
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x, par ^= 1) {
        const int64_t r0 = g * R;
        bf16x8 xv[R], dv[R], av[R];
        float mean[R], rstd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = min(r0 + r, rows - 1);       // block-uniform clamp: a ragged last group re-reads the last row, its results are dropped
            // branch-free: lanes beyond D (the last wave of a block, D / 8 not a multiple of 64) read the row's last vector and drop it
            xv[r] = *(const bf16x8*)(x + row * D + colc);
            dv[r] = *(const bf16x8*)(dy + row * D + colc);
            if (ADD) av[r] = *(const bf16x8*)(dx_add + row * D + colc);
            if (!live) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[r][e] = (bf16)0.f;   // g = 0: contributes nothing to the row statistics
            }
            mean[r] = RMS ? 0.f : mean_in[row];
            rstd[r] = rstd_in[row];
        }
        float st[NS];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = r0 + r < rows;                // block-uniform
            float sgx = 0.f, sg = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)dv[r][e];
                const float xh = ((float)xv[r][e] - mean[r]) * rstd[r];
                const float gg = d * wf[e];
                sgx += gg * xh;
                sg += gg;
                if (valid) {
                    pw[e] += d * (RMS ? rbf(xh) : xh);
                    if (!RMS) pb[e] += d;
                }
            }
            st[r] = sgx;
            if (!RMS) st[R + r] = sg;
            __builtin_amdgcn_sched_barrier(0);   // one row's fp32 expansion at a time: interleaving the four rows costs 60 more VGPRs (a wave per SIMD)
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) st[i] = wave_sum(st[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NS; ++i) red[par][wv][i] = st[i];
        }
        // opaque to the optimiser: the second pass re-derives xh / g from the PACKED rows instead of keeping 64 fp32 values of the first pass
        // alive across the barrier (184-228 VGPRs -> two waves per SIMD; the point of this kernel is many loads in flight)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            asm volatile("" : "+v"(*(u32x4*)&xv[r]));
            asm volatile("" : "+v"(*(u32x4*)&dv[r]));
        }
        __syncthreads();   // one barrier per group: the strip of the NEXT group is the other parity, and a wave can only get two groups ahead by passing the next barrier
        float tot[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) tot[i] = 0.f;
#pragma unroll 1
        for (int k = 0; k < nw; ++k) {                        // fixed order: bit-reproducible
#pragma unroll
            for (int i = 0; i < NS; ++i) tot[i] += red[par][k][i];
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r0 + r >= rows) break;
                const float sgx = tot[r] * invD, sg = RMS ? 0.f : tot[R + r] * invD;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[r][e] - mean[r]) * rstd[r];
                    const float gg = (float)dv[r][e] * wf[e];
                    float v = rstd[r] * (gg - sg - xh * sgx);
                    if (ADD) v = rbf(v) + (float)av[r][e];   // fused residual-gradient merge: dx = bf16(norm-branch) + skip-branch
                    o[e] = (bf16)v;
                    if (CS) pc[e] += (float)o[e];
                }
                *(bf16x8*)(dx + (r0 + r) * D + col) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // this block's partial row: partials[block][0][D] = dw, [1][D] = db - each column written by its one owner
    if (live) {
        float* outp = partials + (int64_t)blockIdx.x * 2 * D + col;
        *(f32x4*)(outp) = f32x4{pw[0], pw[1], pw[2], pw[3]};
        *(f32x4*)(outp + 4) = f32x4{pw[4], pw[5], pw[6], pw[7]};
        if (!RMS) {
            *(f32x4*)(outp + D) = f32x4{pb[0], pb[1], pb[2], pb[3]};
            *(f32x4*)(outp + D + 4) = f32x4{pb[4], pb[5], pb[6], pb[7]};
        }
        if (CS) {
            float* outc = cs_partials + (int64_t)blockIdx.x * D + col;
            *(f32x4*)(outc) = f32x4{pc[0], pc[1], pc[2], pc[3]};
            *(f32x4*)(outc + 4) = f32x4{pc[4], pc[5], pc[6], pc[7]};
        }
    }
}

// rows per group of the column-owned backward: 2 by default (99-109 VGPRs: four waves per SIMD = two 7-wave blocks per CU at D = 3584);
// AFK_NORM_BWD_R=4 selects the 4-row groups of the RMSNorm instantiation (139-157 VGPRs, one 7-wave block per CU) for A/B runs
int norm_bwd_rows_per_group(bool rms) {
    static const int r = [] { const char* e = getenv("AFK_NORM_BWD_R"); return (e && e[0] == '4') ? 4 : 2; }();
    return rms ? r : 2;
}
int norm_bwd_cols_blocks(int64_t rows, int D, bool rms) {
    // wide rows (448 threads = 7 waves at D = 3584): two blocks per CU; narrow rows (3 waves at D = 1280): four
    const int64_t cap = (D / 8 > 256) ? 512 : 1024;
    const int64_t g = afk_cdiv(rows, norm_bwd_rows_per_group(rms));
    return (int)(g < cap ? g : cap);
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
int launch_bwd_cols(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, void* dx,
                    const void* dx_add, float* partials, int nblocks, int64_t rows, int D, hipStream_t st) {
    const int nt = (int)afk_cdiv(D / 8, 64) * 64;
#define AFK_NBC(ADD, R)                                                                                                                            \
    hipLaunchKernelGGL((norm_bwd_cols_kernel<RMS, ADD, R>), dim3(nblocks), dim3(nt), 0, st, (const bf16*)x, (const bf16*)w, (const bf16*)dy, mean, \
                       rstd, (bf16*)dx, (const bf16*)dx_add, partials, rows, D)
    if (RMS && norm_bwd_rows_per_group(true) == 4) {
        if (dx_add) AFK_NBC(true, (RMS ? 4 : 2)); else AFK_NBC(false, (RMS ? 4 : 2));
    } else {
        if (dx_add) AFK_NBC(true, 2); else AFK_NBC(false, 2);
    }
#undef AFK_NBC
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

// upper bound of the partial rows either backward form writes (the workspace is sized with it)
extern "C" int afk_norm_bwd_blocks(int64_t rows) {
    int64_t g = afk_cdiv(rows, 2);   // the column-owned form runs one block per 2-row group (cap 512 / 1024), the row-per-wave form one per 4 rows (cap 512)
    if (g > 1024) g = 1024;
    return (int)g;
}
static int norm_bwd_rowwave_blocks(int64_t rows) {
    int64_t g = afk_cdiv(rows, ROWS_PER_BLOCK);
    if (g > 512) g = 512;
    return (int)g;
}
// AFK_NORM_BWD=rows restores the row-per-wave form (A/B)
static bool norm_bwd_use_cols(int D) {
    static const bool rows_form = [] { const char* e = getenv("AFK_NORM_BWD"); return e && e[0] == 'r'; }();
    return !rows_form && D % 8 == 0 && D / 8 <= 512;
}
#define AFK_NORM_BWD_ALIGN(name)                                                                                                               \
    AFK_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)dx_add % 16 == 0 && (uintptr_t)w % 16 == 0 && \
                    (uintptr_t)dw % 8 == 0 && (uintptr_t)workspace % 16 == 0,                                                               \
                name ": x / dy / dx / dx_add / w / workspace must be 16-byte aligned, dw / db 8-byte aligned (vector accesses)")

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
    AFK_REQUIRE(D % 4 == 0 && D <= 4096 && (D <= 3584 || D % 8 == 0) && rows > 0, "afk_layernorm_bwd: unsupported D=%d", D);
    AFK_NORM_BWD_ALIGN("afk_layernorm_bwd");
    AFK_REQUIRE((uintptr_t)db % 8 == 0, "afk_layernorm_bwd: db must be 8-byte aligned");
    const bool cols = norm_bwd_use_cols(D);
    const int nb = std::min(cols ? norm_bwd_cols_blocks(rows, D, false) : norm_bwd_rowwave_blocks(rows), afk_norm_bwd_blocks(rows));
    hipStream_t st = (hipStream_t)stream;
    if (cols) launch_bwd_cols<false>(x, w, dy, mean, rstd, dx, dx_add, workspace, nb, rows, D, st);
    else launch_bwd<false>(x, w, dy, mean, rstd, dx, dx_add, workspace, nb, rows, D, st);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(D, 64), 2), dim3(256), 0, st, workspace, nb, D, 2 * D, (bf16*)dw, (bf16*)db,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_layernorm_bwd");
    return AFK_OK;
}

// GELU backward, column-owned (round 6): dx = bf16(dy * gelu'(pre)) as gelu_bwd_kernel (elementwise.hip), and the column sums of the dx it writes - dx is the
// grad_output of the Linear that produced `pre` (encoder fc1), so the sums are its bias gradient and the column-sum pass over dx (123 MB per encoder layer at
// B = 8) disappears.  A thread owns 8 columns over the rows blockIdx.x, blockIdx.x + gridDim.x, ...; partials[blockIdx.x][C], folded by fold_partials_kernel.
namespace {
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ pre, bf16* __restrict__ dx, int64_t rows, int C,
                                                              float* __restrict__ partials) {
    const int col = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (col >= C) return;
    float pc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pc[e] = 0.f;
    const int64_t step = gridDim.x;
    int64_t r = blockIdx.x;
    for (; r + 3 * step < rows; r += 4 * step) {   // four rows in flight
        bf16x8 d[4], p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = *(const bf16x8*)(dy + (r + u * step) * C + col);
            p[u] = *(const bf16x8*)(pre + (r + u * step) * C + col);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (bf16)((float)d[u][e] * gelu_grad_f((float)p[u][e]));
                pc[e] += (float)o[e];
            }
            *(bf16x8*)(dx + (r + u * step) * C + col) = o;
        }
    }
    for (; r < rows; r += step) {
        const bf16x8 d = *(const bf16x8*)(dy + r * C + col), p = *(const bf16x8*)(pre + r * C + col);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (bf16)((float)d[e] * gelu_grad_f((float)p[e]));
            pc[e] += (float)o[e];
        }
        *(bf16x8*)(dx + r * C + col) = o;
    }
    float* outp = partials + (int64_t)blockIdx.x * C + col;
    *(f32x4*)(outp) = f32x4{pc[0], pc[1], pc[2], pc[3]};
    *(f32x4*)(outp + 4) = f32x4{pc[4], pc[5], pc[6], pc[7]};
}
}  // namespace

// partial rows of afk_gelu_bwd_colsum (its workspace holds that many x C floats)
extern "C" int afk_gelu_bwd_colsum_parts(int64_t rows) {
    static const int cap = [] { const char* e = getenv("AFK_GELU_CS_PARTS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();   // measurement knob: 256 / 512 / 1024 parts = 113.9 / 105.4 / 103.9 us for [12000, 5120] incl. the fold (elementwise form + column-sum pass: 82 + 25)
    return (int)(rows < cap ? (rows < 1 ? 1 : rows) : cap);
}

extern "C" int afk_gelu_bwd_colsum(const void* dy, const void* pre, void* dx, int64_t rows, int C, void* colsum, int colsum_accumulate, float* workspace,
                                   void* stream) {
    AFK_REQUIRE(dy && pre && dx && colsum && workspace && rows > 0 && C > 0 && C % 8 == 0, "afk_gelu_bwd_colsum: bad arguments (C %% 8 == 0)");
    AFK_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)pre % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)workspace % 16 == 0 && (uintptr_t)colsum % 8 == 0,
                "afk_gelu_bwd_colsum: dy / pre / dx / workspace must be 16-byte aligned, colsum 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int np = afk_gelu_bwd_colsum_parts(rows);
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((unsigned)np, (unsigned)afk_cdiv(C / 8, 256)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)pre, (bf16*)dx, rows, C,
                       workspace);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(C, 64), 1), dim3(256), 0, st, workspace, np, C, C, (bf16*)colsum, (bf16*)nullptr, colsum_accumulate);
    AFK_LAUNCH_CHECK("afk_gelu_bwd_colsum");
    return AFK_OK;
}

// LayerNorm backward with the residual merge (dx = bf16(norm branch) + dx_add) that ALSO returns the column sums of the dx it writes: colsum[c] (+)= sum_r dx[r][c]
// = the bias gradient of the Linear whose output this norm normalised (see norm_bwd_cols_kernel).  Column-owned form only (D % 8 == 0, D <= 4096);
// workspace: afk_norm_bwd_blocks(rows) x 3 x D floats.
extern "C" int afk_layernorm_bwd_colsum(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, void* dx, const void* dx_add,
                                        void* dw, void* db, int accumulate, void* colsum, int colsum_accumulate, float* workspace, int64_t rows, int D,
                                        void* stream) {
    AFK_REQUIRE(x && w && dy && mean && rstd && dx && dx_add && dw && db && colsum && workspace, "afk_layernorm_bwd_colsum: null pointer");
    AFK_REQUIRE(D % 8 == 0 && D / 8 <= 512 && rows > 0, "afk_layernorm_bwd_colsum: unsupported D=%d (D %% 8 == 0, D <= 4096)", D);
    AFK_NORM_BWD_ALIGN("afk_layernorm_bwd_colsum");
    AFK_REQUIRE((uintptr_t)db % 8 == 0 && (uintptr_t)colsum % 8 == 0, "afk_layernorm_bwd_colsum: db / colsum must be 8-byte aligned");
    const int nb = std::min(norm_bwd_cols_blocks(rows, D, false), afk_norm_bwd_blocks(rows));
    hipStream_t st = (hipStream_t)stream;
    float* cs = workspace + (int64_t)afk_norm_bwd_blocks(rows) * 2 * D;
    const int nt = (int)afk_cdiv(D / 8, 64) * 64;
    hipLaunchKernelGGL((norm_bwd_cols_kernel<false, true, 2, true>), dim3(nb), dim3(nt), 0, st, (const bf16*)x, (const bf16*)w, (const bf16*)dy, mean, rstd,
                       (bf16*)dx, (const bf16*)dx_add, workspace, rows, D, cs);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(D, 64), 3), dim3(256), 0, st, workspace, nb, D, 2 * D, (bf16*)dw, (bf16*)db, accumulate,
                       (const float*)cs, D, (bf16*)colsum, colsum_accumulate);
    AFK_LAUNCH_CHECK("afk_layernorm_bwd_colsum");
    return AFK_OK;
}

extern "C" int afk_rmsnorm_bwd(const void* x, const void* w, const void* dy, const float* rstd, void* dx,
                               const void* dx_add, void* dw, int accumulate, float* workspace, int64_t rows, int D,
                               void* stream) {
    AFK_REQUIRE(x && w && dy && rstd && dx && dw && workspace, "afk_rmsnorm_bwd: null pointer");
    AFK_REQUIRE(D % 4 == 0 && D <= 4096 && (D <= 3584 || D % 8 == 0) && rows > 0, "afk_rmsnorm_bwd: unsupported D=%d", D);
    AFK_NORM_BWD_ALIGN("afk_rmsnorm_bwd");
    const bool cols = norm_bwd_use_cols(D);
    const int nb = std::min(cols ? norm_bwd_cols_blocks(rows, D, true) : norm_bwd_rowwave_blocks(rows), afk_norm_bwd_blocks(rows));
    hipStream_t st = (hipStream_t)stream;
    if (cols) launch_bwd_cols<true>(x, w, dy, nullptr, rstd, dx, dx_add, workspace, nb, rows, D, st);
    else launch_bwd<true>(x, w, dy, nullptr, rstd, dx, dx_add, workspace, nb, rows, D, st);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)afk_cdiv(D, 64), 1), dim3(256), 0, st, workspace, nb, D, 2 * D, (bf16*)dw, (bf16*)nullptr,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_rmsnorm_bwd");
    return AFK_OK;
}
