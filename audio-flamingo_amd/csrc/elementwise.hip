// HBM-bound glue kernels of the AF3 training path (gfx950): transposes, activations, RoPE, conv im2col,
// pooling, embedding gather/scatter, bias gradients, AdamW.  All loads/stores are 8- or 16-byte per lane,
// coalesced across the 64-lane wave; grids are capped at ~2048 blocks and grid-stride (guide G11/G13).
#include "common.h"
#include "../../include/afk.h"

namespace {

inline int ew_grid(int64_t n_items, int per_block) {
    int64_t g = afk_cdiv(n_items, per_block);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ batched 2-D transpose (bf16)
// out[b1][b2][c][r] = in[b1][b2][r][c], r < R, c < C; columns R..Rpad-1 of out are zero-filled so that the
// transposed operand can be fed to the NT GEMM with K padded to a multiple of 64.
// 64x64 tile through LDS: 16-byte row reads, 2-byte conflict-free column writes to LDS, 16-byte row writes out.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16* __restrict__ in0, bf16* __restrict__ out0, int R, int C,
                                                        int Rpad, int64_t ld_in, int64_t ld_out, int nb2,
                                                        int64_t bs1_in, int64_t bs2_in, int64_t bs1_out, int64_t bs2_out,
                                                        int tiles_x, int tiles_y, int64_t ntiles) {
    __shared__ bf16 tile[64][66];  // tile[c][r], +2 pad: column writes hit distinct banks
    const int t = threadIdx.x;
    // persistent over tiles: a capped grid ("thin" launch, one block per CU) can stay resident beside the 2 x 224-VGPR GEMM
    // workgroups of the other stream instead of flooding every CU at a kernel boundary
    for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        const int bx = (int)(tix % tiles_x);
        const int by = (int)((tix / tiles_x) % tiles_y);
        const int b = (int)(tix / ((int64_t)tiles_x * tiles_y));
        const int b1 = b / nb2, b2 = b - b1 * nb2;
        const bf16* in = in0 + b1 * bs1_in + b2 * bs2_in;
        bf16* out = out0 + b1 * bs1_out + b2 * bs2_out;
        const int r0 = by * 64, c0 = bx * 64;
        // load: 64 rows x 8 chunks of 8 bf16 = 512 chunks, 2 per thread
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = t + 256 * k;
            const int r = idx >> 3, ch = idx & 7;
            const int gr = r0 + r, gc = c0 + ch * 8;
            bf16x8 v;
            if (gr < R && gc + 7 < C) {
                v = *(const bf16x8*)(in + (int64_t)gr * ld_in + gc);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (gr < R && gc + e < C) ? in[(int64_t)gr * ld_in + gc + e] : (bf16)0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[ch * 8 + e][r] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = t + 256 * k;
            const int c = idx >> 3, ch = idx & 7;
            const int gc = c0 + c, gr = r0 + ch * 8;
            if (gc < C && gr < Rpad) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[c][ch * 8 + e];
                if (gr + 7 < Rpad) {
                    *(bf16x8*)(out + (int64_t)gc * ld_out + gr) = v;
                } else {
                    for (int e = 0; e < 8 && gr + e < Rpad; ++e) out[(int64_t)gc * ld_out + gr + e] = v[e];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ activations
// dx = dy * gelu'(pre)      (exact-erf GELU; oracle F.gelu, activations.py:70-89)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ pre,
                                                       bf16* __restrict__ dx, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 d = *(const bf16x8*)(dy + 8 * i);
        const bf16x8 p = *(const bf16x8*)(pre + 8 * i);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)d[e] * gelu_grad_f((float)p[e]));
        *(bf16x8*)(dx + 8 * i) = o;
    }
}

// y = gelu(x)  (standalone; conv stem uses the GEMM epilogue instead)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 p = *(const bf16x8*)(x + 8 * i);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)gelu_f((float)p[e]);
        *(bf16x8*)(y + 8 * i) = o;
    }
}


// h = bf16(bf16(silu(g)) * u)   gu = [rows, 2I] (gate | up), h = [rows, I]   (Qwen2MLP.forward, modeling_qwen2.py:46-48)
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ h, int64_t rows,
                                                           int I) {
    const int vpr = I >> 3;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        const bf16x8 g = *(const bf16x8*)(gu + r * 2 * I + c);
        const bf16x8 u = *(const bf16x8*)(gu + r * 2 * I + I + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)g[e];
            o[e] = (bf16)(rbf(gf * sigmoid_f(gf)) * (float)u[e]);
        }
        *(bf16x8*)(h + r * I + c) = o;
    }
}

// dgu[:, :I] = dh*u*silu'(g) ; dgu[:, I:] = dh*silu(g)
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const bf16* __restrict__ gu, const bf16* __restrict__ dh,
                                                           bf16* __restrict__ dgu, int64_t rows, int I) {
    const int vpr = I >> 3;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        const bf16x8 g = *(const bf16x8*)(gu + r * 2 * I + c);
        const bf16x8 u = *(const bf16x8*)(gu + r * 2 * I + I + c);
        const bf16x8 d = *(const bf16x8*)(dh + r * I + c);
        bf16x8 og, ou;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)g[e], uf = (float)u[e], df = (float)d[e];
            const float s = sigmoid_f(gf);
            const float silu = gf * s;
            og[e] = (bf16)(df * uf * (s * (1.f + gf * (1.f - s))));
            ou[e] = (bf16)(df * silu);
        }
        *(bf16x8*)(dgu + r * 2 * I + c) = og;
        *(bf16x8*)(dgu + r * 2 * I + I + c) = ou;
    }
}

// ------------------------------------------------------------------ RoPE (rotate-half), in place on the q|k columns
// apply_rotary_pos_emb (modeling_qwen2.py:112-135): x*cos + rotate_half(x)*sin with cos/sin rounded to bf16
// (:102) and every product/sum a bf16 tensor op.  buf = [rows, ld]; heads 0..nheads-1 of width D start at
// column 0 (q heads then k heads, contiguous).  pos[row] gives the position id (null -> row % S).
// sign=+1 forward, -1 backward (the transpose of a rotation is the rotation by -theta).
__global__ __launch_bounds__(256) void rope_kernel(bf16* __restrict__ buf, const bf16* __restrict__ cos_t,
                                                   const bf16* __restrict__ sin_t, const int* __restrict__ pos, int64_t rows,
                                                   int S, int ld, int nheads, int D, float sign) {
    const int half = D >> 1, vph = half >> 2;  // bf16x4 vectors per half-head
    const int64_t total = rows * nheads * vph;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vph);
        const int64_t t = i / vph;
        const int h = (int)(t % nheads);
        const int64_t r = t / nheads;
        const int p = pos ? pos[r] : (int)(r % S);
        bf16* x1p = buf + r * ld + h * D + 4 * v;
        bf16* x2p = x1p + half;
        const bf16x4 x1 = *(const bf16x4*)x1p, x2 = *(const bf16x4*)x2p;
        const bf16x4 c1 = *(const bf16x4*)(cos_t + (int64_t)p * D + 4 * v);
        const bf16x4 s1 = *(const bf16x4*)(sin_t + (int64_t)p * D + 4 * v);
        const bf16x4 c2 = *(const bf16x4*)(cos_t + (int64_t)p * D + half + 4 * v);
        const bf16x4 s2 = *(const bf16x4*)(sin_t + (int64_t)p * D + half + 4 * v);
        bf16x4 o1, o2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = (float)x1[e], b = (float)x2[e];
            o1[e] = (bf16)(rbf_strict(a * (float)c1[e]) + rbf_strict(-sign * b * (float)s1[e]));   // rbf_strict: a contraction-proof rounding point (common.h)
            o2[e] = (bf16)(rbf_strict(b * (float)c2[e]) + rbf_strict(sign * a * (float)s2[e]));
        }
        *(bf16x4*)x1p = o1;
        *(bf16x4*)x2p = o2;
    }
}

// ------------------------------------------------------------------ misc elementwise
__global__ __launch_bounds__(256) void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o,
                                                  int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 x = *(const bf16x8*)(a + 8 * i), y = *(const bf16x8*)(b + 8 * i);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)x[e] + (float)y[e]);
        *(bf16x8*)(o + 8 * i) = r;
    }
}

// y = (accumulate ? y : 0) + (*scale) * x     (scale is a DEVICE scalar: no host sync to apply an upstream loss gradient)
__global__ __launch_bounds__(256) void scale_add_kernel(const bf16* x, bf16* y, int64_t nvec, const float* __restrict__ scale,
                                                        int accumulate) {
    const float a = *scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 xv = *(const bf16x8*)(x + 8 * i);
        bf16x8 o;
        if (accumulate) {
            const bf16x8 yv = *(const bf16x8*)(y + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)yv[e] + a * (float)xv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(a * (float)xv[e]);
        }
        *(bf16x8*)(y + 8 * i) = o;
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ a, bf16* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = (bf16)a[i];
}

// out[r] (+)= sum_c in[r][c]  - bias gradient from the transposed grad (one wave per row, coalesced)
__global__ __launch_bounds__(256) void rowsum_kernel(const bf16* __restrict__ in, int64_t ld, int C, bf16* __restrict__ out,
                                                     int rows, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* p = in + (int64_t)row * ld;
    float s = 0.f;
    const int nv = C >> 3;
    for (int v = lane; v < nv; v += 64) {
        const bf16x8 t = *(const bf16x8*)(p + 8 * v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)t[e];
    }
    for (int c = nv * 8 + lane; c < C; c += 64) s += (float)p[c];
    s = wave_sum(s);
    if (lane == 0) out[row] = (bf16)(accumulate ? s + (float)out[row] : s);
}

// column sums (bias gradient straight from the row-major output gradient): out[c] (+)= sum_r in[r][c]
// stage 1: block = 64 columns x one row slice; 32 row-lanes x 8 column-lanes of bf16x8, LDS fold -> partial[slice][c]
// FUSED (round 4, afk_colsum_bf16_fused): no fold launch - the LAST row slice of a 64-column block to finish folds the slices itself (the hand-over of
// attention_decode.hip: partials as agent-scope write-through stores, drained before the block barrier, one counter bump per block, the last block reads
// the slices with agent-scope loads in slice order - the result does not depend on which block came last - and resets the counter).  The fold launch
// was 6.6 us x 158 bias gradients per step.
template <bool FUSED>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ in, int64_t ld, int64_t rows, int cols,
                                                             float* __restrict__ partial, int rows_per_slice, int* __restrict__ counters,
                                                             bf16* __restrict__ out, int accumulate) {
    __shared__ float red[32][65];
    __shared__ int last_flag;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cl * 8;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t r_end = min(rows, r_begin + rows_per_slice);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < cols) {
        for (int64_t r = r_begin + rl; r < r_end; r += 32) {
            const bf16x8 t = *(const bf16x8*)(in + r * ld + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
        float* dst = partial + (int64_t)blockIdx.y * cols + c;
        if (FUSED) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = s;
    }
    if (!FUSED) return;
    // explicit drain of this wave's agent-scope write-through stores (a workgroup-scope release fence emits no vmcnt wait on gfx950 - see attention_decode.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nslices = gridDim.y;
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (prev == nslices - 1);
        if (last_flag) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next call (a graph replay) starts from zero
    }
    __syncthreads();
    if (!last_flag || threadIdx.x >= 64 || c >= cols) return;
    float s = 0.f;
    for (int k0 = 0; k0 < nslices; k0 += 8) {   // eight slices per round trip, summed in slice order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(partial + (int64_t)min(k0 + u, nslices - 1) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nslices) s += v[u];
    }
    if (accumulate) s += (float)out[c];
    out[c] = (bf16)s;
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ partial, int nslices, int cols, bf16* __restrict__ out,
                                                          int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += partial[(int64_t)k * cols + c];
    if (accumulate) s += (float)out[c];
    out[c] = (bf16)s;
}

// ------------------------------------------------------------------ conv stem as GEMM: im2col / col2im
// conv1 (Conv1d(128->1280,k3,p1), modeling_audioflamingo3.py:328,380): x = [W, C, T] (f32 or bf16 log-mel,
// channel-major as the feature extractor emits it) -> col[(w,t)][kk*C + c] = x[w][c][t+kk-1] (0 outside).
template <typename T>
__global__ __launch_bounds__(256) void im2col_cmajor_kernel(const T* __restrict__ x, bf16* __restrict__ col, int W, int C,
                                                            int Tn) {
    __shared__ float tile[64][65];  // [c][t], t covers 62 outputs + halo
    const int w = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 62;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int c = idx >> 6, tt = idx & 63;
        const int t = t0 - 1 + tt, gc = c0 + c;
        float v = 0.f;
        if (t >= 0 && t < Tn && gc < C) v = (float)x[((int64_t)w * C + gc) * Tn + t];
        tile[c][tt] = v;
    }
    __syncthreads();
    // outputs: 62 time steps x 3 taps x 64 channels
    for (int idx = tid; idx < 62 * 3 * 64; idx += 256) {
        const int c = idx & 63;
        const int kk = (idx >> 6) % 3;
        const int tt = idx / 192;
        const int t = t0 + tt, gc = c0 + c;
        if (t < Tn && gc < C) col[((int64_t)w * Tn + t) * (3 * C) + kk * C + gc] = (bf16)tile[c][tt + kk];
    }
}

// conv2 (Conv1d(1280->1280,k3,s2,p1), :329,381): h = [W, Tin, C] time-major (conv1 GEMM output) ->
// col[(w,t')][kk*C + c] = h[w][2t'+kk-1][c]; pure 16-byte row copies.
__global__ __launch_bounds__(256) void im2col_tmajor_s2_kernel(const bf16* __restrict__ h, bf16* __restrict__ col, int W,
                                                               int Tin, int Tout, int C) {
    const int vpr = C >> 3;
    const int64_t total = (int64_t)W * Tout * 3 * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        int64_t t = i / vpr;
        const int kk = (int)(t % 3);
        t /= 3;
        const int to = (int)(t % Tout);
        const int w = (int)(t / Tout);
        const int ti = 2 * to + kk - 1;
        bf16x8 val;
        if (ti >= 0 && ti < Tin) {
            val = *(const bf16x8*)(h + ((int64_t)w * Tin + ti) * C + 8 * v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = (bf16)0.f;
        }
        *(bf16x8*)(col + ((int64_t)w * Tout + to) * (3 * C) + kk * C + 8 * v) = val;
    }
}

// col2im for conv2 dgrad (gather form, no atomics): dh[w][j][c] = sum over (t',kk) with 2t'+kk-1 == j of dcol[(w,t')][kk*C+c]
__global__ __launch_bounds__(256) void col2im_tmajor_s2_kernel(const bf16* __restrict__ dcol, bf16* __restrict__ dh, int W,
                                                               int Tin, int Tout, int C) {
    const int vpr = C >> 3;
    const int64_t total = (int64_t)W * Tin * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t t = i / vpr;
        const int j = (int)(t % Tin);
        const int w = (int)(t / Tin);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int num = j + 1 - kk;
            if (num >= 0 && (num & 1) == 0) {
                const int to = num >> 1;
                if (to < Tout) {
                    const bf16x8 d = *(const bf16x8*)(dcol + ((int64_t)w * Tout + to) * (3 * C) + kk * C + 8 * v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)d[e];
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
        *(bf16x8*)(dh + ((int64_t)w * Tin + j) * C + 8 * v) = o;
    }
}

// conv weight [Co][Ci][3] <-> GEMM operand [Co][3][Ci]   (dir=0: w->wperm, dir=1: wperm(+accumulate) -> w layout)
__global__ __launch_bounds__(256) void conv_w_permute_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int Co, int Ci,
                                                             int dir, int accumulate) {
    const int64_t total = (int64_t)Co * Ci * 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int kk = (int)(i % 3);
        const int64_t t = i / 3;
        const int ci = (int)(t % Ci);
        const int64_t co = t / Ci;
        const int64_t iw = (co * Ci + ci) * 3 + kk, ip = (co * 3 + kk) * Ci + ci;
        if (dir == 0) out[ip] = in[iw];
        else out[iw] = (bf16)(accumulate ? (float)out[iw] + (float)in[ip] : (float)in[ip]);
    }
}

// ------------------------------------------------------------------ AvgPool1d(2,2) over time (rows)  (:337,401-402)
// y[w][t][c] = bf16(0.5*(x[w][2t][c] + x[w][2t+1][c]));  bwd: dx[2t] = dx[2t+1] = bf16(0.5*dy[t])
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t out_rows,
                                                           int C) {
    const int vpr = C >> 3;
    const int64_t total = out_rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const bf16x8 a = *(const bf16x8*)(x + (2 * r) * C + 8 * v), b = *(const bf16x8*)(x + (2 * r + 1) * C + 8 * v);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(0.5f * ((float)a[e] + (float)b[e]));
        *(bf16x8*)(y + r * C + 8 * v) = o;
    }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int64_t out_rows,
                                                           int C) {
    const int vpr = C >> 3;
    const int64_t total = out_rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const bf16x8 d = *(const bf16x8*)(dy + r * C + 8 * v);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(0.5f * (float)d[e]);
        *(bf16x8*)(dx + (2 * r) * C + 8 * v) = o;
        *(bf16x8*)(dx + (2 * r + 1) * C + 8 * v) = o;
    }
}

// ------------------------------------------------------------------ embedding gather + <sound> scatter (:532-545)
// src[i] = rank of position i among placeholders (row-major order) if ids[i]==audio_id else -1; count -> *n_audio.
// Single block exclusive scan (the id matrix is B*S <= a few 10^4 entries).
__global__ __launch_bounds__(1024) void placeholder_scan_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t audio_id,
                                                                int* __restrict__ src, int* __restrict__ n_audio) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const int m = (i < n && ids[i] == audio_id) ? 1 : 0;
        int incl = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wv; ++k) woff += wsum[k];
        const int c = carry;
        if (i < n) src[i] = m ? (c + woff + incl - 1) : -1;
        __syncthreads();
        if (tid == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (tid == 0) *n_audio = carry;
}

// out[i] = src[i] >= 0 ? audio[src[i]] : embed[ids[i]]
__global__ __launch_bounds__(256) void embed_scatter_fwd_kernel(const int64_t* __restrict__ ids, const int* __restrict__ src,
                                                                const bf16* __restrict__ embed, const bf16* __restrict__ audio,
                                                                bf16* __restrict__ out, int64_t n, int H) {
    const int vpr = H >> 3;
    const int64_t total = n * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const int s = src ? src[r] : -1;
        const bf16* p = (s >= 0) ? audio + (int64_t)s * H : embed + ids[r] * H;
        *(bf16x8*)(out + r * H + 8 * v) = *(const bf16x8*)(p + 8 * v);
    }
}

// backward: placeholder rows are gathered into d_audio[src]; text rows are summed into d_embed[token id].
// Token ids repeat, so this is a segmented reduction: `perm` lists the rows sorted (stably) by token id (integer index plumbing done by
// the caller); the thread that sits on the FIRST row of an id's run sums the whole run in fp32, in row order, and rounds once into
// d_embed (+= what is already there: the caller clears the slice at the first backward of a step).  Fixed order, one rounding:
// bit-deterministic, and as accurate as torch's fp32-accumulating embedding backward (no atomics).
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(const int64_t* __restrict__ ids, const int* __restrict__ src,
                                                                const bf16* __restrict__ dout, bf16* __restrict__ d_embed,
                                                                bf16* __restrict__ d_audio, const int* __restrict__ perm, int64_t n, int H) {
    const int vpr = H >> 3;
    const int64_t total = n * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t j = i / vpr;
        const int64_t r = perm ? perm[j] : j;
        const int s = src ? src[r] : -1;
        if (s >= 0) {
            if (d_audio) *(bf16x8*)(d_audio + (int64_t)s * H + 8 * v) = *(const bf16x8*)(dout + r * H + 8 * v);
            continue;
        }
        if (!d_embed) continue;
        const int64_t id = ids[r];
        if (j > 0 && ids[perm[j - 1]] == id) continue;  // not the head of this id's run
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t k = j; k < n; ++k) {
            const int64_t rk = perm[k];
            if (ids[rk] != id) break;
            const bf16x8 d = *(const bf16x8*)(dout + rk * H + 8 * v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)d[e];
        }
        bf16* p = d_embed + id * H + 8 * v;
        const bf16x8 old = *(const bf16x8*)p;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)old[e] + acc[e]);
        *(bf16x8*)p = o;
    }
}

// One AdamW element update, shared by adamw_kernel and adamw_t_kernel.  Floating-point contraction is switched OFF inside: left to the
// optimiser, the two kernels fused different mul/add pairs into FMAs and their parameters differed in the last bit (measured) - the flat and
// the transposed-shadow launch of the same step must be interchangeable bit for bit.
__device__ __forceinline__ float adamw_elem(float& w, float& mm, float& vv, float gr, float lr, float b1, float b2, float eps, float wd, float bc1,
                                            float bc2_sqrt) {
#pragma clang fp contract(off)
    w = w * (1.f - lr * wd);
    mm = b1 * mm + (1.f - b1) * gr;
    vv = b2 * vv + ((1.f - b2) * gr) * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    w = w - (lr / bc1) * (mm / denom);
    return w;
}

// ------------------------------------------------------------------ AdamW (bf16 param + fp32 master/m/v)
// torch.optim.AdamW semantics: p *= 1-lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  grad_scale multiplies g (DP averaging / loss scaling).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16* __restrict__ g, bf16* __restrict__ p, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale, const int* __restrict__ gate, const float* __restrict__ hyper) {
    if (gate != nullptr && *gate == 0) return;  // data parallel: no rank produced a gradient for this bucket this step
    if (hyper != nullptr) {  // step-dependent scalars from device memory: a captured HIP graph replays with fresh values
        lr = hyper[0];
        bc1 = hyper[1];
        bc2_sqrt = hyper[2];
        grad_scale *= hyper[3];  // global-norm clip coefficient of this step (afk_clip_coef), 1 when clipping is off
    }
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        f32x4 w = *(const f32x4*)(master + 4 * i), mm = *(const f32x4*)(m + 4 * i), vv = *(const f32x4*)(v + 4 * i);
        const bf16x4 gg = *(const bf16x4*)(g + 4 * i);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float we = w[e], me = mm[e], ve = vv[e];
            o[e] = (bf16)adamw_elem(we, me, ve, (float)gg[e] * grad_scale, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
            w[e] = we, mm[e] = me, vv[e] = ve;
        }
        *(f32x4*)(master + 4 * i) = w;
        *(f32x4*)(m + 4 * i) = mm;
        *(f32x4*)(v + 4 * i) = vv;
        *(bf16x4*)(p + 4 * i) = o;
    }
    // tail (n % 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        float w = master[i], mm = m[i], vv = v[i];
        adamw_elem(w, mm, vv, (float)g[i] * grad_scale, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        master[i] = w;
        m[i] = mm;
        v[i] = vv;
        p[i] = (bf16)w;
    }
}


// AdamW on ONE 2-D weight [N, K] that also writes the K-major copy of the updated bf16 weight (the W^T shadow the dgrad GEMM reads,
// arena.py): the optimizer already holds every new value in registers, so the shadow costs its 2 B/param of writes and nothing else -
// the separate afk_transpose_bf16 pass per weight (2 B read + 2 B write per parameter, 15.4 GB of reads and ~245 launches per AF3-7B step)
// disappears.  64x64 tiles: a row of a tile is 256 contiguous bytes of each fp32 state (8 lanes x 32 B) and 128 B of bf16 gradient /
// parameter; the transposed tile goes through LDS exactly as in transpose_kernel (2-byte column writes into a +2-padded tile, 16-byte row
// reads) and leaves as 128-byte rows of the shadow.  Same arithmetic, element for element, as adamw_kernel: parameters are bit-identical to
// the unfused step and shadow == transpose(param) (tests/test_ops_gpu.py::test_adamw_fused_transposed_shadow).  N % 64 == 0 and K % 64 == 0.
// <= 64 VGPRs (8 waves/SIMD): a thin launch of this kernel must fit beside the two 224-VGPR waves a GEMM workgroup keeps on every SIMD (the first
// build needed 72 and could not co-reside: +17 ms per step, measured)
__global__ __launch_bounds__(256) void adamw_t_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                      const bf16* __restrict__ g, bf16* __restrict__ p, bf16* __restrict__ shadow, int N, int K,
                                                      int64_t ld_shadow, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                      float bc2_sqrt, float grad_scale, const int* __restrict__ gate, const float* __restrict__ hyper) {
    if (gate != nullptr && *gate == 0) return;
    if (hyper != nullptr) {
        lr = hyper[0];
        bc1 = hyper[1];
        bc2_sqrt = hyper[2];
        grad_scale *= hyper[3];
    }
    __shared__ bf16 tile[64][66];  // tile[k][n]
    const int t = threadIdx.x;
    const int tiles_x = K >> 6;
    const int64_t ntiles = (int64_t)tiles_x * (N >> 6);
    for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        const int n0 = (int)(tix / tiles_x) * 64, k0 = (int)(tix % tiles_x) * 64;
        // 16 lanes x 4 elements per row, 16 rows per pass, 4 passes: the same 4-element granularity (and register footprint) as adamw_kernel
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int r = (t >> 4) + 16 * j, c4 = (t & 15) * 4;
            const int64_t off = (int64_t)(n0 + r) * K + k0 + c4;
            f32x4 w = *(const f32x4*)(master + off), mm = *(const f32x4*)(m + off), vv = *(const f32x4*)(v + off);
            const bf16x4 gg = *(const bf16x4*)(g + off);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float we = w[e], me = mm[e], ve = vv[e];
                o[e] = (bf16)adamw_elem(we, me, ve, (float)gg[e] * grad_scale, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
                w[e] = we, mm[e] = me, vv[e] = ve;
            }
            *(f32x4*)(master + off) = w;
            *(f32x4*)(m + off) = mm;
            *(f32x4*)(v + off) = vv;
            *(bf16x4*)(p + off) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[c4 + e][r] = o[e];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = t + 256 * j;
            const int c = idx >> 3, ch = idx & 7;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tile[c][ch * 8 + e];
            *(bf16x8*)(shadow + (int64_t)(k0 + c) * ld_shadow + n0 + ch * 8) = o;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ Flamingo glue (BASELINE config 4; stand-in oracle: Idefics)
// ReLU (IdeficsMLP of the Perceiver resampler, perceiver.py:171-187)
__global__ __launch_bounds__(256) void relu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(x + 8 * i);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e] > 0.f ? v[e] : (bf16)0.f;
        *(bf16x8*)(y + 8 * i) = o;
    }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dx,
                                                       int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 d = *(const bf16x8*)(dy + 8 * i), v = *(const bf16x8*)(y + 8 * i);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e] > 0.f ? d[e] : (bf16)0.f;
        *(bf16x8*)(dx + 8 * i) = o;
    }
}
// tanh-gated residual (IdeficsGatedCrossAttentionLayer.forward, modeling_idefics.py:792-793,800):
//   y = x + tanh(alpha) * (gate[row] ? h : 0)      alpha: vector [D] or scalar [1]; gate: int32 [rows] or null
__global__ __launch_bounds__(256) void gate_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ h, const bf16* __restrict__ alpha,
                                                       int alpha_vec, const int* __restrict__ gate, bf16* __restrict__ y, int64_t rows,
                                                       int D) {
    const int vpr = D >> 3;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const bf16x8 xv = *(const bf16x8*)(x + r * D + 8 * v), hv = *(const bf16x8*)(h + r * D + 8 * v);
        const bool on = gate ? gate[r] != 0 : true;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = rbf(tanhf((float)alpha[alpha_vec ? 8 * v + e : 0]));  // tanh of a bf16 parameter is a bf16 tensor in the oracle
            const float g = on ? rbf_strict(t * (float)hv[e]) : 0.f;   // rbf_strict: the product's rounding must survive fp-contract (common.h)
            o[e] = (bf16)((float)xv[e] + g);
        }
        *(bf16x8*)(y + r * D + 8 * v) = o;
    }
}
// backward: dh = dy * tanh(alpha) * gate   (d alpha comes from gate_prod_colsum_kernel)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h, const bf16* __restrict__ alpha,
                                                       int alpha_vec, const int* __restrict__ gate, bf16* __restrict__ dh,
                                                       int64_t rows, int D) {
    const int vpr = D >> 3;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const bf16x8 dv = *(const bf16x8*)(dy + r * D + 8 * v), hv = *(const bf16x8*)(h + r * D + 8 * v);
        const bool on = gate ? gate[r] != 0 : true;
        bf16x8 o;
        (void)hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = rbf(tanhf((float)alpha[alpha_vec ? 8 * v + e : 0]));
            o[e] = (bf16)(on ? (float)dv[e] * t : 0.f);
        }
        *(bf16x8*)(dh + r * D + 8 * v) = o;
    }
}
// partial[slice][c] = sum over the slice's rows of dy*h*gate, accumulated in fp32 (the alpha gradient is a heavily cancelling
// sum: rounding the products to bf16 first costs ~25 % relative error on the scalar-alpha form)
__global__ __launch_bounds__(256) void gate_prod_colsum_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h,
                                                               const int* __restrict__ gate, int64_t rows, int cols,
                                                               float* __restrict__ partial, int rows_per_slice) {
    __shared__ float red[32][65];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cl * 8;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t r_end = min(rows, r_begin + rows_per_slice);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < cols) {
        for (int64_t r = r_begin + rl; r < r_end; r += 32) {
            if (gate && gate[r] == 0) continue;
            const bf16x8 a = *(const bf16x8*)(dy + r * cols + c0), b = *(const bf16x8*)(h + r * cols + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)a[e] * (float)b[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < cols) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
            partial[(int64_t)blockIdx.y * cols + c] = s;
        }
    }
}
// d alpha from the column sums cs[D] (fp32): vector: out[d] (+)= (1 - tanh^2(alpha_d)) cs[d]; scalar: out[0] (+)= (1 - tanh^2(alpha)) sum_d cs[d]
__global__ __launch_bounds__(1024) void gate_alpha_grad_kernel(const float* __restrict__ cs, const bf16* __restrict__ alpha, int alpha_vec,
                                                               bf16* __restrict__ out, int D, int accumulate) {
    __shared__ float scratch[16];
    if (alpha_vec) {
        for (int d = threadIdx.x; d < D; d += 1024) {
            const float t = tanhf((float)alpha[d]);
            const float g = (1.f - t * t) * cs[d];
            out[d] = (bf16)(accumulate ? (float)out[d] + g : g);
        }
    } else {
        float s = 0.f;
        for (int d = threadIdx.x; d < D; d += 1024) s += cs[d];
        s = block_sum<16>(s, scratch);
        if (threadIdx.x == 0) {
            const float t = tanhf((float)alpha[0]);
            const float g = (1.f - t * t) * s;
            out[0] = (bf16)(accumulate ? (float)out[0] + g : g);
        }
    }
}
// fp32 column sums (deterministic two-stage) used by the alpha gradient
__global__ __launch_bounds__(256) void colsum_fold_f32_kernel(const float* __restrict__ partial, int nslices, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += partial[(int64_t)k * cols + c];
    out[c] = s;
}

}  // namespace

// ==================================================================== C ABI
#define ST ((hipStream_t)stream)

extern "C" int afk_transpose_bf16(const void* in, void* out, int R, int C, int Rpad, int64_t ld_in, int64_t ld_out, int nb1,
                                  int nb2, int64_t bs1_in, int64_t bs2_in, int64_t bs1_out, int64_t bs2_out, int max_blocks,
                                  void* stream) {
    AFK_REQUIRE(in && out && R > 0 && C > 0 && Rpad >= R && nb1 > 0 && nb2 > 0, "afk_transpose_bf16: bad args");
    AFK_REQUIRE(ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                    bs1_in % 8 == 0 && bs2_in % 8 == 0 && bs1_out % 8 == 0 && bs2_out % 8 == 0,
                "afk_transpose_bf16: 16-byte alignment required");
    const int tiles_x = (int)afk_cdiv(C, 64), tiles_y = (int)afk_cdiv(Rpad, 64);
    const int64_t ntiles = (int64_t)tiles_x * tiles_y * nb1 * nb2;
    int64_t grid = ntiles;
    if (grid > 65536) grid = 65536;
    if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)grid), dim3(256), 0, ST, (const bf16*)in, (bf16*)out, R, C, Rpad, ld_in, ld_out,
                       nb2, bs1_in, bs2_in, bs1_out, bs2_out, tiles_x, tiles_y, ntiles);
    AFK_LAUNCH_CHECK("afk_transpose_bf16");
    return AFK_OK;
}

extern "C" int afk_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    AFK_REQUIRE(x && y && n % 8 == 0, "afk_gelu_fwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, n / 8);
    AFK_LAUNCH_CHECK("afk_gelu_fwd");
    return AFK_OK;
}

extern "C" int afk_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, void* stream) {
    AFK_REQUIRE(dy && pre && dx && n % 8 == 0, "afk_gelu_bwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)pre,
                       (bf16*)dx, n / 8);
    AFK_LAUNCH_CHECK("afk_gelu_bwd");
    return AFK_OK;
}

extern "C" int afk_silu_mul_fwd(const void* gu, void* h, int64_t rows, int I, void* stream) {
    AFK_REQUIRE(gu && h && I % 8 == 0 && rows > 0, "afk_silu_mul_fwd: bad args");
    hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(ew_grid(rows * (I / 8), 256)), dim3(256), 0, ST, (const bf16*)gu, (bf16*)h,
                       rows, I);
    AFK_LAUNCH_CHECK("afk_silu_mul_fwd");
    return AFK_OK;
}

extern "C" int afk_silu_mul_bwd(const void* gu, const void* dh, void* dgu, int64_t rows, int I, void* stream) {
    AFK_REQUIRE(gu && dh && dgu && I % 8 == 0 && rows > 0, "afk_silu_mul_bwd: bad args");
    hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(ew_grid(rows * (I / 8), 256)), dim3(256), 0, ST, (const bf16*)gu,
                       (const bf16*)dh, (bf16*)dgu, rows, I);
    AFK_LAUNCH_CHECK("afk_silu_mul_bwd");
    return AFK_OK;
}

extern "C" int afk_rope_inplace(void* buf, const void* cos_t, const void* sin_t, const int* pos, int64_t rows, int S, int ld,
                                int nheads, int D, int backward, void* stream) {
    AFK_REQUIRE(buf && cos_t && sin_t && rows > 0 && D % 8 == 0 && ld % 4 == 0, "afk_rope_inplace: bad args");
    hipLaunchKernelGGL(rope_kernel, dim3(ew_grid(rows * nheads * (D / 8), 256)), dim3(256), 0, ST, (bf16*)buf,
                       (const bf16*)cos_t, (const bf16*)sin_t, pos, rows, S, ld, nheads, D, backward ? -1.f : 1.f);
    AFK_LAUNCH_CHECK("afk_rope_inplace");
    return AFK_OK;
}

extern "C" int afk_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    AFK_REQUIRE(a && b && out && n % 8 == 0, "afk_add_bf16: n must be a multiple of 8");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)a, (const bf16*)b, (bf16*)out,
                       n / 8);
    AFK_LAUNCH_CHECK("afk_add_bf16");
    return AFK_OK;
}

extern "C" int afk_scale_add_bf16(const void* x, void* y, int64_t n, const float* scale_dev, int accumulate, void* stream) {
    AFK_REQUIRE(x && y && scale_dev && n % 8 == 0, "afk_scale_add_bf16: n must be a multiple of 8");
    hipLaunchKernelGGL(scale_add_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, n / 8, scale_dev,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_scale_add_bf16");
    return AFK_OK;
}

extern "C" int afk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
    AFK_REQUIRE(in && out && n > 0, "afk_cast_f32_bf16: bad args");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ST, in, (bf16*)out, n);
    AFK_LAUNCH_CHECK("afk_cast_f32_bf16");
    return AFK_OK;
}

extern "C" int afk_rowsum_bf16(const void* in, int64_t ld, int C, void* out, int rows, int accumulate, void* stream) {
    AFK_REQUIRE(in && out && rows > 0 && C > 0 && ld % 8 == 0, "afk_rowsum_bf16: bad args");
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)afk_cdiv(rows, 4)), dim3(256), 0, ST, (const bf16*)in, ld, C, (bf16*)out,
                       rows, accumulate);
    AFK_LAUNCH_CHECK("afk_rowsum_bf16");
    return AFK_OK;
}

extern "C" int afk_colsum_slices(int64_t rows) {
    int64_t s = afk_cdiv(rows, 512);
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

static int colsum_impl(const void* in, int64_t ld, int64_t rows, int cols, void* out, int accumulate, float* workspace, int* counters, void* stream) {
    AFK_REQUIRE(in && out && workspace && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0, "afk_colsum_bf16: bad args");
    const int ns = afk_colsum_slices(rows);
    const int rps = (int)afk_cdiv(rows, ns);
    const dim3 grid((unsigned)afk_cdiv(cols, 64), (unsigned)ns);
    if (counters) {
        hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(256), 0, ST, (const bf16*)in, ld, rows, cols, workspace, rps, counters, (bf16*)out, accumulate);
    } else {
        hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(256), 0, ST, (const bf16*)in, ld, rows, cols, workspace, rps, (int*)nullptr, (bf16*)out, accumulate);
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)afk_cdiv(cols, 256)), dim3(256), 0, ST, workspace, ns, cols, (bf16*)out, accumulate);
    }
    AFK_LAUNCH_CHECK("afk_colsum_bf16");
    return AFK_OK;
}

extern "C" int afk_colsum_bf16(const void* in, int64_t ld, int64_t rows, int cols, void* out, int accumulate, float* workspace, void* stream) {
    return colsum_impl(in, ld, rows, cols, out, accumulate, workspace, nullptr, stream);
}

// one launch: counters = ceil(cols / 64) ints that read ZERO before the first call (the kernel leaves them at zero); one counter array per stream
extern "C" int afk_colsum_bf16_fused(const void* in, int64_t ld, int64_t rows, int cols, void* out, int accumulate, float* workspace, int* counters,
                                     void* stream) {
    AFK_REQUIRE(counters, "afk_colsum_bf16_fused: null counters");
    return colsum_impl(in, ld, rows, cols, out, accumulate, workspace, counters, stream);
}

extern "C" int afk_im2col_conv1(const void* x, int x_is_f32, void* col, int W, int C, int T, void* stream) {
    AFK_REQUIRE(x && col && W > 0 && C > 0 && T > 0, "afk_im2col_conv1: bad args");
    dim3 grid((unsigned)afk_cdiv(T, 62), (unsigned)afk_cdiv(C, 64), (unsigned)W);
    if (x_is_f32)
        hipLaunchKernelGGL(im2col_cmajor_kernel<float>, grid, dim3(256), 0, ST, (const float*)x, (bf16*)col, W, C, T);
    else
        hipLaunchKernelGGL(im2col_cmajor_kernel<bf16>, grid, dim3(256), 0, ST, (const bf16*)x, (bf16*)col, W, C, T);
    AFK_LAUNCH_CHECK("afk_im2col_conv1");
    return AFK_OK;
}

extern "C" int afk_im2col_conv2(const void* h, void* col, int W, int Tin, int Tout, int C, void* stream) {
    AFK_REQUIRE(h && col && C % 8 == 0, "afk_im2col_conv2: bad args");
    hipLaunchKernelGGL(im2col_tmajor_s2_kernel, dim3(ew_grid((int64_t)W * Tout * 3 * (C / 8), 256)), dim3(256), 0, ST,
                       (const bf16*)h, (bf16*)col, W, Tin, Tout, C);
    AFK_LAUNCH_CHECK("afk_im2col_conv2");
    return AFK_OK;
}

extern "C" int afk_col2im_conv2(const void* dcol, void* dh, int W, int Tin, int Tout, int C, void* stream) {
    AFK_REQUIRE(dcol && dh && C % 8 == 0, "afk_col2im_conv2: bad args");
    hipLaunchKernelGGL(col2im_tmajor_s2_kernel, dim3(ew_grid((int64_t)W * Tin * (C / 8), 256)), dim3(256), 0, ST,
                       (const bf16*)dcol, (bf16*)dh, W, Tin, Tout, C);
    AFK_LAUNCH_CHECK("afk_col2im_conv2");
    return AFK_OK;
}

extern "C" int afk_conv_weight_permute(const void* in, void* out, int Co, int Ci, int dir, int accumulate, void* stream) {
    AFK_REQUIRE(in && out && Co > 0 && Ci > 0, "afk_conv_weight_permute: bad args");
    hipLaunchKernelGGL(conv_w_permute_kernel, dim3(ew_grid((int64_t)Co * Ci * 3, 256)), dim3(256), 0, ST, (const bf16*)in,
                       (bf16*)out, Co, Ci, dir, accumulate);
    AFK_LAUNCH_CHECK("afk_conv_weight_permute");
    return AFK_OK;
}

extern "C" int afk_avgpool2_fwd(const void* x, void* y, int64_t out_rows, int C, void* stream) {
    AFK_REQUIRE(x && y && C % 8 == 0 && out_rows > 0, "afk_avgpool2_fwd: bad args");
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(ew_grid(out_rows * (C / 8), 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y,
                       out_rows, C);
    AFK_LAUNCH_CHECK("afk_avgpool2_fwd");
    return AFK_OK;
}

extern "C" int afk_avgpool2_bwd(const void* dy, void* dx, int64_t out_rows, int C, void* stream) {
    AFK_REQUIRE(dy && dx && C % 8 == 0 && out_rows > 0, "afk_avgpool2_bwd: bad args");
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(ew_grid(out_rows * (C / 8), 256)), dim3(256), 0, ST, (const bf16*)dy, (bf16*)dx,
                       out_rows, C);
    AFK_LAUNCH_CHECK("afk_avgpool2_bwd");
    return AFK_OK;
}

extern "C" int afk_placeholder_scan(const int64_t* ids, int64_t n, int64_t audio_id, int* src, int* n_audio, void* stream) {
    AFK_REQUIRE(ids && src && n_audio && n > 0, "afk_placeholder_scan: bad args");
    hipLaunchKernelGGL(placeholder_scan_kernel, dim3(1), dim3(1024), 0, ST, ids, n, audio_id, src, n_audio);
    AFK_LAUNCH_CHECK("afk_placeholder_scan");
    return AFK_OK;
}

extern "C" int afk_embed_scatter_fwd(const int64_t* ids, const int* src, const void* embed, const void* audio, void* out,
                                     int64_t n, int H, void* stream) {
    AFK_REQUIRE(ids && embed && out && n > 0 && H % 8 == 0, "afk_embed_scatter_fwd: bad args");
    AFK_REQUIRE(!src || audio, "afk_embed_scatter_fwd: src without audio rows");
    hipLaunchKernelGGL(embed_scatter_fwd_kernel, dim3(ew_grid(n * (H / 8), 256)), dim3(256), 0, ST, ids, src,
                       (const bf16*)embed, (const bf16*)audio, (bf16*)out, n, H);
    AFK_LAUNCH_CHECK("afk_embed_scatter_fwd");
    return AFK_OK;
}

extern "C" int afk_embed_scatter_bwd(const int64_t* ids, const int* src, const void* dout, void* d_embed, void* d_audio,
                                     const int* perm, int64_t n, int H, void* stream) {
    AFK_REQUIRE(ids && dout && n > 0 && H % 8 == 0, "afk_embed_scatter_bwd: bad args");
    AFK_REQUIRE(perm || !d_embed, "afk_embed_scatter_bwd: d_embed needs perm (rows sorted by token id)");
    hipLaunchKernelGGL(embed_scatter_bwd_kernel, dim3(ew_grid(n * (H / 8), 256)), dim3(256), 0, ST, ids, src,
                       (const bf16*)dout, (bf16*)d_embed, (bf16*)d_audio, perm, n, H);
    AFK_LAUNCH_CHECK("afk_embed_scatter_bwd");
    return AFK_OK;
}

extern "C" int afk_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks, const int* gate,
                              const float* hyper, void* stream) {
    AFK_REQUIRE(master && m && v && grad && param && n > 0 && step >= 1, "afk_adamw_step: bad args");
    AFK_REQUIRE(((uintptr_t)master % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                    ((uintptr_t)grad % 8 == 0) && ((uintptr_t)param % 8 == 0),
                "afk_adamw_step: misaligned buffer");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    // max_blocks > 0 caps the grid (grid-stride covers the rest): a THIN launch (one block per CU, 1 wave/SIMD, 56 VGPRs)
    // can stay resident beside a 2 x 224-VGPR GEMM workgroup, so the optimizer's HBM stream overlaps MFMA-bound backward
    // kernels; a full-size grid would instead fill every SIMD and lock the GEMM out until it drains.
    int grid = ew_grid(afk_cdiv(n, 4), 256);
    if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, ST, master, m, v, (const bf16*)grad,
                       (bf16*)param, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, gate, hyper);
    AFK_LAUNCH_CHECK("afk_adamw_step");
    return AFK_OK;
}


// AdamW on one 2-D weight [N, K] (row-major views of the arena) that also writes its K-major copy `shadow` [K, ld_shadow] (ld_shadow >= N)
extern "C" int afk_adamw_step_t(float* master, float* m, float* v, const void* grad, void* param, void* shadow, int N, int K, int64_t ld_shadow,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks,
                                const int* gate, const float* hyper, void* stream) {
    AFK_REQUIRE(master && m && v && grad && param && shadow && step >= 1, "afk_adamw_step_t: bad args");
    AFK_REQUIRE(N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0 && ld_shadow >= N && ld_shadow % 8 == 0, "afk_adamw_step_t: N=%d, K=%d must be multiples of 64", N, K);
    AFK_REQUIRE(((uintptr_t)master % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)grad % 16 == 0) &&
                    ((uintptr_t)param % 16 == 0) && ((uintptr_t)shadow % 16 == 0),
                "afk_adamw_step_t: misaligned buffer");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    int64_t grid = (int64_t)(N / 64) * (K / 64);
    if (grid > 16384) grid = 16384;
    if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
    hipLaunchKernelGGL(adamw_t_kernel, dim3((unsigned)grid), dim3(256), 0, ST, master, m, v, (const bf16*)grad, (bf16*)param, (bf16*)shadow, N, K,
                       ld_shadow, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, gate, hyper);
    AFK_LAUNCH_CHECK("afk_adamw_step_t");
    return AFK_OK;
}

namespace {
// ---- global gradient norm (torch.nn.utils.clip_grad_norm_, TORCH/nn/utils/clip_grad.py: total_norm = ||g||_2 over every gradient,
// clip_coef = clamp(max_norm / (total_norm + 1e-6), max = 1)).  The gradient arena is flat, so the norm is ONE streaming pass (2 B/param)
// instead of a foreach over ~700 tensors, and the coefficient is applied inside the AdamW launch (hyper[3]) instead of a read-modify-write
// pass over the gradients.  Two stages with a fixed grid and a fixed fold order: bit-deterministic.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const bf16* __restrict__ x, int64_t n, float* __restrict__ ws) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(x + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)v[e] * (float)v[e];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const float t = (float)x[(nv << 3) + threadIdx.x];
        acc += t * t;
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ acc, const int* __restrict__ gate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += ws[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0 && (gate == nullptr || *gate != 0)) acc[0] += s;
}
__global__ void clip_coef_kernel(const float* sumsq, int n, float scale, float max_norm, float* coef, float* norm_out) {
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < n; ++i) tot += sumsq[i];  // per-bucket slots, always folded in index order: the schedule that filled them is irrelevant
        const float norm = sqrtf(tot) * scale;
        const float c = max_norm / (norm + 1e-6f);
        coef[0] = c < 1.f ? c : 1.f;
        if (norm_out) norm_out[0] = norm;
    }
}
}  // namespace

// acc[0] += sum x[i]^2 (fp32), skipped when *gate == 0.  workspace: afk_sumsq_workspace_floats() floats, private to this call until it ran.
extern "C" int afk_sumsq_workspace_floats(void) { return SUMSQ_BLOCKS; }
extern "C" int afk_sumsq_bf16(const void* x, int64_t n, float* acc, const int* gate, float* workspace, void* stream) {
    AFK_REQUIRE(x && acc && workspace && n > 0, "afk_sumsq_bf16: bad args");
    AFK_REQUIRE(((uintptr_t)x & 15) == 0, "afk_sumsq_bf16: x must be 16-byte aligned");
    int grid = (int)afk_cdiv(afk_cdiv(n, 8), 256);
    if (grid > SUMSQ_BLOCKS) grid = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, ST, (const bf16*)x, n, workspace);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, ST, workspace, grid, acc, gate);
    AFK_LAUNCH_CHECK("afk_sumsq_bf16");
    return AFK_OK;
}
// coef[0] = min(1, max_norm / (sqrt(sum sumsq[0..n)) * scale + 1e-6)); norm_out[0] (may be NULL) = sqrt(sum sumsq) * scale
extern "C" int afk_clip_coef(const float* sumsq, int n, float scale, float max_norm, float* coef, float* norm_out, void* stream) {
    AFK_REQUIRE(sumsq && coef && n >= 1 && max_norm > 0.f && scale > 0.f, "afk_clip_coef: bad args");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, ST, sumsq, n, scale, max_norm, coef, norm_out);
    AFK_LAUNCH_CHECK("afk_clip_coef");
    return AFK_OK;
}

namespace {
__global__ void set_f32_kernel(float* dst, int n, float a, float b, float c, float d) {
    const float v[4] = {a, b, c, d};
    if (threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}
}  // namespace

// dst[0..n) = (a, b, c, d)[0..n): step-dependent scalars (learning rate, Adam bias corrections) written by a kernel whose arguments
// travel with the launch - no pinned staging buffer the host could overwrite while an earlier copy is still queued
extern "C" int afk_set_f32(float* dst, int n, float a, float b, float c, float d, void* stream) {
    AFK_REQUIRE(dst && n >= 1 && n <= 4, "afk_set_f32: 1..4 values");
    hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(64), 0, ST, dst, n, a, b, c, d);
    AFK_LAUNCH_CHECK("afk_set_f32");
    return AFK_OK;
}

namespace {
// Music Flamingo rotary time embedding (apply_rotary_time_emb, modeling_musicflamingo.py:187-204): interleaved-pair rotation of the
// first R features of every encoder output row by per-(row, feature) angles; the other E-R features pass through.  The reference
// does the arithmetic in fp64 on fp32 cos/sin tables and rounds to the activation dtype; fp32 here (the bf16 rounding of the result
// is 2^-9, fp32 products are exact to 2^-24).  backward = the transposed rotation applied to the gradient.
__global__ __launch_bounds__(256) void rotary_time_kernel(const bf16* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                          bf16* __restrict__ y, int64_t rows, int E, int R, int backward) {
    const int half = E >> 1;
    const int64_t total = rows * half;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / half;
        const int c = (int)(t % half) * 2;
        const bf16x2 v = *(const bf16x2*)(x + r * E + c);
        bf16x2 o = v;
        if (c < R) {
            const float x0 = (float)v[0], x1 = (float)v[1];
            const float c0 = cs[r * R + c], c1 = cs[r * R + c + 1], s0 = sn[r * R + c], s1 = sn[r * R + c + 1];
            if (!backward) {
                o[0] = (bf16)(x0 * c0 - x1 * s0);
                o[1] = (bf16)(x1 * c1 + x0 * s1);
            } else {
                o[0] = (bf16)(x0 * c0 + x1 * s1);
                o[1] = (bf16)(x1 * c1 - x0 * s0);
            }
        }
        *(bf16x2*)(y + r * E + c) = o;
    }
}
}  // namespace

extern "C" int afk_rotary_time(const void* x, const float* cos_t, const float* sin_t, void* y, int64_t rows, int E, int R, int backward,
                               void* stream) {
    AFK_REQUIRE(x && cos_t && sin_t && y && rows > 0 && E > 0 && E % 2 == 0 && R >= 0 && R <= E && R % 2 == 0, "afk_rotary_time: bad args");
    hipLaunchKernelGGL(rotary_time_kernel, dim3(ew_grid(rows * (E / 2), 256)), dim3(256), 0, ST, (const bf16*)x, cos_t, sin_t, (bf16*)y, rows,
                       E, R, backward);
    AFK_LAUNCH_CHECK("afk_rotary_time");
    return AFK_OK;
}

namespace {
// KV-cache append (decode path): rows r = b*n + i of the fused projection output hold the new token's K (already rotated) and V;
// K goes to Kc[b][start+i][:] (row-major), V to Vt[b][h][d][start+i] (stored transposed: the layout the interval attention kernels
// read).  `start` comes from device memory when start_dev != null, so a captured HIP graph of one decode step can be replayed
// while the position advances on the device.
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16* __restrict__ qkv, int64_t ld, int k_col0, bf16* __restrict__ Kc,
                                                        int64_t kc_bs, bf16* __restrict__ Vt, int64_t vt_bs, int spad,
                                                        const int* __restrict__ start_dev, int start_host, int B, int n, int Hkv, int D) {
    const int nk = Hkv * D;
    const int start = start_dev ? *start_dev : start_host;
    const int64_t total = (int64_t)B * n * nk;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % nk);
        const int64_t r = t / nk;
        const int i = (int)(r % n), b = (int)(r / n);
        const bf16* row = qkv + r * ld + k_col0;
        Kc[b * kc_bs + (int64_t)(start + i) * nk + c] = row[c];
        Vt[b * vt_bs + (int64_t)c * spad + start + i] = row[nk + c];  // c = h*D + d
    }
}
}  // namespace

extern "C" int afk_kv_cache_append(const void* qkv, int64_t ld, int k_col0, void* kcache, int64_t kc_bs, void* vtcache, int64_t vt_bs,
                                   int spad, const int* start_dev, int start_host, int B, int n, int Hkv, int D, void* stream) {
    AFK_REQUIRE(qkv && kcache && vtcache && B > 0 && n > 0 && Hkv > 0 && D > 0 && spad > 0, "afk_kv_cache_append: bad args");
    const int64_t total = (int64_t)B * n * Hkv * D;
    hipLaunchKernelGGL(kv_append_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ST, (const bf16*)qkv, ld, k_col0, (bf16*)kcache, kc_bs,
                       (bf16*)vtcache, vt_bs, spad, start_dev, start_host, B, n, Hkv, D);
    AFK_LAUNCH_CHECK("afk_kv_cache_append");
    return AFK_OK;
}

extern "C" int afk_relu_fwd(const void* x, void* y, int64_t n, void* stream) {
    AFK_REQUIRE(x && y && n % 8 == 0, "afk_relu_fwd: n must be a multiple of 8");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, n / 8);
    AFK_LAUNCH_CHECK("afk_relu_fwd");
    return AFK_OK;
}

extern "C" int afk_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, void* stream) {
    AFK_REQUIRE(dy && y && dx && n % 8 == 0, "afk_relu_bwd: n must be a multiple of 8");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n / 8);
    AFK_LAUNCH_CHECK("afk_relu_bwd");
    return AFK_OK;
}

extern "C" int afk_gate_fwd(const void* x, const void* h, const void* alpha, int alpha_is_vector, const int* gate, void* y, int64_t rows,
                            int D, void* stream) {
    AFK_REQUIRE(x && h && alpha && y && rows > 0 && D % 8 == 0, "afk_gate_fwd: bad args");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(ew_grid(rows * (D / 8), 256)), dim3(256), 0, ST, (const bf16*)x, (const bf16*)h,
                       (const bf16*)alpha, alpha_is_vector, gate, (bf16*)y, rows, D);
    AFK_LAUNCH_CHECK("afk_gate_fwd");
    return AFK_OK;
}

// workspace `fws` = (afk_colsum_slices(rows) + 1) * D floats
extern "C" int afk_gate_bwd(const void* dy, const void* h, const void* alpha, int alpha_is_vector, const int* gate, void* dh,
                            float* fws, void* dalpha, int accumulate, int64_t rows, int D, void* stream) {
    AFK_REQUIRE(dy && h && alpha && dh && fws && dalpha && rows > 0 && D % 8 == 0, "afk_gate_bwd: bad args");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(ew_grid(rows * (D / 8), 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)h,
                       (const bf16*)alpha, alpha_is_vector, gate, (bf16*)dh, rows, D);
    const int ns = afk_colsum_slices(rows);
    const int rps = (int)afk_cdiv(rows, ns);
    float* cs = fws + (int64_t)ns * D;
    hipLaunchKernelGGL(gate_prod_colsum_kernel, dim3((unsigned)afk_cdiv(D, 64), (unsigned)ns), dim3(256), 0, ST, (const bf16*)dy,
                       (const bf16*)h, gate, rows, D, fws, rps);
    hipLaunchKernelGGL(colsum_fold_f32_kernel, dim3((unsigned)afk_cdiv(D, 256)), dim3(256), 0, ST, fws, ns, D, cs);
    hipLaunchKernelGGL(gate_alpha_grad_kernel, dim3(1), dim3(1024), 0, ST, cs, (const bf16*)alpha, alpha_is_vector, (bf16*)dalpha, D,
                       accumulate);
    AFK_LAUNCH_CHECK("afk_gate_bwd");
    return AFK_OK;
}
