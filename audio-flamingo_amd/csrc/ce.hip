// Shifted causal-LM cross-entropy over bf16 logits, forward + gradient in one sweep (gfx950, HBM-bound).
//
// Oracle: ForCausalLMLoss / fixed_cross_entropy, transformers/loss/loss_utils.py:33-72 - logits.float(),
// labels padded with -100 and shifted by the host, F.cross_entropy(ignore_index=-100) with mean (or sum /
// num_items_in_batch) reduction.  The lm_head GEMM (modeling_audioflamingo3.py:625-627) produces the logits
// in row chunks; this kernel turns a chunk into per-row losses and overwrites it with d(loss)/d(logits), so
// the [B*S, 152064] fp32 tensor the oracle materialises (623 MB / sample) never exists.
// One 256-thread block per row: pass 1 = online max / sum-exp (16-byte loads), pass 2 = write gradient.
// Algorithmic traffic per row: 2*V bytes read + 2*V written (+ the second read, served from L2: V*2 = 304 KB).
#include "common.h"
#include "../../include/afk.h"

namespace {

__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(bf16* __restrict__ logits, int64_t ld, int V,
                                                         const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                                                         const float* __restrict__ denom, float upstream, int write_grad) {
    __shared__ float scratch[8];
    const int64_t row = blockIdx.x;
    bf16* x = logits + row * ld;
    const int64_t label = labels[row];
    const int tid = threadIdx.x;
    const int nv = V >> 3;
    if (label < 0) {  // ignore_index
        if (tid == 0) row_loss[row] = 0.f;
        if (write_grad) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
            for (int v = tid; v < nv; v += 256) *(bf16x8*)(x + 8 * v) = z;
            for (int c = nv * 8 + tid; c < V; c += 256) x[c] = (bf16)0.f;
        }
        return;
    }
    float m = -INFINITY, s = 0.f;
    for (int v = tid; v < nv; v += 256) {
        const bf16x8 t = *(const bf16x8*)(x + 8 * v);
        float f[8], mx = m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = (float)t[e];
            mx = fmaxf(mx, f[e]);
        }
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += __expf(f[e] - mx);
        s = s * __expf(m - mx) + acc;
        m = mx;
    }
    for (int c = nv * 8 + tid; c < V; c += 256) {
        const float f = (float)x[c];
        const float mx = fmaxf(m, f);
        s = s * __expf(m - mx) + __expf(f - mx);
        m = mx;
    }
    const float gm = block_max<4>(m, scratch);
    s = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    const float gs = block_sum<4>(s, scratch);
    const float lse = gm + __logf(gs);
    const float xl = (float)x[label];
    if (tid == 0) row_loss[row] = lse - xl;
    if (!write_grad) return;
    const float scale = upstream / fmaxf(*denom, 1.f);
    const float inv = scale / gs;
    __syncthreads();
    for (int v = tid; v < nv; v += 256) {
        const bf16x8 t = *(const bf16x8*)(x + 8 * v);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = __expf((float)t[e] - gm) * inv;
            if ((int64_t)(8 * v + e) == label) g -= scale;
            o[e] = (bf16)g;
        }
        *(bf16x8*)(x + 8 * v) = o;
    }
    for (int c = nv * 8 + tid; c < V; c += 256) {
        float g = __expf((float)x[c] - gm) * inv;
        if (c == label) g -= scale;
        x[c] = (bf16)g;
    }
}

// count = #labels != ignore ; single block, deterministic
__global__ __launch_bounds__(1024) void count_valid_kernel(const int64_t* __restrict__ labels, int64_t n, float* __restrict__ out) {
    __shared__ float scratch[16];
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) c += (labels[i] >= 0) ? 1.f : 0.f;
    c = block_sum<16>(c, scratch);
    if (threadIdx.x == 0) *out = c;
}

// loss = (acc_in ? *loss : 0) + sum(row_loss)/denom ; single block, fixed order -> deterministic
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ row_loss, int64_t n,
                                                           const float* __restrict__ denom, float* __restrict__ loss, int accumulate) {
    __shared__ float scratch[16];
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) c += row_loss[i];
    c = block_sum<16>(c, scratch);
    if (threadIdx.x == 0) {
        const float v = c / fmaxf(*denom, 1.f);
        *loss = accumulate ? *loss + v : v;
    }
}

}  // namespace

extern "C" int afk_ce_fwd_bwd(void* logits, int64_t ld, int64_t rows, int V, const int64_t* shift_labels, float* row_loss,
                              const float* denom, float upstream, int write_grad, void* stream) {
    AFK_REQUIRE(logits && shift_labels && row_loss && denom && rows > 0 && V > 0, "afk_ce_fwd_bwd: bad args");
    AFK_REQUIRE(ld % 8 == 0 && ((uintptr_t)logits % 16 == 0), "afk_ce_fwd_bwd: logits rows must be 16-byte aligned");
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (bf16*)logits, ld, V,
                       shift_labels, row_loss, denom, upstream, write_grad);
    AFK_LAUNCH_CHECK("afk_ce_fwd_bwd");
    return AFK_OK;
}

extern "C" int afk_count_valid(const int64_t* labels, int64_t n, float* out, void* stream) {
    AFK_REQUIRE(labels && out && n > 0, "afk_count_valid: bad args");
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n, out);
    AFK_LAUNCH_CHECK("afk_count_valid");
    return AFK_OK;
}

extern "C" int afk_loss_reduce(const float* row_loss, int64_t n, const float* denom, float* loss, int accumulate, void* stream) {
    AFK_REQUIRE(row_loss && denom && loss && n > 0, "afk_loss_reduce: bad args");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, n, denom, loss, accumulate);
    AFK_LAUNCH_CHECK("afk_loss_reduce");
    return AFK_OK;
}
