/* afk — Audio Flamingo kernels for MI355X (gfx950).  C ABI of libafk.so.
 *
 * This is the drop-in boundary of the repo.  The reference has no FFI of its own: the
 * hot path it runs is PyTorch ATen ops called from the modules of
 *   TF = transformers 5.15 (models/audioflamingo3/modeling_audioflamingo3.py, models/qwen2/modeling_qwen2.py,
 *        models/whisper/feature_extraction_whisper.py, loss/loss_utils.py)
 * so every entry point below names the oracle expression (file:line) it replaces.  A host binds these with
 * ctypes / cffi / pybind; audio_flamingo_amd/_lib.py is the ctypes binding used here.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless named host_*.
 *   - bf16 tensors are row-major, "ld*" are leading dimensions in ELEMENTS.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no host sync, no
 *     allocation).  Calls are re-entrant and may be issued from any thread (autograd's backward thread too).
 *   - return 0 on success, negative on error; afk_last_error() returns the message for the calling thread.
 */
#ifndef AFK_H
#define AFK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int afk_version(void);
const char* afk_last_error(void);
/* sha256 prefix (16 hex digits) of csrc/*.hip + the headers this library was built from */
const char* afk_build_id(void);
/* 1 if the library was built with -DAFK_PROBES (make PROBES=1: timing probes that give WRONG results, rejected GEMM schedules); 0 for the shipped build */
int afk_has_probes(void);

/* ---- launch counters per kernel family: host_out[i] = launches of family i since the last reset (i < n <= AFK_CNT_MAX).
 * Test infrastructure: lets a parity test assert WHICH kernel served a shape (256x256 ping-pong GEMM, TN wgrad, GQA-split dK/dV...). */
#define AFK_CNT_GEMM_NT128 0
#define AFK_CNT_GEMM_NT256 1
#define AFK_CNT_GEMM_NN256 2
#define AFK_CNT_GEMM_TN256 3
#define AFK_CNT_GEMM_SPLITK 4
#define AFK_CNT_GEMV 5
#define AFK_CNT_ATTN2_FWD_D64 6
#define AFK_CNT_ATTN2_FWD_D128 7
#define AFK_CNT_ATTN2_BWD_D64 8
#define AFK_CNT_ATTN2_BWD_D128 9
#define AFK_CNT_GQA_REDUCE 10
#define AFK_CNT_XATTN_FWD 11
#define AFK_CNT_XATTN_BWD 12
#define AFK_CNT_ATTN1_FWD 13
#define AFK_CNT_ATTN1_BWD 14
#define AFK_CNT_GEMM_GENERIC 15 /* 256x256 launches served by a runtime-flag (generic-epilogue) instantiation: 0 on the AF3 training step */
#define AFK_CNT_MAX 16
int afk_kernel_counts(int64_t* host_out, int n);
int afk_kernel_counts_reset(void);

/* ---- profiling: HIP events around every afk_gemm_nt_bf16 launch (bench.py roofline leg) ------------- */
int afk_prof_enable(int on);
int afk_prof_reset(void);
int afk_prof_collect(double* host_total_ms, double* host_total_flops, int64_t* host_launches);
/* CSV of the profiling window: M,N,K,variant,ms per GEMM launch (host_path is a host string) */
int afk_prof_dump(const char* host_path);

/* ---- dense contraction ------------------------------------------------------------------------------
 * C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)  bf16 in, fp32 MFMA accumulate, bf16 (or f32) out.
 * Replaces F.linear at modeling_audioflamingo3.py:109-115 (q/k/v/out_proj), :209-210 (fc1/fc2),
 * :427-433 (projector), :576 (lm_head); modeling_qwen2.py:40-42 (gate/up/down), :189-192 (q/k/v/o);
 * and nn.Conv1d :328-329 after im2col.  Backward dgrad/wgrad use the same entry with transposed operands.
 * K must be a multiple of 64 and N of 4; M and N edges are handled in-kernel.
 * residual row index is m, or m % res_mod when res_mod > 0 (broadcast table: embed_positions add, :385).
 * epilogue order: +bias[n] -> (round bf16, write preact_out, GELU-erf) -> (round bf16, +residual[m,n]) -> (+C if ACCUM)
 */
/* kernel variant override for tests/benchmarks: 0 auto (by tile count), 1 = 128x128x64 kernel, 2 = 256x256x64 ping-pong kernel; + 16 = 8-byte epilogue
 * stores (same results); + 256 * g (g = 1..63) = rasterization group height.  Any other value selects a timing probe or a rejected schedule: those
 * exist only in -DAFK_PROBES builds (make PROBES=1) and are refused (AFK_ERR_ARG) by the default library. */
int afk_gemm_set_variant(int variant);
#define AFK_GEMM_BIAS 1
#define AFK_GEMM_GELU 2
#define AFK_GEMM_RESIDUAL 4
#define AFK_GEMM_OUT_F32 8
#define AFK_GEMM_ACCUM 16
#define AFK_GEMM_SWIGLU_BWD 32 /* C is [M, 2N]: (dgate | dup) = SwiGLU backward of the product, `residual` = saved gate|up [M, 2N] (Qwen2MLP, modeling_qwen2.py:46-48) */
#define AFK_GEMM_ROPE 128 /* afk_gemm_nt_bf16_rope only (with AFK_GEMM_BIAS or alone): rotate-half RoPE applied to the first rope_cols output columns in the epilogue */
#define AFK_GEMM_SWIGLU_FWD 64 /* NT only, B = the fused gate|up weight [2I, K] (N = 2I, I % 128 == 0): C [M, 2I] = gate|up pre-activations as without the flag, and preact_out [M, I] = bf16(bf16(silu(gate)) * up) - Qwen2MLP's activation (modeling_qwen2.py:46-48) from the same launch; no other flag */
int afk_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                     int res_mod, void* preact_out, float alpha, int flags, void* stream);
/* General form: trans_x = 0 -> operand stored [rows][K] (k contiguous, as above); trans_x = 1 -> stored reduction-major
 * [K][rows] (ld = row pitch).  Implemented: NT (0,0), NN (0,1) = dgrad  dX = dY . W  with W as nn.Linear stores it,
 * TN (1,1) = wgrad  dW = dY^T . X  with both activations as they lie.  For trans_a the reduction length K is free
 * (ragged last tile is masked in-kernel); reduction-major operands need their row count (M resp. N) % 8 == 0. */
int afk_gemm_bf16(int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                  int64_t ldc, int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                  int res_mod, void* preact_out, float alpha, int flags, void* stream);
/* Split-K form of afk_gemm_nt_bf16 for outputs with few tiles and a long reduction (decode-time Linear layers with a handful of rows,
 * weight gradients of narrow layers): `splits` workgroups per 128x128 output tile each reduce a K range into fp32 partials
 * workspace[splits][M][N] (splits*M*N*4 bytes, caller-owned), a second kernel sums them in fixed order (bit-deterministic) and applies
 * the same fused epilogue (bias / GELU / residual / accumulate).  Same oracle lines as afk_gemm_nt_bf16.
 * M == 1 (decode) takes a weight-streaming first pass instead of MFMA tiles: every weight row is read once, coalesced, against the M
 * activation rows held in registers (HBM-bound: 2*N*K bytes per launch); there `splits` <= ceil(K / 512) and may be 1. */
/* The qkv projection with the rotary embedding in its epilogue (round 6): C[M, N] = bf16(A.B^T + bias) as afk_gemm_nt_bf16, then - on the first rope_cols columns
 * (the q and k heads, 128 columns each) - apply_rotary_pos_emb (modeling_qwen2.py:112-135, called on q and k at :213) with afk_rope_inplace's rounding points
 * (the Linear output is a bf16 tensor, cos / sin are bf16, every product and the sum are rounded): the SAME bits as afk_gemm_nt_bf16 + afk_rope_inplace.
 * cos_t / sin_t [positions, 128] bf16 (16-byte aligned), pos [M] int32 or null (row % S).  Needs head_dim 128, N % 256 == 0, rope_cols % 256 == 0, the 16-byte
 * epilogue form (aligned C, ldc % 8 == 0) and a shape the 256 x 256 kernel takes (AFK_ERR_UNSUPPORTED otherwise: call the two-launch form).  bias nullable.
 * cos_lanes / sin_lanes (nullable; used when pos is null and S % 32 == 0): afk_rope_lanes_table(form = 1) copies for coalesced table reads in the epilogue. */
int afk_gemm_nt_bf16_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                          const void* cos_t, const void* sin_t, const int* pos, int S, int rope_cols, const void* cos_lanes, const void* sin_lanes, void* stream);
int afk_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                            const void* bias, const void* residual, int64_t ldr, int res_mod, void* preact_out, float alpha,
                            int flags, int splits, void* workspace, void* stream);
/* the same for the general form (NN / TN on the 256x256 transposed-operand kernels; splits <= ceil(K / 64)) */
int afk_gemm_bf16_splitk(int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                         int M, int N, int K, const void* bias, const void* residual, int64_t ldr, int res_mod, void* preact_out,
                         float alpha, int flags, int splits, void* workspace, void* stream);


/* ---- normalisation ------------------------------------------------------------------------------------
 * LayerNorm: nn.LayerNorm(eps=1e-5) at modeling_audioflamingo3.py:207,212 (per layer) and :335,403 (final).
 * RMSNorm:   Qwen2RMSNorm.forward, modeling_qwen2.py:247-252 (eps 1e-6, cast to bf16 BEFORE the weight multiply).
 * bwd: dx = norm-branch grad (+ dx_add if non-null: fused residual-gradient merge); dw/db (bf16) are
 * overwritten or accumulated; workspace = afk_norm_bwd_blocks(rows) * 2 * D floats.  Alignment (vector accesses): x, dy, dx, dx_add, w and
 * the workspace 16 bytes, dw / db 8 bytes; D % 8 == 0 takes the column-owned kernel (D <= 4096), D % 4 == 0 the row-per-wave form (D <= 3584). */
int afk_norm_bwd_blocks(int64_t rows);
int afk_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                      int64_t rows, int D, float eps, void* stream);
int afk_layernorm_bwd(const void* x, const void* w, const void* dy, const float* mean, const float* rstd,
                      void* dx, const void* dx_add, void* dw, void* db, int accumulate, float* workspace,
                      int64_t rows, int D, void* stream);
int afk_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int D, float eps, void* stream);
/* afk_gelu_bwd (dx = bf16(dy * gelu'(pre)), exact-erf GELU, activations.py:70-89) that also returns the column sums of the dx it writes: colsum[c] (+)= sum_r dx[r][c]
 * = the bias gradient of the Linear that produced `pre` (AudioFlamingo3EncoderLayer fc1 :240; torch: bias.grad = grad_output.sum(0)).  workspace:
 * afk_gelu_bwd_colsum_parts(rows) x C floats.  C % 8 == 0. */
int afk_gelu_bwd_colsum_parts(int64_t rows);
int afk_gelu_bwd_colsum(const void* dy, const void* pre, void* dx, int64_t rows, int C, void* colsum, int colsum_accumulate, float* workspace, void* stream);
/* afk_layernorm_bwd with dx_add (dx = bf16(norm branch) + dx_add: the gradient of the residual stream) that also returns the column sums of the dx it writes:
 * colsum[c] (+)= sum_r dx[r][c] = the bias gradient of the Linear whose output this LayerNorm normalised (AudioFlamingo3EncoderLayer :211-245: out_proj before
 * final_layer_norm, the lower layer's fc2 before self_attn_layer_norm; torch: bias.grad = grad_output.sum(0)) - the separate column-sum pass over dx disappears.
 * workspace: afk_norm_bwd_blocks(rows) x 3 x D floats.  D % 8 == 0, D <= 4096. */
int afk_layernorm_bwd_colsum(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, void* dx, const void* dx_add, void* dw, void* db,
                             int accumulate, void* colsum, int colsum_accumulate, float* workspace, int64_t rows, int D, void* stream);
int afk_rmsnorm_bwd(const void* x, const void* w, const void* dy, const float* rstd, void* dx, const void* dx_add,
                    void* dw, int accumulate, float* workspace, int64_t rows, int D, void* stream);

/* ---- layout + elementwise glue ------------------------------------------------------------------------
 * transpose: out[b1][b2][c][r] = in[b1][b2][r][c]; columns R..Rpad-1 of out are zero (K padding for wgrad GEMMs,
 *            and the [B,H,D,Spad] operand copies of the attention kernels). */
int afk_transpose_bf16(const void* in, void* out, int R, int C, int Rpad, int64_t ld_in, int64_t ld_out, int nb1,
                       int nb2, int64_t bs1_in, int64_t bs2_in, int64_t bs1_out, int64_t bs2_out, int max_blocks, void* stream);
/* max_blocks > 0 caps the (persistent) grid: a thin launch that co-resides with GEMM workgroups of another stream */
/* exact-erf GELU (transformers/activations.py:70-89) */
int afk_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int afk_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, void* stream);
/* SwiGLU: Qwen2MLP.forward modeling_qwen2.py:46-48; gu = [rows, 2I] (gate | up) */
int afk_silu_mul_fwd(const void* gu, void* h, int64_t rows, int I, void* stream);
int afk_silu_mul_bwd(const void* gu, const void* dh, void* dgu, int64_t rows, int I, void* stream);
/* rotate-half RoPE in place on the first nheads*D columns of buf[rows, ld]: apply_rotary_pos_emb modeling_qwen2.py:112-135 */
int afk_rope_inplace(void* buf, const void* cos_t, const void* sin_t, const int* pos, int64_t rows, int S, int ld,
                     int nheads, int D, int backward, void* stream);
int afk_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
int afk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);
/* y = (accumulate ? y : 0) + (*scale_dev) * x ; applies an upstream loss gradient held on the device without a host sync */
int afk_scale_add_bf16(const void* x, void* y, int64_t n, const float* scale_dev, int accumulate, void* stream);
/* out[r] (+)= sum_c in[r][c] : bias gradient from the transposed output-gradient */
int afk_rowsum_bf16(const void* in, int64_t ld, int C, void* out, int rows, int accumulate, void* stream);

/* out[c] (+)= sum_r in[r][c] : bias gradient straight from the row-major output gradient;
 * workspace = afk_colsum_slices(rows) * cols floats */
int afk_colsum_slices(int64_t rows);
int afk_colsum_bf16(const void* in, int64_t ld, int64_t rows, int cols, void* out, int accumulate, float* workspace, void* stream);
/* the same in ONE launch: the last row slice of every 64-column block folds the slices (fixed slice order: bit-identical to afk_colsum_bf16).
 * counters: ceil(cols / 64) ints that must read zero before the first call; the kernel leaves them at zero.  Launches that may overlap in time
 * (different streams) need different counter arrays. */
int afk_colsum_bf16_fused(const void* in, int64_t ld, int64_t rows, int cols, void* out, int accumulate, float* workspace, int* counters, void* stream);

/* ---- Flamingo glue (BASELINE config 4; stand-in oracle lines: transformers/models/idefics) --------------------------------
 * ReLU of the Perceiver MLP (perceiver.py:171-187); tanh-gated residual of the gated cross-attention layer
 * (modeling_idefics.py:792-793,800):  y = x + tanh(alpha) * (gate[row] ? h : 0), alpha a vector [D] or a scalar. */
int afk_relu_fwd(const void* x, void* y, int64_t n, void* stream);
int afk_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, void* stream);
int afk_gate_fwd(const void* x, const void* h, const void* alpha, int alpha_is_vector, const int* gate, void* y, int64_t rows,
                 int D, void* stream);
int afk_gate_bwd(const void* dy, const void* h, const void* alpha, int alpha_is_vector, const int* gate, void* dh,
                 float* fws, void* dalpha, int accumulate, int64_t rows, int D, void* stream);

/* ---- conv stem as GEMM (modeling_audioflamingo3.py:328-329,380-382) ------------------------------------ */
int afk_im2col_conv1(const void* x, int x_is_f32, void* col, int W, int C, int T, void* stream);
int afk_im2col_conv2(const void* h, void* col, int W, int Tin, int Tout, int C, void* stream);
int afk_col2im_conv2(const void* dcol, void* dh, int W, int Tin, int Tout, int C, void* stream);
int afk_conv_weight_permute(const void* in, void* out, int Co, int Ci, int dir, int accumulate, void* stream);
/* AvgPool1d(2,2) over time, :337,401-402 */
int afk_avgpool2_fwd(const void* x, void* y, int64_t out_rows, int C, void* stream);
int afk_avgpool2_bwd(const void* dy, void* dx, int64_t out_rows, int C, void* stream);

/* ---- embedding gather + <sound> placeholder scatter (:490-512, :532-545) ------------------------------- */
int afk_placeholder_scan(const int64_t* ids, int64_t n, int64_t audio_id, int* src, int* n_audio, void* stream);
int afk_embed_scatter_fwd(const int64_t* ids, const int* src, const void* embed, const void* audio, void* out,
                          int64_t n, int H, void* stream);
/* backward: d_audio[src[r]] = dout[r] for placeholder rows; d_embed[ids[r]] += sum of dout rows with that id, accumulated in fp32 in
 * row order and rounded once (bit-deterministic, no atomics).  perm = the n row indices sorted STABLY by ids (required with d_embed). */
int afk_embed_scatter_bwd(const int64_t* ids, const int* src, const void* dout, void* d_embed, void* d_audio,
                          const int* perm, int64_t n, int H, void* stream);

/* ---- attention (encoder :117-189 bidirectional 20x64; decoder modeling_qwen2.py:195-234 causal GQA 28:4x128) --
 * tensors are addressed base + b*bs + h*hs + s*rs + d (element strides) so q/k/v can live inside the fused
 * projection output; Vt/Kt/Qt/dOt are [B, H, D, Spad] transposed copies (afk_transpose_bf16, zero padded).
 * LSE/delta are [B, Hq, S] fp32.  kv_len[b] (or NULL) = number of valid keys (encoder key padding). */
int afk_attn_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                 int64_t k_rs, const void* Vt, void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, float* LSE,
                 const int* kv_len, int B, int Hq, int Hkv, int S, int Spad, int D, float scale, int causal, void* stream);
int afk_attn_delta(const void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, const void* dO, int64_t do_bs,
                   int64_t do_hs, int64_t do_rs, float* delta, int B, int H, int S, int D, void* stream);
int afk_attn_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                 int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO,
                 int64_t do_bs, int64_t do_hs, int64_t do_rs, const void* Qt, const void* Kt, const void* dOt,
                 const float* LSE, const float* delta, void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs,
                 void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV, int64_t dv_bs, int64_t dv_hs,
                 int64_t dv_rs, const int* kv_len, int B, int Hq, int Hkv, int S, int Spad, int D, float scale,
                 int causal, void* stream);

/* cross-attention form (BASELINE config 4: Perceiver resampler and Flamingo gated cross-attention; stand-in oracle lines
 * transformers/models/idefics/perceiver.py:106-168, modeling_idefics.py:776-802): Sq != Sk, optional per-query key range
 * krange[B, Sq, 2] = [begin, end) ("attend only to the media of your own segment"; an empty range yields a zero row).
 * Qt/dOt/LSE/delta use the query pitch Sqpad / Sq, Kt/Vt the key pitch Skpad. */
int afk_xattn_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                  int64_t k_rs, const void* Vt, void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, float* LSE,
                  const int* kv_len, const int* krange, int B, int Hq, int Hkv, int Sq, int Sk, int Sqpad, int Skpad,
                  int D, float scale, void* stream);
int afk_xattn_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                  int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO,
                  int64_t do_bs, int64_t do_hs, int64_t do_rs, const void* Qt, const void* Kt, const void* dOt,
                  const float* LSE, const float* delta, void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs,
                  void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV, int64_t dv_bs, int64_t dv_hs,
                  int64_t dv_rs, const int* kv_len, const int* krange, int B, int Hq, int Hkv, int Sq, int Sk,
                  int Sqpad, int Skpad, int D, float scale, void* stream);

/* ---- attention v2: LDS-staged tiles + ds_read_b64_tr_b16 transposed operands; no transposed copies in HBM.
 * Same oracle lines as afk_attn_*.  head_dim 64 / 128.  LSE and delta are [B, Hq, Spad] (Spad % 64 == 0, zero-initialised) and INTERNAL
 * to these three calls (forward: minus the log-sum-exp in score units; delta: minus rowsum(dO o O)).
 * kv_len[b] (nullable): keys >= kv_len[b] are padding.  kv_lo[b] (nullable, causal only): keys < kv_lo[b] are padding - the left-padded
 * batches of the reference processor (processing_audioflamingo3.py:46); query rows < kv_lo[b] produce zero output rows and zero gradients. */
int afk_attn2_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                  int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, void* O, int64_t o_bs,
                  int64_t o_hs, int64_t o_rs, float* LSE, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S, int Spad,
                  int D, float scale, int causal, void* stream);
/* A/B switch of the attention forward / dQ instruction schedule (round 6): 1 (default; AFK_ATTN_SCHED) = row fragments through explicit rings of opaque
 * ds_read_b128 groups, the dQ tile as two 32-key halves, LDS-DMA pieces dealt out between the MFMA groups; 0 = the compiler-scheduled reads of rounds 1-5.
 * Both compute the same MFMAs on the same operands in the same order: bit-identical results (tests/test_ops_gpu.py).  Replaces nothing in the reference:
 * it selects between two instruction streams of the same SDPA (sdpa_attention.py:79-166). */
int afk_attn_set_sched(int sched);
/* Forward / dQ block -> work map (round 6; AFK_ATTN_XCD): 1 (default) = 1-D grid whose linear block order hands every XCD (block id % 8) a contiguous chunk
 * of the work sorted by the K / V stream it reads - the query heads of a GQA group side by side at each causal level, the query blocks of a head back to back
 * when not causal - so that heads sharing K / V share an L2; 0 = the (heads, batch, query blocks) grid of rounds 2-5.  Same blocks, same arithmetic. */
int afk_attn_set_xcd_map(int on);
/* afk_attn2_fwd_persistent without its work queue (measurement of the paired-tile causal schedule): 1 = block k computes item k and then item total - 1 - k of
 * the longest-first item order, so every block sweeps the same number of key tiles; grid = ceil(total / 2).  Same results bit for bit. */
int afk_attn_set_persist_paired(int on);
/* GQA dK/dV sweep (afk_attn2_bwd*, gqa_scratch given): number of blocks - and of bf16 partial dK/dV images - per kv head.  Each block sweeps ~group / parts
 * query heads, accumulating in registers; `group` = one block per query head (rounds 2-5).  0 = AFK_ATTN_DKDV_PARTS / the default.  Fewer parts = fewer and
 * longer blocks, fewer partials through HBM, fewer roundings; the result differs from other part counts in the last bf16 bit (summation grouping). */
int afk_attn_set_dkdv_parts(int parts);
/* afk_attn2_fwd with RESIDENT blocks that pull (sample, head, 128-query block) items from an atomic queue, heavy first, and overlap the next item's cold
 * loads (first K / V tile, Q rows) with the current item's last tile and store tail (round 5, opt-in: AFK_ATTN_PERSIST=1).  Same results bit for bit.
 * queue: two device ints, zero before the first call, left at zero by every call (one queue per stream that launches concurrently).  Restrictions: no
 * kv_len / kv_lo, S % 128 == 0 - otherwise the call forwards to afk_attn2_fwd. */
int afk_attn2_fwd_persistent(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, void* O, int64_t o_bs,
                             int64_t o_hs, int64_t o_rs, float* LSE, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S, int Spad,
                             int D, float scale, int causal, int* queue, void* stream);
int afk_attn2_delta(const void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs,
                    int64_t do_rs, float* delta, int B, int H, int S, int Spad, int D, void* stream);
int afk_attn2_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                  int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO, int64_t do_bs,
                  int64_t do_hs, int64_t do_rs, const float* LSE, const float* delta, void* dQ, int64_t dq_bs,
                  int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV,
                  int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S,
                  int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream);
/* afk_attn2_bwd with delta = rowsum(dO o O) computed inside the dQ kernel (which then runs ahead of the dK/dV sweep): no afk_attn2_delta call;
 * delta_ws [B, Hq, Spad] fp32 is a workspace written by this call whose padding tail [S, Spad) must read zero. */
int afk_attn2_bwd_fused(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                        int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* O, int64_t o_bs, int64_t o_hs,
                        int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs, int64_t do_rs, const float* LSE, float* delta_ws,
                        void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs,
                        void* dV, int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq,
                        int Hkv, int S, int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream);
/* afk_attn2_bwd_fused with the BACKWARD of the rotary embedding applied to dQ and dK where their final values are formed (dQ kernel epilogue; GQA reduce or the
 * sweep's epilogue for dK): the reference rotates q and k ahead of the attention (apply_rotary_pos_emb, modeling_qwen2.py:112-135, called at :213) and its
 * autograd applies the transposed rotation to their gradients.  cos_t / sin_t: [positions, D] bf16, 16-byte aligned; pos: [B * S] int32 or null (row % S).
 * Result bits = afk_attn2_bwd_fused followed by afk_rope_inplace(backward = 1) on the q | k columns.  dV is not rotated.  cos_lanes / sin_lanes: see
 * afk_rope_lanes_table below (nullable). */
int afk_attn2_bwd_fused_rope(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* O, int64_t o_bs, int64_t o_hs,
                             int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs, int64_t do_rs, const float* LSE, float* delta_ws,
                             void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs,
                             void* dV, int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq,
                             int Hkv, int S, int Spad, int D, float scale, int causal, void* gqa_scratch, const void* cos_t, const void* sin_t,
                             const int* pos, const void* cos_lanes, const void* sin_lanes, void* stream);
/* cos_lanes / sin_lanes above (nullable, used only when pos is null): lane-major copies of the same tables, built by this call - table [rows, D] bf16 ->
 * out [ceil(rows / 32) * 32 * D] bf16, rows >= S.  With them the dQ epilogue reads its cos / sin pieces coalesced (a wave's 32 rows are one block of the copy);
 * without them each lane reads its own table row (32 cache lines per instruction).  Same results.
 * form 0 = the layout of the attention backward's dQ epilogue, form 1 = the layout of afk_gemm_nt_bf16_rope's epilogue (its cos_lanes / sin_lanes). */
int afk_rope_lanes_table(const void* table, void* out, int rows, int D, int form, void* stream);
/* gqa_scratch: NULL, or 2*B*S*Hq*D bf16 - enables the one-block-per-query-head dK/dV sweep + group reduce (GQA) */

/* Music Flamingo rotary time embedding on the encoder output (apply_rotary_time_emb, transformers/models/musicflamingo/
 * modeling_musicflamingo.py:187-204; angles :97-118): y[r, 2i..2i+1] = rotation of x[r, 2i..2i+1] by the angle whose cos / sin are
 * cos_t / sin_t [rows, R] (fp32) for 2i < R, y = x elsewhere.  backward != 0 applies the transposed rotation (gradient). */
int afk_rotary_time(const void* x, const float* cos_t, const float* sin_t, void* y, int64_t rows, int E, int R, int backward,
                    void* stream);

/* KV cache of the decode path (Qwen2Attention.forward modeling_qwen2.py:213-214 `past_key_values.update`): rows b*n+i of the fused
 * projection output (K at column k_col0, V right behind it) are appended at cache position start+i; K row-major
 * kcache[b][pos][Hkv*D] (batch stride kc_bs), V transposed vtcache[b][h][d][pos] with pitch spad (batch stride vt_bs) - the Vt operand
 * of afk_xattn_fwd.  start = *start_dev if start_dev != NULL (graph-replayable), else start_host. */
int afk_kv_cache_append(const void* qkv, int64_t ld, int k_col0, void* kcache, int64_t kc_bs, void* vtcache, int64_t vt_bs,
                        int spad, const int* start_dev, int start_host, int B, int n, int Hkv, int D, void* stream);

/* Decode-time attention over the KV cache, one query row per sample (Qwen2Attention.forward with past_key_values, modeling_qwen2.py:
 * 195-234): split-KV.  Q [B][Hq][D] (strides q_bs, q_hs), K cache [B][pos][Hkv][D] (k_bs, k_rs, k_hs), V^T cache [B][Hkv][D][spad]
 * (vt_bs) as written by afk_kv_cache_append, O like Q.  krange[B][2] = visible key interval [lo, hi) per sample, read on the device.
 * workspace: afk_attn_decode_workspace_floats(B, Hq, D, nsplit) floats.  head_dim 64 / 128. */
int afk_attn_decode_workspace_floats(int B, int Hq, int D, int nsplit);
int afk_attn_decode(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                    const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                    int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream);
/* the same in ONE launch: the last chunk block of each (batch, head) pair merges the chunks (agent-scope release / acquire around an arrival counter, no waiting).
 * The last B * Hq words of the workspace are those counters: zero before the first call, left at zero by every call. */
int afk_attn_decode_fused(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                          const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                          int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream);
/* Form of afk_attn_decode_fused for GQA caches: 0 one block per (sample, QUERY head, chunk); 1 = one block per (sample, KV head, chunk) serving all Hq / Hkv query
 * heads of the group on the vector units when the launch has >= 128 such blocks, 2 = whenever the form's limits hold (Hq / Hkv in {2, 4, 7, 8}, spad <= nsplit * 1024,
 * (Hq / Hkv) * nsplit * (D + 2) <= 8384) - forms 0 .. 2 are bit-identical to each other; 3 (round 6) = the group form on the matrix pipe (K.Q^T and Vt.P as
 * v_mfma_f32_32x32x16_bf16 chains, probabilities rounded to bf16 as SDPA's bf16 softmax output is: equal to the others within that rounding, not bit for bit);
 * -1 (default) = form 0, and form 3 by itself when form 0 would launch more than 1 024 blocks (B = 8 step of AF3-7B: 21.1 -> 17.6 us per layer).
 * Env AFK_ATTN_DECODE_GROUP sets the initial mode. */
int afk_attn_decode_set_group(int mode);

/* Decode-step glue (csrc/decode_glue.hip): the weight-streaming first pass alone, and one kernel per Linear of a decoder layer that sums
 * its fp32 partials ws[splits][M][N] and applies everything up to the next Linear's input (one live row per call today: M <= AFK_GEMV_MAX_M; same arithmetic and bf16
 * rounding points as the stand-alone kernels; oracle lines: modeling_qwen2.py:46-48, 112-135, 213-214, 247-252, 284-297). */
int afk_gemv_partials(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int splits, float* workspace,
                      void* stream);
int afk_decode_qkv_finish(const float* ws, int splits, int M, const void* bias, const void* cos_t, const void* sin_t, const int* pos,
                          void* q_out, void* kcache, int64_t kc_bs, void* vtcache, int64_t vt_bs, int spad, const int* start_dev,
                          int Hq, int Hkv, int D, void* stream);
int afk_decode_residual_rmsnorm(const float* ws, int splits, int M, int N, const void* residual, const void* w, float eps, void* x_out,
                                void* h_out, void* stream);
int afk_decode_swiglu(const float* ws, int splits, int M, int I, void* a_out, void* stream);

/* Decode step, ONE sequence, one launch per Linear (csrc/decode_chain.hip, round 4): a wave owns two complete output rows (no split-K partials), the
 * RMSNorm of the consumer's input is computed in its prologue and bias / RoPE / cache append / residual / SwiGLU in the producer's epilogue - five launches
 * per decoder layer with afk_attn_decode(nsplit = 1).  Same arithmetic and bf16 rounding points as the stand-alone kernels; oracle lines: modeling_qwen2.py:46-48
 * (SwiGLU), :112-135 (RoPE), :213-214 (cache), :247-252 (RMSNorm), :284-297 (residuals).  x: the residual stream / activation row [K]; K <= 4096 where a norm is fused. */
int afk_decode_chain_qkv(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int K, const void* bias, const void* cos_t,
                         const void* sin_t, const int* pos, void* q_out, void* kcache, void* vtcache, int spad, const int* start_dev, int Hq,
                         int Hkv, int D, void* stream);
int afk_decode_chain_linear_residual(const void* x, const void* W, int64_t ldw, int N, int K, const void* residual, void* out, void* stream);
int afk_decode_chain_gate_up(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int I, int K, void* act_out, void* stream);
/* final RMSNorm (Qwen2Model.norm, modeling_qwen2.py:383) + lm_head (modeling_audioflamingo3.py lm_head on the last position) of one row in one launch:
 * logits[N] fp32 = float(bf16(W . (norm_w * bf16(x * rstd))))  - the values `lm_head(norm(x)).float()` holds (logits may be null when only the token is wanted).
 * part_val / part_idx [N / 8] (both or neither): largest logit and its row of every group of eight consecutive rows (ties: lowest row) - the first stage of the
 * greedy selection GenerationMixin._sample does with torch.argmax (transformers/generation/utils.py:2790).  N % 8 == 0, K % 8 == 0. */
int afk_decode_chain_lm_head(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int N, int K, float* logits, float* part_val, int* part_idx,
                             void* stream);
/* second stage + the bookkeeping between two decode steps of one sequence, one launch: token = argmax over the nparts pairs (ties: lowest row);
 * next_token[0] = token; tokens_out[state[2] + tok_off] = token (tokens_out may be null); state = int32 [key-range start, key-range end, cache slot, position]:
 * the last three += 1; x_out[H] = emb[token][:H] (the embedding lookup of the next step, Qwen2Model.embed_tokens). */
int afk_decode_select_greedy(const float* part_val, const int* part_idx, int nparts, int64_t* next_token, int64_t* tokens_out, int tok_off, int* state,
                             const void* emb, int64_t ld_emb, int H, void* x_out, void* stream);
/* The same launches for 2 .. 8 sequences decoded together (one new position each; the weights are still read once per step): M input rows h [M][K] (row stride
 * ldh) that are ALREADY normalised where the Linear follows a norm (Qwen2DecoderLayer :270 / :294, Qwen2Model.norm); pos[M] = position of each sequence's new token,
 * *start_dev = the cache slot all of them write; q_out [M][Hq*D] (row stride ldq); k_bs / vt_bs = batch strides of the K / V^T caches (elements).  Rounding points as above. */
/* (round 6) the four _batched entry points below also take 9 .. 32 sequences - groups of eight as further columns of the same MFMA, input rows through the wave-private
 * LDS strip (K % 64 == 0; 32-row / 16-row output groups: N % 32 == 0) - where rounds 3-5 fell back to split-K tiles + glue kernels. */
int afk_decode_chain_qkv_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int K, const void* bias, const void* cos_t, const void* sin_t,
                                 const int* pos, void* q_out, int64_t ldq, void* kcache, int64_t k_bs, void* vtcache, int64_t vt_bs, int spad, const int* start_dev,
                                 int Hq, int Hkv, int D, void* stream);
int afk_decode_chain_linear_residual_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual, int64_t ld_res,
                                             void* out, int64_t ld_out, void* stream);
/* The batched launches with the RMSNorm in front of the Linear taken in the Linear's OWN prologue (round 6; matrix-pipe form, 32-row groups): x = the raw residual
 * stream rows [M][ldx], norm_w / eps = input_layernorm (qkv, Qwen2DecoderLayer :271), post_attention_layernorm (gate|up, :294), Qwen2Model.norm (lm_head).  Every
 * block derives the row statistic itself (its K split covers all columns) and normalises the rows it consumes through a wave-private LDS strip (Qwen2RMSNorm
 * :247-252, cast before the weight multiply): no norm launch, no hand-over between blocks.  ss_part / ss_nparts: the producer's partial sums (below), or null / 0.
 * Otherwise as the _batched entry points above.  K % 64 == 0.  1 <= M <= 32: nine and more sequences run as two / four GROUPS of eight (more columns of the same MFMA) in one pass over the
 * weights (every group an independent instance of the eight-sequence arithmetic); ss_part is then [G][8][ss_nparts], G = 2 (M <= 16) or 4.  ss_nparts % 4 == 0, ss_part 16-byte aligned. */
int afk_decode_chain_qkv_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int K, const void* bias,
                                      const void* cos_t, const void* sin_t, const int* pos, void* q_out, int64_t ldq, void* kcache, int64_t k_bs, void* vtcache,
                                      int64_t vt_bs, int spad, const int* start_dev, int Hq, int Hkv, int D, const float* ss_part, int ss_nparts, void* stream);
int afk_decode_chain_gate_up_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int I, int K, void* act_out,
                                          int64_t ld_act, const float* ss_part, int ss_nparts, void* stream);
int afk_decode_chain_lm_head_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int N, int K, float* logits,
                                          int64_t ld_logits, const float* ss_part, int ss_nparts, void* stream);
/* afk_decode_chain_linear_residual_batched on the matrix-pipe form that also leaves ss_part[8][N / 16]: every block's share of sum_n out[m][n]^2 per sequence m (N % 64 == 0).  Passed
 * to the NEXT Linear's afk_decode_chain_*_norm_batched (ss_part, ss_nparts = N / 16) it replaces that launch's own pass over the rows: every wave folds the
 * partial sums in part order (bit-reproducible).  ss_part = null there: the block takes the statistic from x itself (the first layer, whose rows no Linear wrote).
 * 1 <= M <= 32; ss_part holds G x 8 x (N / 16) floats, G = the groups of eight sequences the launch RUNS: 1 (M <= 8), 2 (M <= 16) or 4 (an empty group writes zeros). */
int afk_decode_chain_linear_residual_ss_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual, int64_t ld_res,
                                                void* out, int64_t ld_out, float* ss_part, void* stream);
/* Linear + residual + the RMSNorm that follows it in ONE launch, 1 .. 8 sequences (round 6): out = bf16(W x) + residual (Qwen2DecoderLayer :284 / :297), h_out =
 * RMSNorm(out) with norm_w (the layer's post_attention_layernorm :294, the NEXT layer's input_layernorm :271, or Qwen2Model.norm; Qwen2RMSNorm :247-252: cast before
 * the weight multiply).  The statistic needs every output column: the last block of the launch to arrive normalises (hand-over through agent-scope stores and a
 * self-resetting counter, as afk_attn_decode_fused).  counter: one zero-initialised int32.  N % 32 == 0, N <= 4096, K % 64 == 0; out / h_out 4-byte aligned. */
int afk_decode_chain_linear_residual_norm_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual, int64_t ld_res,
                                                  void* out, int64_t ld_out, const void* norm_w, float eps, void* h_out, int64_t ld_h, int* counter, void* stream);
int afk_decode_chain_gate_up_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int I, int K, void* act_out, int64_t ld_act, void* stream);
int afk_decode_chain_lm_head_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int N, int K, float* logits, int64_t ld_logits, void* stream);

/* ---- loss: ForCausalLMLoss / fixed_cross_entropy, loss/loss_utils.py:33-72 ----------------------------- 
 * logits chunk [rows, V] bf16 is overwritten with d(loss)/d(logits) when write_grad; row_loss[rows] fp32;
 * denom = device scalar (number of valid labels, or num_items_in_batch). */
int afk_ce_fwd_bwd(void* logits, int64_t ld, int64_t rows, int V, const int64_t* shift_labels, float* row_loss,
                   const float* denom, float upstream, int write_grad, void* stream);
int afk_count_valid(const int64_t* labels, int64_t n, float* out, void* stream);
int afk_loss_reduce(const float* row_loss, int64_t n, const float* denom, float* loss, int accumulate, void* stream);

/* ---- log-mel frontend: WhisperFeatureExtractor._torch_extract_fbank_features, feature_extraction_whisper.py:135-168
 * wav [W, nsamp] fp32 -> out [W, nmel, nsamp/160] (fp32 or bf16).  cosb/sinb = hann-folded DFT basis
 * [400, nbins_pad]; melT = [201, nmel]; raw_ws = W*nmel*T floats, wmax_ws = W ints. */
int afk_logmel(const float* wav, int W, int64_t nsamp, const float* cosb, const float* sinb, int nbins_pad,
               const float* melT, int nmel, float* raw_ws, int* wmax_ws, void* out, int out_is_bf16, void* stream);

/* ---- optimizer: AdamW on a flat parameter arena (bf16 param + fp32 master/m/v, 28 B/param) -------------
 * gate (may be NULL): device int; when it reads 0 the launch leaves every buffer untouched (torch.optim skips parameters whose
 * grad is None; under data parallelism "did any rank produce a gradient for this bucket" is only known on the device).
 * hyper (may be NULL): device float[4] = {lr, 1 - beta1^t, sqrt(1 - beta2^t), gradient multiplier}; when given it overrides lr and the bias
 * corrections derived from `step`, so that a captured HIP graph of the training step replays with the current values (afk_set_f32 writes
 * them), and hyper[3] multiplies every gradient (the global-norm clip coefficient written by afk_clip_coef; 1 = no clipping). */
int afk_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks, const int* gate,
                   const float* hyper, void* stream);
/* the same update on ONE 2-D weight [N, K] (N, K multiples of 64) that also writes the K-major copy of the new bf16 weight, shadow [K, ld_shadow]:
 * the W^T operand of the dgrad GEMM comes out of the optimizer instead of a separate afk_transpose_bf16 pass; parameters bit-identical to
 * afk_adamw_step, shadow == transpose(param). */
int afk_adamw_step_t(float* master, float* m, float* v, const void* grad, void* param, void* shadow, int N, int K, int64_t ld_shadow, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks, const int* gate,
                     const float* hyper, void* stream);
int afk_set_f32(float* dst, int n, float a, float b, float c, float d, void* stream);
/* global-norm gradient clipping (torch.nn.utils.clip_grad_norm_, TORCH/nn/utils/clip_grad.py; HF Trainer default max_grad_norm = 1.0,
 * TF/trainer.py): acc[0] += sum of squares of a bf16 range (skipped when *gate == 0; gate may be NULL), deterministic two-stage fold;
 * workspace = afk_sumsq_workspace_floats() floats.  afk_clip_coef: sumsq[0..n) = partial sums (one slot per gradient bucket, folded in index
 * order whatever schedule filled them), coef[0] = min(1, max_norm / (sqrt(sum) * scale + 1e-6)), norm_out[0] (may be NULL) = sqrt(sum) * scale. */
int afk_sumsq_workspace_floats(void);
int afk_sumsq_bf16(const void* x, int64_t n, float* acc, const int* gate, float* workspace, void* stream);
int afk_clip_coef(const float* sumsq, int n, float scale, float max_norm, float* coef, float* norm_out, void* stream);

/* ---- data-parallel gradient exchange: RCCL over xGMI behind the C ABI.
 * Replaces the C++ Reducer -> ncclAllReduce path of torch DistributedDataParallel (TORCH/nn/parallel/distributed.py:662-666, 828-834):
 * the gradient arena keeps a transformer layer's gradients contiguous, so a bucket is (pointer, count).  Every call enqueues on `stream`
 * (a side HIP stream ordered after the layer's last wgrad kernel by an event) and returns; the reduction is a SUM (averaging = the
 * optimizer's grad_scale).  One communicator per process (= per GPU): rank 0 creates the 128-byte id with afk_comm_unique_id, hands it to
 * the other ranks out of band (launcher / rendezvous store), every rank calls afk_comm_init on its own device.
 * afk_reduce_scatter_allgather_bucket gives the same result as afk_allreduce_bucket with every xGMI link of the mesh busy. */
#define AFK_COMM_UID_BYTES 128
#define AFK_COMM_BF16 0
#define AFK_COMM_F32 1
#define AFK_COMM_I32 2
int afk_comm_unique_id(char* host_out128);
int afk_comm_init(int rank, int world, const char* host_uid128, void** host_comm_out);
int afk_comm_destroy(void* comm);
int afk_allreduce_bucket(void* comm, void* buf, int64_t n, int dtype, int op_max, void* stream);
int afk_reduce_scatter_allgather_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream);
/* The two halves of the call above, for an optimizer that is SHARDED across the data-parallel ranks between them (round 5; the hook point the oracle
 * exposes is the DDP communication hook, TORCH/distributed/algorithms/ddp_comm_hooks/default_hooks.py:18-35; its DDP itself replicates the optimizer):
 *   afk_comm_share(n, world, dtype)   elements per rank: (n / world) rounded down to a multiple of 128 bytes.  Rank r OWNS buf[r * share, (r + 1) * share);
 *                                     the tail buf[share * world, n) is replicated (every rank reduces and updates it).
 *   afk_reduce_scatter_bucket         in place: afterwards rank r's own share and the tail hold the SUM over the ranks; the other shares hold what
 *                                     they held before (this rank's local values) and must not be read as reduced gradients.
 *   afk_allgather_bucket              in place: every rank's own share is distributed to all ranks; the tail is left alone (it is already equal
 *                                     everywhere).  Called on the bf16 PARAMETER bucket after each rank's AdamW update of its share: same bytes on
 *                                     the wire as gathering the gradients, 1 / world of the optimizer traffic and fp32 state per GPU. */
int64_t afk_comm_share(int64_t n, int world, int dtype);
int afk_reduce_scatter_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream);
int afk_allgather_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream);
int afk_comm_broadcast(void* comm, void* buf, int64_t n, int dtype, int root, void* stream);
/* CU-contention probe (pre-flight of the multi-GPU run; models the CUs RCCL's persistent channel kernels take from the GEMM rounds - reference
 * behaviour being prepared for: TORCH/nn/parallel/distributed.py:1012,1442): parks `nblocks` workgroups of 64 threads + lds_bytes of LDS on `stream`
 * until *stop_flag (device memory) becomes non-zero or max_ticks of the 100 MHz clock have passed.  report (nullable, device, int64 [nblocks][3]):
 * start tick, ticks alive, XCC id << 32 | HW_ID of every parked workgroup - proof that they sat where the GEMM workgroups wanted to be. */
int afk_cu_hog(int nblocks, int lds_bytes, const int* stop_flag, int64_t max_ticks, int64_t* report, void* stream);

/* ---- measured ceiling of the matrix pipe under the socket power cap (round 6; measurement utility like afk_cu_hog, no reference counterpart: the
 * reference's GEMMs are the vendor BLAS behind TORCH F.linear, modeling_qwen2.py:46-48).  Launches `nblocks` workgroups of 8 waves that run `iters`
 * segments of 16 v_mfma_f32_32x32x16_bf16 (2 x 2 tiles x 4 k-steps, the MFMA segment of gemm_nt_bf16_k256) on operands taken from `operands`
 * (device, bf16, >= 65536 elements, e.g. N(0,1)).  mode 0: operands resident in registers, no memory access inside the loop.  mode 1: the GEMM's
 * LDS fragment traffic added (12 ds_read_b128 per segment from a 64 KiB LDS image of `operands`).  *host_flops (nullable) receives the flops
 * of the launch; time it with HIP events on `stream`.  sink: device float, never written in practice.  modes 2 / 3: issue pacing of ONE wave per SIMD
 * (4-wave workgroups) over four accumulators - round-robin (2) or each accumulator four times in a row (3: back-to-back dependent MFMAs);
 * sink[0] = shader cycles per MFMA, sink needs 2 floats. */
int afk_mfma_ceiling(int mode, int nblocks, int iters, const void* operands, float* sink, double* host_flops, void* stream);

/* ---- HIP streams with an explicit queue priority (round 6).  The training step runs on three streams: the critical path (forward, dgrad, attention,
 * norms), the weight-gradient GEMMs and the optimizer / gradient exchange (reference counterpart: DDP's separate reduction stream,
 * TORCH/nn/parallel/distributed.py:1442; autograd itself is single-stream there).  torch.cuda.Stream can only ask for "normal" or "high"; these entry points
 * expose the device's whole range so that the side streams can sit BELOW the default and the compute stream above it (torch wraps the handle with
 * torch.cuda.ExternalStream).  host_least / host_greatest receive hipDeviceGetStreamPriorityRange (numerically greater = lower priority);
 * afk_stream_create clamps `priority` into that range, creates a non-blocking stream on the current device and stores the hipStream_t in *host_stream_out. */
int afk_stream_priority_range(int* host_least, int* host_greatest);
int afk_stream_create(int priority, void** host_stream_out);
/* A stream restricted to logical compute units [first_cu, first_cu + n_cus) of the current device (hipExtStreamCreateWithCUMask; default queue priority).  The mask
 * bits are dealt out XCD-first, so 8 k consecutive CUs are k CUs on every XCD.  For the HBM-bound side work of the step (the fused AdamW of a bucket beside the
 * MFMA-bound backward): the step is power-limited, CUs taken from the GEMMs cost them nothing, and the optimizer's waves stop landing on every CU of the chip.
 * No reference counterpart (torch.optim runs after backward on the one stream, TORCH/optim/adamw.py). */
int afk_stream_create_cu_mask(int first_cu, int n_cus, void** host_stream_out);
int afk_stream_destroy(void* stream);

/* ---- EXACT fp32 inference forward (round 5; SURVEY.md 8c "an fp32 mode of our kernels ... should give bit-exact tokens unconditionally on the tiny config").
 * A VERIFICATION mode (AFK_EXACT_FP32=1, audio_flamingo_amd/exact.py), not a fast path: activations fp32, weights the bf16 values of the checkpoint widened in
 * registers, every product and sum fp32 (v_mfma_f32_32x32x2_f32 for the Linears, fp32 VALU elsewhere) - no bf16 rounding point between the log-mel features
 * and the logits.  Same oracle lines as the bf16 entry points they mirror.
 *   afk_x32_linear      C[M,N] = epi((A[M,K] . W[N,K]^T + bias[n]) * alpha): optional exact-erf GELU, then + residual[m or m % res_mod][n].  K % 8 == 0.
 *   afk_x32_norm        LayerNorm (rms = 0: weight + bias, eps as given) or Qwen2 RMSNorm (rms = 1, b may be NULL), one row = D floats.
 *   afk_x32_attention   softmax(Q K^T * scale) V with fp32 softmax; row (b, s) of head h lies at X + (b * S + s) * ldx + h * D; GQA by Hq / Hkv; visible keys of
 *                       query s of sample b: [kv_lo[b], min(kv_len[b], causal ? s + 1 : S)) (NULL = 0 / S); a row without a visible key yields zeros.
 *   afk_x32_rope        rotate-half RoPE in place on the first `nheads` heads of every fused row (position = row % S), cos / sin tables [S, D] fp32.
 *   afk_x32_silu_mul    out[r][c] = silu(gu[r][c]) * gu[r][I + c].
 *   afk_x32_conv3_gelu  y = gelu(conv1d(x, w[E][C][3], pad 1, stride)) (+ pos[t][e]); x is [W][C][T] (x_cmajor) or [W][T][C]; y [W][T_out][E].
 *   afk_x32_avgpool2    mean of row pairs;  afk_x32_embed_scatter  out[row] = audio[src[row]] if src[row] >= 0 else embed[ids[row]]. */
int afk_x32_linear(const float* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K, const void* bias,
                   const float* residual, int64_t ldr, int res_mod, float alpha, int gelu, void* stream);
int afk_x32_norm(const float* x, const void* w, const void* b, float* y, int64_t rows, int D, float eps, int rms, void* stream);
int afk_x32_attention(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo, int B, int S, int Hq,
                      int Hkv, int D, float scale, int causal, const int* kv_lo, const int* kv_len, void* stream);
int afk_x32_rope(float* x, int64_t ld, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int nheads, int D, void* stream);
int afk_x32_silu_mul(const float* gate_up, float* out, int64_t rows, int I, void* stream);
int afk_x32_conv3_gelu(const float* x, int x_cmajor, const void* w, const void* bias, const void* pos, float* y, int W, int C, int T, int E, int stride,
                       void* stream);
int afk_x32_avgpool2(const float* x, float* y, int64_t W, int T, int E, void* stream);
int afk_x32_embed_scatter(const int64_t* ids, const int* src, const float* audio, const void* embed, float* out, int64_t rows, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif
