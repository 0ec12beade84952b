/*
 * pmhip.h -- C ABI of libpaintmind_hip.so: the MI355X (gfx950) generation path of PaintMind.
 *
 * The reference (Qiyuan-Ge/PaintMind) has no FFI: its boundary is a Python object protocol whose
 * arithmetic is delegated to PyTorch ATen.  This header is the boundary a maintainer would bind
 * instead (ctypes stub in INTEGRATION.md).  Every entry point cites the reference code whose
 * arithmetic it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless the name ends in _host
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream
 *   - return value: 0 = ok, otherwise a PMHIP_E* code; pmhip_last_error() gives the message
 *     (thread-local).  Nothing throws across this boundary.
 *   - the caller owns every tensor and every weight; the library owns only its handles'
 *     workspace arenas (hipMalloc'ed, grown on demand, freed by *_destroy)
 *   - handles are single-stream objects, not thread-safe
 *   - matrices are row-major; "T" below is the handle's compute dtype (PMHIP_F32 or PMHIP_BF16);
 *     weights are stored [out_features, in_features] exactly like torch.nn.Linear
 *   - GEMM reduction widths (K) must be multiples of 64 elements; the host packer zero-pads
 */
#ifndef PMHIP_H
#define PMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMHIP_ABI_VERSION 10

enum { PMHIP_OK = 0, PMHIP_EINVAL = 1, PMHIP_EHIP = 2, PMHIP_ENOMEM = 3, PMHIP_ESTATE = 4 };
enum { PMHIP_F32 = 0, PMHIP_BF16 = 1 };
/* what a block of packed projection rows is, for pmhip_gemm_heads */
enum { PMHIP_PART_Q = 0, PMHIP_PART_K = 1, PMHIP_PART_V = 2 };

typedef void* pmhip_stream;

int pmhip_abi_version(void);
const char* pmhip_last_error(void);
/* number of compute units / LDS bytes per workgroup / gcnArchName of `device` */
int pmhip_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Operator level (stateless).  These are what the reference's operator plug-point would call
 * (Layer.ATTENTION_MODES, stage1/layers.py:41-48, stage2/transformer.py:29-36; SwiGLU swap,
 * modules/mlp.py:34-40).
 * ---------------------------------------------------------------------------------------------- */

/* out[M,N] = A[M,K] . W[N,K]^T (+ bias[N]) (+ residual[m % res_rows, N]);  nn.Linear forward
 * (modules/attention.py:46-49,59; stage1/vqmodel.py:23,28; stage1/layers.py:149;
 * stage2/transformer.py:81,85,91).  A, W are `dtype`; bias/residual fp32 or NULL; out is
 * `out_dtype`.  res_rows lets one [tokens, N] table (a position embedding,
 * stage1/layers.py:108,146, stage2/transformer.py:82) be added to every image. */
int pmhip_gemm(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
               const float* residual, int ldr, int res_rows, void* out, int ldo, int out_dtype,
               int M, int N, int K, pmhip_stream stream);

/* SwiGLU first half (modules/mlp.py:27-30): out[M,Hp] = silu(x1) * x2 with
 * [x1|x2] = A . W12^T + b12.  W12p/b12p are the *packed* form: rows interleaved in groups of 16
 * (16 rows of x1, the matching 16 rows of x2, ...), hidden width zero-padded to Hp (mult. of 64). */
int pmhip_gemm_swiglu(int dtype, const void* A, int lda, const void* W12p, const float* b12p,
                      void* out, int ldo, int M, int Hp, int K, pmhip_stream stream);

/* Head-split projection (modules/attention.py:46-52): A[M,K] . W[nparts*inner,K]^T, no bias,
 * written per part as  Q -> [B,H,tokens,64] * q_scale ;  K -> [B,H,tokens_pad,64] ;
 * V -> transposed [B,H,64,tokens_pad].  M = B*tokens, inner = heads*64 (dim_head is 64). */
int pmhip_gemm_heads(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K,
                     int heads, int tokens, int tokens_pad, int nparts, const int* part_kinds_host,
                     void* const* part_outs_host, float q_scale, pmhip_stream stream);

/* ---- bf16 mode (the model-level default since round 3; PMHIP_HILO=0 restores the fp32 stream; DESIGN.md sections 4d, 4e): the
 * residual stream as a bf16 PAIR, and the LayerNorm folded into the GEMM that consumes it
 * (stage1/layers.py:54-58, stage2/transformer.py:44-49: x = f(LN(x)) + x, every projection preceded by a LayerNorm).
 * x = hi + lo with hi = bf16(x), lo = bf16(x - hi): two bf16 planes [M, D] -- the same 4 bytes per element as fp32 and, for a
 * stream built by adding bf16-GEMM outputs, the same accuracy (tools/residual_precision_probe.py: logits identical to the
 * fp32 stream to 5e-4, against +65 % error for a single bf16 plane).  What it buys: the hi plane IS bf16(x), the operand of
 * the next projection, so with  y = LN(x) = (x - mean) * rstd * gamma + beta  and
 *     y . W^T = rstd * (x . (gamma (.) W)^T) - rstd * mean * c + d,   c[n] = sum_k gamma[k] W[n,k],  d[n] = sum_k beta[k] W[n,k]
 * the projection multiplies hi by gamma-scaled weights and normalises in its epilogue; the separate LayerNorm pass over
 * the stream (read 4 B + write 2 B per element, 21 % of the GPU time of a decode step in round 2) is replaced by
 * pmhip_ln_coef, which reads the hi plane only.  Deterministic: no atomics. */

/* (hi, lo) <- split(A[M,K] . W[N,K]^T + bias + (res_hi + res_lo)[m % res_rows]); bf16 operands.  In place when the output
 * planes are the residual planes.  N, ldr, ldo multiples of 8. */
int pmhip_gemm_hilo(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                    const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N,
                    int K, pmhip_stream stream);
/* The same, and row_stats[M][N/64][2] = per row and per 64-column part (sum, sum of squares centred on the part's own mean) of
 * the NEW hi plane, written by the epilogue (N a multiple of 64): the LayerNorm folded into the next GEMM then needs no pass
 * over the plane -- pmhip_ln_coef_parts below instead of pmhip_ln_coef. */
int pmhip_gemm_hilo_stats(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                          const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N,
                          int K, float* row_stats, pmhip_stream stream);

/* Round 4, the CENTRED hi plane.  The folded LayerNorm normalises bf16(x); a row whose common offset is large against its spread
 * (trained transformers: massive activations, drifting row means) then loses 2^-9 |x| / std per element.  LayerNorm is blind to
 * a per-row shift, so this producer stores the pair of x - c, c = the row mean of the PREVIOUS hi plane read from `center_coef`
 * ([M][2] = the (rstd, -rstd * mean) pairs of pmhip_ln_coef / pmhip_ln_coef_parts for the LayerNorm in front of this branch;
 * NULL = no centring) plus `center_extra`, a launch-wide constant (the mean of `bias` over its N columns: the part of the new row
 * mean that is known in advance; a sudden common offset is then removed in the producer that introduces it).  `shift` ([M] floats, optional): running sum of the subtracted values for callers that need the absolute
 * x again (pmhip_unshift_hilo); shift_mode 1 = this producer opens the stream (shift <- 0), 2 = shift += c, 0 = untouched.
 * Everything else as pmhip_gemm_hilo_stats (row_stats may be NULL).  Replaces the same residual adds:
 * stage1/layers.py:55-56, stage2/transformer.py:45-48. */
int pmhip_gemm_hilo_center(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                           const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N, int K,
                           float* row_stats, const float* center_coef, float center_extra, float* shift, int shift_mode,
                           pmhip_stream stream);
/* (hi, lo) <- split(hi + lo + shift[row]) in place: a centred stream back to the plain pair (D <= 1024, D % 4 == 0) */
int pmhip_unshift_hilo(void* x_hi, void* x_lo, const float* shift, int M, int D, pmhip_stream stream);
/* row operators on the pair (D a multiple of 4, <= 1024): LayerNorm of hi + lo -> f32 / bf16 (the unfolded path);
 * LayerNorm of an f32 row -> hi, lo; f32 -> hi, lo and back */
int pmhip_layernorm_hilo(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, float eps,
                         void* out, int out_dtype, int M, int D, pmhip_stream stream);
int pmhip_layernorm_to_hilo(const float* x, const float* gamma, const float* beta, float eps, void* out_hi,
                            void* out_lo, int M, int D, pmhip_stream stream);
int pmhip_split_hilo(const float* x, void* out_hi, void* out_lo, int M, int D, pmhip_stream stream);
int pmhip_join_hilo(const void* x_hi, const void* x_lo, float* out, int M, int D, pmhip_stream stream);
/* coef[M][2] = (rstd, -rstd * mean) of the rows of the hi plane (two-pass, like the LayerNorm kernel) */
int pmhip_ln_coef(const void* x_hi, float eps, float* coef, int M, int D, pmhip_stream stream);
/* the same coefficients from pmhip_gemm_hilo_stats' partial statistics (D = 64 * nparts <= 1024): Chan's combination in part
 * order, deterministic; equal to pmhip_ln_coef to rounding (a few ulp of rstd) */
int pmhip_ln_coef_parts(const float* row_stats, int nparts, float eps, float* coef, int M, pmhip_stream stream);

typedef struct pmhip_lnfold {
    const float* coef;    /* [M][2]: (rstd, -rstd * mean) per row, from pmhip_ln_coef */
    const float* c;       /* [N]: sum_k of the (rounded) gamma-scaled weight row */
    const float* d;       /* [N]: sum_k beta[k] * W[n,k] */
    /* ABI 9, optional (parts NULL: coef is an input, as before).  parts = the [M][nparts][2] partial statistics
     * pmhip_gemm_hilo_stats left behind for this hi plane (nparts * 64 = K), eps the LayerNorm's: coef is then an OUTPUT of the
     * call as well -- written by the GEMM itself where a small launch computes its rows' coefficients in its prologue, by
     * pmhip_ln_coef_parts (launched by the call) otherwise; bit-identical either way. */
    const float* parts; int nparts; float eps;
} pmhip_lnfold;

/* Consumers: same arguments as pmhip_gemm (no residual) / pmhip_gemm_swiglu / pmhip_gemm_heads, with A = the hi plane,
 * W = the gamma-scaled weights and `ln` the fold descriptor.  Only shapes served by the 256x256 kernel
 * (pmhip_lnfold_supported; epi_kind 0 = plain, 1 = SwiGLU, 2 = head split; N counts the packed output rows). */
int pmhip_gemm_ln(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                  int ldo, int out_dtype, int M, int N, int K, const pmhip_lnfold* ln, pmhip_stream stream);
int pmhip_gemm_swiglu_ln(int dtype, const void* A, int lda, const void* W12p, const float* b12p, void* out,
                         int ldo, int M, int Hp, int K, const pmhip_lnfold* ln, pmhip_stream stream);
int pmhip_gemm_heads_ln(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K, int heads,
                        int tokens, int tokens_pad, int nparts, const int* part_kinds_host,
                        void* const* part_outs_host, float q_scale, const pmhip_lnfold* ln,
                        pmhip_stream stream);
int pmhip_lnfold_supported(int dtype, int epi_kind, int M, int N, int K);

/* The logits GEMM of the MaskGIT step (stage2/transformer.py:91: to_logits) with the sampler's statistics from its epilogue (round 5):
 * out (f32) [M,N] = A . W^T + bias -- A, W as for pmhip_gemm, or, with `ln` not NULL, as for pmhip_gemm_ln -- and block_stats
 * [M][N/64][2] = (max, sum_j 2^((x_j - max) log2 e)) of every 64-column block of every row, computed from the values as they are
 * stored.  pmhip_sample_rows_stats then needs 8 bytes per block and the top-k blocks of a row instead of the row.  N % 64 == 0. */
int pmhip_gemm_softmax_stats(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo,
                             int M, int N, int K, const pmhip_lnfold* ln, float* block_stats, pmhip_stream stream);

/* softmax(Q K^T) V per (batch, head), no mask, no dropout (modules/attention.py:51-58; the same
 * maths as xformers.ops.memory_efficient_attention at :100).  Q is already scaled.  Layouts as
 * written by pmhip_gemm_heads.  out[B*Nq, heads*64] (`dtype`), head-major inside a row
 * ('(b h) n d -> b n (h d)', attention.py:58).  use_exp2 != 0: Q carries an extra log2(e) factor
 * and the kernel exponentiates with exp2. */
int pmhip_attention(int dtype, const void* Q, const void* K, const void* Vt, void* out, int ldo,
                    int B, int heads, int Nq, int Nkv, int Nkv_pad, int use_exp2,
                    pmhip_stream stream);

/* Diagnostic of the bf16 kernel behind pmhip_attention (round 5).  Its fast path measures every probability against the
 * maximum of the query's first 32 keys and never looks at the running maximum again; a workgroup in which a probability
 * left the f32 range (scores more than 2^6 octaves above that reference; the softmax of modules/attention.py:53 itself
 * has no such limit) notices it when it normalises and runs again through its exact path, which raises the running
 * maximum at every half-tile.  *count = such workgroups on the current device since the last reset (synchronises the
 * device). */
int pmhip_attention_fallbacks(unsigned long long* count, int reset);

/* ---- any dim_head (modules/attention.py:27-33: inner_dim = dim_head * heads, scale = dim_head^-0.5).  The reference's
 * configs all use 64 and the tuned kernels above are built for it; these entry points forward to them when
 * dim_head == 64 and otherwise take a plain path: dim_head a multiple of 16 up to 128, heads*dim_head a multiple of 64.
 *   pmhip_gemm_heads_dh : the projection runs as an ordinary GEMM into `scratch` (f32 [M, nparts*heads*dim_head], caller
 *                         owned, unused when dim_head == 64) and a split kernel writes Q [B,H,tokens,dh] (* q_scale),
 *                         K [B,H,tokens_pad,dh] and V^T [B,H,dh,tokens_pad], rounding to `dtype` once.
 *   pmhip_attention_dh  : flash-style online softmax on the vector ALU, 4 lanes per query row. */
int pmhip_gemm_heads_dh(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K, int heads, int dim_head,
                        int tokens, int tokens_pad, int nparts, const int* part_kinds_host,
                        void* const* part_outs_host, float q_scale, float* scratch, pmhip_stream stream);
int pmhip_attention_dh(int dtype, const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads,
                       int dim_head, int Nq, int Nkv, int Nkv_pad, int use_exp2, pmhip_stream stream);

/* torch.nn.LayerNorm over the last dim, eps inside the sqrt (stage1/layers.py:49,51,89,128;
 * stage2/transformer.py:37,39,41,62).  x fp32 [M,D] -> out (`out_dtype`). */
int pmhip_layernorm(const float* x, const float* gamma, const float* beta, float eps, void* out,
                    int out_dtype, int M, int D, pmhip_stream stream);

/* Non-overlapping PxP patches of img[B,C,H,W] (fp32) as GEMM rows: out[B*(H/P)*(W/P), C*P*P],
 * column order (c, kh, kw) = the flattened Conv2d weight (stage1/layers.py:81-84,107). */
int pmhip_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int P,
                   pmhip_stream stream);

/* 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)' then clamp(lo,hi) (stage1/layers.py:150;
 * stage1/vqmodel.py:30).  y fp32 [B*(H/P)*(W/P), P*P*C] -> img fp32 [B,C,H,W]. */
int pmhip_unpatchify_clamp(const float* y, float* img, int B, int C, int H, int W, int P, float lo,
                           float hi, pmhip_stream stream);

/* fp32 [M,K] -> `out_dtype` [M,Kpad], zero-padded columns (feeds K<64 projections). */
int pmhip_convert_pad(const float* in, int K, void* out, int out_dtype, int Kpad, int M,
                      pmhip_stream stream);

/* out[m,:] = x[m,:] + table[m % table_rows,:], all fp32 (x + position_embedding,
 * stage1/layers.py:108,146; stage2/transformer.py:82). */
int pmhip_add_rows(const float* x, const float* table, int table_rows, float* out, int M, int D,
                   pmhip_stream stream);

/* out = uncond + scale * (cond - uncond) over n fp32 elements (n % 4 == 0), fmaf per element; out may alias either input.
 * Guidance between the text-conditioned and the unconditional logits of one MaskGIT step: the reference trains for it by dropping
 * the text 10 % of the time (utils/trainer.py:379,387-388: `text = None` -> attn2 becomes a second self-attention,
 * modules/attention.py:47) but its own sampling (generate.py:159-181) never combines the two; SURVEY.md section 8(f) row 2. */
int pmhip_guidance_combine(const float* cond, const float* uncond, float scale, float* out, size_t n, pmhip_stream stream);
/* The same, and block_stats [n/64][2] = the softmax statistics (max, sum of exp) of every 64-element block of the result, for
 * pmhip_sample_rows_stats: rows must be contiguous and a multiple of 64 long (n % 64 == 0). */
int pmhip_guidance_combine_stats(const float* cond, const float* uncond, float scale, float* out, size_t n, float* block_stats,
                                 pmhip_stream stream);

/* Row gather out[m,:] = table[ids[m],:] (nn.Embedding: stage1/quantize.py:41, generate.py:148-157),
 * table fp32 [V,E], ids int64, out `out_dtype` [M,Kpad] zero-padded. */
int pmhip_embed_rows(const float* table, const int64_t* ids, void* out, int out_dtype, int Kpad,
                     int M, int V, int E, pmhip_stream stream);

/* Codebook preparation, hoisted out of the per-call path: en = W / max(||W||,1e-12) row-wise and
 * sq[j] = sum(en[j]^2) (stage1/quantize.py:5-6,21,24-25). */
int pmhip_vq_prepare(const float* codebook, float* en, float* sq, int V, int E,
                     pmhip_stream stream);

/* VectorQuantizer.forward (stage1/quantize.py:18-38) on z fp32 [M,E] (the prev_quant output):
 * zn = l2norm(z); d = sum(zn^2) + sq - 2 zn.en^T; idx = first argmin; zq = en[idx];
 * z_out = zn + (zq - zn); loss = beta*mean((zq-zn)^2) + mean((zq-zn)^2).
 * The arithmetic order is fixed (sequential fmaf over E) and restated in oracle/vq_ref.c, so idx
 * is bit-exact against the oracle.  `scratch` >= pmhip_vq_scratch_bytes(M,V) bytes. */
size_t pmhip_vq_scratch_bytes(int M, int V);
int pmhip_vq_quantize(const float* z, const float* en, const float* sq, float beta, float* z_out,
                      int64_t* idx_out, float* loss_out, void* scratch, int M, int V, int E,
                      pmhip_stream stream);

/* Per-sample random masking of the latent sequence (Pipeline.random_masking, generate.py:78-110):
 * position i of sample b is KEPT iff fewer than len_keep positions of the sample have smaller noise
 * (ties: smaller index first, i.e. the order of a stable ascending argsort); masked positions get
 * mask_token.  z, x_out fp32 [B,N,E]; noise, mask_out fp32 [B,N] (mask: 0 keep, 1 masked);
 * len_keep = N - max(int(N*mask_ratio),1) is computed by the caller (generate.py:86-87). */
int pmhip_random_mask(const float* z, const float* noise, const float* mask_token, int len_keep,
                      float* x_out, float* mask_out, int B, int N, int E, pmhip_stream stream);

/* Masked label-smoothed cross entropy, forward only (Pipeline.loss, generate.py:112-125):
 * row_loss[m] = mask[m] * CE(logits[m,:], labels[m]; label_smoothing) with torch's definition
 * (1-eps)*nll + eps*mean_c(-log p_c); loss_out[0] = sum(row_loss) / sum(mask) (nan if nothing is
 * masked, as in the reference).  logits fp32 [M,V] with row stride ldl, labels int64 in [0,V). */
int pmhip_masked_ce(const float* logits, int ldl, const int64_t* labels, const float* mask,
                    float label_smoothing, float* row_loss, float* loss_out, int M, int V,
                    pmhip_stream stream);

/* One MaskGIT sampling pass over logits rows (generate.py:163-173 with helpers :29-46):
 * top-k filter, gumbel-argmax at `temperature`, confidence from the UNfiltered softmax, merge into
 * the previously masked positions.  noise: NULL -> counter-based Philox keyed by
 * (seed, step, row_base+row, column); else fp32 [M,V] uniform(0,1) samples (parity mode, the
 * reference's torch.zeros_like(t).uniform_(0,1), generate.py:41).
 * Outputs: pred[M] (all positions, what the image is decoded from, generate.py:165),
 * ids_out[M] = where(ids_in==mask_id, pred, ids_in), score[M] = is_mask ? 1-p[pred] : -1e5.
 * ids_out may alias ids_in.  Ties: (value desc, index asc). 1 <= topk <= 64. */
int pmhip_sample_rows(const float* logits, int ldl, const int64_t* ids_in, int64_t mask_id,
                      int topk, float temperature, const float* noise, uint64_t seed,
                      uint32_t step, uint64_t row_base, int64_t* pred_out, int64_t* ids_out,
                      float* score_out, int M, int V, pmhip_stream stream);

/* The same step for a caller that holds the SOFTMAX STATISTICS of the rows' 64-column blocks (round 5): block_stats [M][V/64][2] =
 * (max, sum_j 2^((x_j - max) log2 e)) per block, as pmhip_gemm_softmax_stats / pmhip_guidance_combine_stats leave them behind.
 * For topk <= 8 the kernel reads the statistics (8 bytes per block) and the topk blocks with the largest maxima -- they contain
 * the topk largest elements -- instead of the whole row: 1 KiB + topk x 256 B instead of 32 KiB at V = 8192.
 * Same result, bit for bit, as pmhip_sample_rows on the same logits: for topk <= 8 and V % 64 == 0 that entry runs the same
 * kernel and derives the statistics from the stored row with the same arithmetic.  (topk > 8: the statistics are ignored.)
 * V % 64 == 0.  Reference: generate.py:163-173, as above. */
int pmhip_sample_rows_stats(const float* logits, int ldl, const float* block_stats, const int64_t* ids_in, int64_t mask_id,
                            int topk, float temperature, const float* noise, uint64_t seed, uint32_t step,
                            uint64_t row_base, int64_t* pred_out, int64_t* ids_out, float* score_out, int M, int V,
                            pmhip_stream stream);

/* Re-mask the num_mask least confident tokens of every image (generate.py:175-179):
 * ids[b, topk(scores[b], num_mask)] = mask_id.  Ties: (score desc, index asc). N <= 4096. */
int pmhip_remask(int64_t* ids, const float* scores, int num_mask, int64_t mask_id, int B, int N,
                 pmhip_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Model level.  Weight tables are filled by the host packer (paintmind_amd/engine.py) from the
 * reference's state_dict layout.
 * ---------------------------------------------------------------------------------------------- */

typedef struct pmhip_layer_weights {
    const float* ln1_g; const float* ln1_b;    /* norm1                                         */
    const void* wqkv;                          /* attn1 to_q|to_k|to_v rows, [3*inner, dim] T   */
    const void* wo; const float* bo;           /* attn1 to_out.0  [dim, inner] T, [dim]         */
    const float* lnx_g; const float* lnx_b;    /* stage 2 only: norm2 (else NULL)               */
    const void* wqkv2;                         /* stage 2 only: attn2 to_q|to_k|to_v            */
    const void* wo2; const float* bo2;         /* stage 2 only: attn2 to_out.0                  */
    const float* ln2_g; const float* ln2_b;    /* the norm in front of the FFN                  */
    const void* w12p; const float* b12p;       /* packed SwiGLU w12, [2*hidden_pad, dim] T      */
    const void* w3p; const float* b3;          /* w3 zero-padded in K, [dim, hidden_pad] T      */
    /* LayerNorm fold (bf16 mode; all NULL = not folded): gamma-scaled weights + the c / d vectors of pmhip_lnfold  */
    const void* wqkv_f; const float* qkv_c; const float* qkv_d;        /* norm1 into attn1 q|k|v               */
    const void* wqkv2_f; const float* qkv2_c; const float* qkv2_d;     /* stage 2: norm2 into attn2 q|k|v      */
    const void* w12p_f; const float* w12_c; const float* w12_d;        /* the FFN norm into packed w12         */
    /* centred hi plane (pmhip_gemm_hilo_center): mean over the columns of the three residual-producer biases -- what a producer
     * adds to EVERY row's mean, known before the row is computed (0 when unused)                                          */
    float bo_mean, bo2_mean, b3_mean;
} pmhip_layer_weights;

typedef struct pmhip_tower_cfg {
    int dim, depth, heads, hidden_pad;
    int dim_head;                              /* 0 means 64 (the reference's configs); see pmhip_attention_dh */
} pmhip_tower_cfg;

typedef struct pmhip_vqgan_cfg {
    int image_size, patch_size, channels;
    int n_embed, embed_dim;
    float beta;
    pmhip_tower_cfg enc, dec;
} pmhip_vqgan_cfg;

typedef struct pmhip_vqgan_weights {
    const void* patch_w;                       /* conv weight as [dim, C*P*P] T                 */
    const float* enc_pos;                      /* [tokens, dim]                                 */
    const float* pre_g; const float* pre_b;    /* encoder.norm_pre                              */
    const pmhip_layer_weights* enc_layers;     /* host array [enc.depth]                        */
    const void* prevq_w; const float* prevq_b; /* [embed_dim, dim] T                            */
    const float* codebook_n;                   /* pmhip_vq_prepare output, [n_embed, embed_dim] */
    const float* codebook_sq;                  /* [n_embed]                                     */
    const void* postq_w; const float* postq_b; /* [dim, 64] T (K zero-padded)                   */
    const float* dec_pos;
    const pmhip_layer_weights* dec_layers;
    const float* dn_g; const float* dn_b;      /* decoder.norm                                  */
    const void* proj_w; const float* proj_b;   /* [P*P*C, dim] T                                */
} pmhip_vqgan_weights;

typedef struct pmhip_vqgan pmhip_vqgan;

/* VQModel (stage1/vqmodel.py:7-44).  The tables are copied; the pointed-to weights are not. */
int pmhip_vqgan_create(pmhip_vqgan** out, int device, int dtype, const pmhip_vqgan_cfg* cfg,
                       const pmhip_vqgan_weights* w);
void pmhip_vqgan_destroy(pmhip_vqgan* h);
/* VQModel.encode (vqmodel.py:21-25): img fp32 [B,C,H,W] -> z fp32 [B,N,E], idx int64 [B,N], loss[1] */
int pmhip_vqgan_encode(pmhip_vqgan* h, const float* img, int B, float* z_out, int64_t* idx_out,
                       float* loss_out, pmhip_stream stream);
/* VQModel.decode (vqmodel.py:27-30): z fp32 [B,N,E] -> img fp32 [B,C,H,W] clamped to [-1,1] */
int pmhip_vqgan_decode(pmhip_vqgan* h, const float* z, int B, float* img_out, pmhip_stream stream);
/* VQModel.decode_from_indice (vqmodel.py:38-41) */
int pmhip_vqgan_decode_indices(pmhip_vqgan* h, const int64_t* idx, int B, float* img_out,
                               pmhip_stream stream);
/* Encoder.forward / Decoder.forward alone (stage1/layers.py:106-112,145-152); the decoder variant
 * takes the post_quant output x fp32 [B,N,dim] and returns the un-clamped image. */
int pmhip_vqgan_encoder_forward(pmhip_vqgan* h, const float* img, int B, float* x_out,
                                pmhip_stream stream);
int pmhip_vqgan_decoder_forward(pmhip_vqgan* h, const float* x, int B, float* img_out,
                                pmhip_stream stream);

typedef struct pmhip_s2_cfg {
    int tokens, embed_dim, n_embed, context_dim, context_dim_pad;  /* context_dim_pad: mult. of 64 */
    pmhip_tower_cfg tower;
} pmhip_s2_cfg;

typedef struct pmhip_s2_weights {
    const float* tok_table;                    /* [n_embed+1, embed_dim]: RAW codebook rows, then
                                                  mask_token (generate.py:148-157)              */
    const void* tokproj_w; const float* tokproj_b;   /* [dim, 64] T                            */
    const float* pos;                          /* [tokens, dim]                                 */
    const void* ctxproj_w;                     /* [dim, context_dim_pad] T, NULL when Identity  */
    const pmhip_layer_weights* layers;         /* host array [depth]                            */
    const float* norm_g; const float* norm_b;
    const void* logits_w; const float* logits_b;     /* [n_embed, dim] T                       */
    const void* logits_wf; const float* logits_c; const float* logits_d;   /* final norm folded into to_logits (or NULL) */
} pmhip_s2_weights;

typedef struct pmhip_s2 pmhip_s2;

/* CondTransformer (stage2/transformer.py:52-93) */
int pmhip_s2_create(pmhip_s2** out, int device, int dtype, const pmhip_s2_cfg* cfg,
                    const pmhip_s2_weights* w);
void pmhip_s2_destroy(pmhip_s2* h);
/* CondTransformer.forward (transformer.py:80-93): tokens fp32 [B,N,E]; context fp32 [B,L,ctx] or
 * NULL (then attn2 is a second self-attention, modules/attention.py:47); logits fp32 [B,N,V]. */
int pmhip_s2_forward(pmhip_s2* h, const float* tokens, const float* context, int L, int B,
                     float* logits_out, pmhip_stream stream);

/* Pipeline.sample (generate.py:159-181), one MaskGIT step on ids int64 [B,N] in place.
 * img_out may be NULL (skip the ViT decode of generate.py:165); pred_out/score_out may be NULL. */
int pmhip_pipeline_sample(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L,
                          int B, int topk, float temperature, int num_mask, const float* noise,
                          uint64_t seed, uint32_t step, uint64_t image_base, float* img_out,
                          int64_t* pred_out, float* score_out, pmhip_stream stream);

/* Pipeline.generate's loop body (generate.py:189-196) for T steps.  temps_host[T], nmask_host[T]
 * are the per-step temperature and num_token_masked the caller derived exactly as the reference
 * does (generate.py:191-193,175); decode_host[T] != 0 selects the steps whose image is produced,
 * written consecutively into imgs_out [n_decoded, B, C, H, W] (device; may be NULL when imgs_host
 * is given).  The context projection and the cross-attention K/V of the static context are
 * computed once (the reference recomputes them every step, transformer.py:84-85).
 * use_graph: bit flags.  PMHIP_GENERATE_GRAPH (1): the loop is a chain of hipGraphs, one per segment ending in a decoded step.
 * The request is IGNORED (eager loop, same bits) while per-kernel timing is on, for T > 64, and when the process runs with
 * AMD_DIRECT_DISPATCH=0, where graph replay is broken on ROCm 7.2 (tools/hwtests/graph_dispatch_mode.hip); bit 5 of
 * pmhip_s2_switches tells a caller that this process is in that mode.
 * PMHIP_GENERATE_CONCURRENT_LANES (2): the caller runs other micro-batches on other streams at the same time; the loop then never
 * defers a step's ViT decode to a side stream.  The deferral exists on the GRAPH path only (fork / join inside the captured
 * segments, for B * tokens <= 65536: one lane alone leaves the chip partly idle: +15 % at B = 8, +1.5 % at B = 64); the eager
 * loop decodes in line -- results are the same either way.
 * imgs_host != NULL replaces the reference's `imgs.append(img.cpu())` (generate.py:195-196): decoded
 * image d is copied to imgs_host + d * host_stride (floats; the caller's PINNED buffer, so that a
 * lane can fill its rows of a [n_decoded, B_total, C, H, W] tensor) on copy_stream as soon as it is
 * complete, under the following steps.  There is no cross-stream wait on the device (a barrier parked
 * in the copy queue starves concurrent lanes): the CALL paces itself instead -- with imgs_host it
 * blocks the calling thread between segments (hipEventSynchronize on the image just finished, one
 * segment always queued ahead) and returns once the last copy is enqueued; concurrent lanes are
 * driven from one thread each.  The caller synchronises copy_stream before reading the host buffer.
 * copy_stream NULL: the copies are enqueued on `stream`. */
#define PMHIP_GENERATE_GRAPH 1
#define PMHIP_GENERATE_CONCURRENT_LANES 2
int pmhip_pipeline_generate(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context,
                            int L, int B, int T, const float* temps_host, const int* nmask_host,
                            const unsigned char* decode_host, int topk, uint64_t seed,
                            uint64_t image_base, float* imgs_out, int use_graph,
                            pmhip_stream stream, float* imgs_host, size_t host_stride,
                            pmhip_stream copy_stream);

/* The same two entry points with GUIDANCE (round 5; an extension behind an explicit argument):
 * every step runs the tower twice on the step's tokens -- with the context, and as the unconditional branch the reference
 * trains by dropping the text 10 % of the time (utils/trainer.py:379,387-388: context None, attn2 a second self-attention,
 * modules/attention.py:47) -- and samples from uncond + guidance_scale * (cond - uncond) (pmhip_guidance_combine); everything
 * after the logits is the reference's step (generate.py:161-179).  context must not be NULL.  The loop is graph-captured and
 * lane-able like pmhip_pipeline_generate (one executable graph per guidance_scale value); scale 0 reproduces the unconditional
 * step bit for bit, and the result equals the operator-level composition pmhip_s2_forward x 2 + pmhip_guidance_combine +
 * sampling bit for bit (tests/test_gpu_model.py). */
int pmhip_pipeline_sample_guided(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L,
                                 int B, int topk, float temperature, int num_mask, const float* noise,
                                 uint64_t seed, uint32_t step, uint64_t image_base, float* img_out,
                                 int64_t* pred_out, float* score_out, float guidance_scale,
                                 pmhip_stream stream);
int pmhip_pipeline_generate_guided(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context,
                                   int L, int B, int T, const float* temps_host, const int* nmask_host,
                                   const unsigned char* decode_host, int topk, uint64_t seed,
                                   uint64_t image_base, float* imgs_out, int use_graph,
                                   pmhip_stream stream, float* imgs_host, size_t host_stride,
                                   pmhip_stream copy_stream, float guidance_scale);

/* The PMHIP_* development switches are read from the environment when a handle is CREATED and stay with it (its workspace,
 * fold decisions and captured graphs depend on them); editing the environment of a live handle does nothing.  These return
 * what a handle latched: bit 0 LayerNorm fold (PMHIP_LN_UNFOLD unset), bit 1 bf16 hi/lo stream (PMHIP_HILO != 0), bit 2 row
 * statistics from the producers (PMHIP_LN_STATS != 0), bit 3 centred hi plane (PMHIP_HILO_CENTER != 0), bit 4
 * PMHIP_BLOCKING_WAIT, bit 5 (pmhip_s2_switches only; process-wide) AMD_DIRECT_DISPATCH=0: PMHIP_GENERATE_GRAPH requests run
 * the eager loop; -1 for a NULL handle.  A tool that A/Bs a switch asserts the mode it believes it measures. */
int pmhip_s2_switches(const pmhip_s2* h);
int pmhip_vqgan_switches(const pmhip_vqgan* h);

/* Per-kernel timing hook used by bench.py: when enabled, every kernel launch of the named family
 * is bracketed by hipEvents on its own stream and accumulated (count, total ms). */
int pmhip_timing_enable(int on);
int pmhip_timing_reset(void);
/* family: "gemm" (all GEMM launches = the sum of "gemm_plain", "gemm_heads", "gemm_swiglu", "gemm_resid", "gemm_resid2b": bias-only /
 * head-split q|k|v / SwiGLU w12 / residual producers on the one-workgroup kernels / on the two-workgroups-per-CU kernel),
 * "attention", "layernorm", "sample", "vq", "rowops"; returns PMHIP_EINVAL if unknown.  The accumulators are process-wide and
 * guarded by a mutex; while timing is on pmhip_pipeline_generate runs eagerly (no graph replay). */
int pmhip_timing_get(const char* family, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* PMHIP_H */
