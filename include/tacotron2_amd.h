/*
 * tacotron2_amd.h — C ABI of the MI355X (gfx950) Tacotron 2 mel-spectrogram engine.
 *
 * The reference (NVIDIA/tacotron2) has no FFI/plugin interface: its hot path is Python
 * (`model.py`) over PyTorch/cuDNN/NCCL.  This header is the boundary the
 * reference *would* bind if its `model.py` delegated to a native library: every entry
 * point below names the reference lines whose arithmetic it replaces.  The Python host
 * (`tacotron2_amd/model.py`, same class/method/state_dict surface as reference
 * model.py:457-529) binds these symbols with ctypes; a maintainer of the reference would
 * add exactly the same ctypes stubs.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types.  All tensors are fp32 row-major
 *     in HBM unless stated; `ld*` are row strides in ELEMENTS; masks are uint8 keep-masks
 *     (1 = keep) applied as x * keep * keep_scale (keep_scale = 1/(1-p) formed in fp32).
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream) and returns immediately: T2AMD_OK or an error code; the message is
 *     available from t2amd_last_error().  Nothing allocates or frees device memory: the
 *     caller owns all buffers, including workspaces (graph-capture safe).
 *   - time-major slabs: [T][B][F]; batch-major: [B][T][F].
 */
#ifndef TACOTRON2_AMD_H
#define TACOTRON2_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2AMD_ABI_VERSION 1
#define T2AMD_OK 0
#define T2AMD_ERR_ARG 1
#define T2AMD_ERR_LAUNCH 2

/* attention geometry compiled into the kernels (reference hparams.py:66-70 defaults) */
#define T2AMD_ATT_DIM 128
#define T2AMD_LOC_FILTERS 32
#define T2AMD_LOC_KERNEL 31
#define T2AMD_LOC_TAPS (2 * T2AMD_LOC_KERNEL) /* 62 = [prev ; cumulative] x 31 */
#define T2AMD_ATT_SLICES 4 /* workgroups per utterance in every attention kernel; partial slabs have this many slices */

int t2amd_abi_version(void);
/* SHA-1 (40 hex digits) over csrc/ and this header, compiled in by tacotron2_amd/build.py: the binding refuses a library
 * that was not built from the sources beside it, and bench.py prints it next to the hash recorded with the PMC passes. */
const char* t2amd_source_sha1(void);
const char* t2amd_last_error(void);
/* sizeof() of every struct below, in declaration order, for binding self-checks. */
int t2amd_struct_sizes(int* out, int max_n);
/* Validate-only mode: all argument checks and host loops run, no kernel is launched (outputs are
 * left untouched).  For CPU-side tests of the binding; not a compute path. */
int t2amd_set_validate_only(int on);
/* Live kernel timing for bench.py's roofline: while enabled, every LSTM-step launch whose tag equals
 * `tag` is bracketed by two hipEvents on its own stream (at most `max_launches` launches).
 * t2amd_profile_read synchronises those events and returns the summed elapsed time and the count. */
int t2amd_profile_enable(int tag, int max_launches);
int t2amd_profile_read(float* total_ms, int* count);
/* Elapsed time of an EMPTY bracket (two event records back to back on the launch stream), measured at the start
 * of the profiling session: what a bracket costs besides the kernel.  Call before t2amd_profile_read. */
int t2amd_profile_event_overhead(float* ms);

/* ------------------------------------------------------------------------------------
 * Dense / implicit-convolution GEMM on exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
 *   C[M,N] (+)= act(A[M,K] . B[K,N] + bias[N]) * keep * keep_scale
 * replaces every torch.nn.Linear / Conv1d forward, dgrad and wgrad on the path:
 * reference layers.py:17-18,37-39 (LinearNorm/ConvNorm.forward), model.py:99 (Prenet),
 * :141-146 (Postnet convs), :174-175 (Encoder convs), :288 (memory_layer), :375-378
 * (linear_projection + gate_layer) and autograd's matching backward GEMMs.
 * ------------------------------------------------------------------------------------ */
typedef struct t2amd_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long long lda, ldb, ldc;
    int a_kcontig;   /* 1: A stored [M][K] (K contiguous); 0: A stored [K][M] */
    int b_kcontig;   /* 1: B stored [N][K] (K contiguous, the nn.Linear weight layout); 0: [K][N] */
    int batch;       /* >=1 independent problems, pointers advance by stride* */
    long long strideA, strideB, strideC;
    int splitk;      /* >=1; when >1 the kernel writes `splitk` partial C's (stride strideSplitC)
                        with a plain store and the epilogue fields must be unset */
    long long strideSplitC;
    int accumulate;  /* 1: C += result */
    const float* bias;      /* [N] or NULL */
    int act;                /* 0 none, 1 relu */
    const uint8_t* keep;    /* [M][ldkeep] or NULL */
    long long ldkeep;
    float keep_scale;
    /* implicit 1-D convolution over channel-last rows r = b*T + t (stride 1, 'same' zero pad):
     * A-side (needs a_kcontig=1): K = taps*convA_C, column (tap,c) of row r reads
     *   A[(r + (tap-pad)*sign)*lda + c] if 0 <= t + (tap-pad)*sign < T else 0.
     * B-side (needs b_kcontig=0, used by wgrad): N = taps*convB_C, column (tap,c) of K-row r reads
     *   B[(r + tap - pad)*ldb + c] if 0 <= t + tap - pad < T else 0. */
    int convA_T, convA_C, convA_pad, convA_sign;
    int convB_T, convB_C, convB_pad;
    /* 0: exact f32 MFMA (bitwise an fmaf chain).  1: split-bf16: A = Ah+Al, B = Bh+Bl in bf16, product =
     * Ah.Bh + Ah.Bl + Al.Bh on the bf16 MFMA with f32 accumulation (~2^-17 relative per product, 5.3x the
     * f32 MFMA ceiling).  The engine requests 1 for gradient GEMMs in f32 mode.  2: plain bf16 product
     * (operands rounded to bf16, f32 accumulation) — the engine's bf16 compute mode. */
    int precision;
} t2amd_gemm_desc;

/* Output tile edge (128 or 256) t2amd_gemm_f32 will use for an M x N product at `precision` with the given
 * operand layouts, launched as `nz` = batch*splitk slices: callers that split K pick the split count from the resulting workgroup count.
 * (Host-side sizing helper; the reference has no counterpart, torch.mm hides its tiling.) */
int t2amd_gemm_tile_size(int M, int N, int precision, int nz, int a_kcontig, int b_kcontig);
int t2amd_gemm_f32(const t2amd_gemm_desc* d, void* stream);

/* out[i] (+)= sum_s partials[s*stride + i]; with `perm_taps`>0 the flat index i = (co, tap, ci)
 * is written to (co, ci, tap) (packed conv-weight grad -> torch Conv1d layout, Ci = perm_ci). */
int t2amd_splitk_reduce_f32(const float* partials, int nsplit, long long stride, float* out,
                            long long n, int accumulate, int perm_taps, int perm_ci, void* stream);
/* out[r][c] (row stride ldo >= cols) (+)= sum_s partials[s*stride + r*cols + c]: partials of a product that fills a column
 * block of a wider matrix. */
int t2amd_splitk_reduce2d_f32(const float* partials, int nsplit, long long stride, float* out, int rows, int cols,
                              long long ldo, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * bf16-resident product (csrc/gemm16.hip): C[M][N] (f32) = A[M][K] . B[N][K]^T (+ bias[n]), A and B bf16 and
 * K-contiguous (row strides lda / ldb in elements, multiples of 8, >= K), K a multiple of 64.  In the bf16 compute mode
 * it carries the products whose operands exist as bf16 images: the deferred weight gradients dW = dG^T . X of the
 * decoder LSTMs and of the hoisted input projection (reference model.py:352-371 under autograd) on K-contiguous images
 * made by t2amd_transpose_cast_bf16.  splitk > 1 writes partial slabs strideSplitC apart (plain epilogue only).
 * ------------------------------------------------------------------------------------ */
typedef struct t2amd_gemm16_desc {
    const void* A;          /* [M][lda] bf16 */
    const void* B;          /* [N][ldb] bf16 */
    float* C;               /* [splitk][M][ldc] f32 */
    int M, N, K;
    long long lda, ldb, ldc;
    int splitk;
    long long strideSplitC;
    int accumulate;         /* C += (splitk must be 1) */
    const float* bias;      /* [N] or NULL (splitk must be 1) */
    /* window mode (win_Tp > 0; nn.Conv1d over channel-last rows, reference layers.py:37-39 / model.py:141-146, 174-175):
     * A is a bf16 image [B][Tp = T + 2 pad][Ci] with zero halo rows (t2amd_cast_halo_bf16), lda = Ci, K = k Ci rounded up to 64
     * (B's columns behind k Ci zero, t2amd_pack_conv_bf16; the windows' overhang stays inside the image's last halo) -- row
     * m = b Tp + t of A is the k-tap window of output (b, t), overlapping its neighbours; M = B Tp window rows are
     * multiplied and those with t < win_T are stored to the compact rows b win_T + t of C.  splitk must be 1. */
    int win_T, win_Tp;
} t2amd_gemm16_desc;
int t2amd_gemm16_tn(const t2amd_gemm16_desc* d, void* stream);
/* The same product with K-MAJOR operands: C[M][N] = sum_k A[k][m] . B[k][n], A = [K][lda], B = [K][ldb] bf16 with m / n
 * contiguous -- the layout of the time loops' [To.B][.] slabs and of channel-last activation images, so the weight gradients
 * dW = dG^T . X of nn.LSTMCell (reference model.py:352-371 under autograd) and nn.Conv1d (layers.py:37-39) need no
 * transposed copies; fragments are formed by gfx950's transposing LDS read.  M, N multiples of 8; lda / ldb multiples of 8
 * and possibly SMALLER than M / N (overlapping rows: B[k][n] = img[k Ci + n], n < taps Ci, is the sliding window of a
 * convolution over a channel-last image); any K > 0; win_T = win_Tp = 0; bias / accumulate / splitk as above. */
int t2amd_gemm16_kk(const t2amd_gemm16_desc* d, void* stream);
/* Up to four such products in ONE launch (same M, same splitk; each with its own A offset / K / B / C): their workgroups run
 * side by side on the same rows of A, which is then read once -- the input blocks [x | ctx | h] of an LSTM's weight gradient. */
int t2amd_gemm16_kk_group(const t2amd_gemm16_desc* d, int count, void* stream);

/* dst[(b (T + 2 pad) + pad + t)][c] (bf16) = src[(b T + t)][c]; dst ([B (T + 2 pad) + 2 pad][C]; every row is written, the halo
 * rows with zeros: it need not be initialised) is the image the window mode reads. */
int t2amd_cast_halo_bf16(const float* src, long long lds, void* dst, long long rows, int C, int T, int pad, void* stream);

/* bf16 weight image of nn.Conv1d (W f32 [Co][Ci][k], reference layers.py:37-39) for the window mode, Kp >= row length, the
 * columns behind it zero: reversed = 0 -> out[Co][Kp], out[co][tap Ci + ci] = W[co][ci][tap] (forward);
 * reversed = 1 -> out[Ci][Kp], out[ci][(k - 1 - tap) Co + co] = W[co][ci][tap] (data gradient: windows of g's image). */
int t2amd_pack_conv_bf16(const float* W, void* out, int Co, int Ci, int k, int Kp, int reversed, void* stream);

/* dst[c][r] (bf16, row stride ldd >= rows_padded; columns rows..rows_padded-1 zeroed) = src[r][c]; src is f32
 * (src_is_bf16 = 0) or bf16, row stride lds elements: the K-contiguous image of a [rows][cols] slab. */
int t2amd_transpose_cast_bf16(const void* src, int src_is_bf16, long long lds, void* dst, long long ldd, int rows, int cols,
                              int rows_padded, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm1d (+activation +dropout) over channel-last rows; replaces nn.BatchNorm1d +
 * relu/tanh + F.dropout in reference model.py:141-146, 174-175 (train: biased batch
 * variance for normalisation, unbiased for the running update, momentum 0.1, eps 1e-5).
 * ------------------------------------------------------------------------------------ */
/* column statistics of x[M][N] -> mean, invstd (=1/sqrt(var_biased+eps)); updates running
 * stats if non-NULL.  ws: >= 2*64*N doubles. */
int t2amd_bn_stats_f32(const float* x, long long ldx, int M, int N, double* ws, float* mean,
                       float* invstd, float* running_mean, float* running_var, float momentum,
                       float eps, void* stream);
/* invstd from running variance (eval mode): invstd = 1/sqrt(var+eps) */
int t2amd_bn_eval_invstd_f32(const float* running_var, float* invstd, int N, float eps, void* stream);
/* y = act((x-mean)*invstd*gamma+beta) * keep*scale ; act: 0 none, 1 relu, 2 tanh.
 * `row_valid_T`>0 with `lens`!=NULL zeroes rows whose t >= lens[b] (batched ragged inference). */
int t2amd_bn_act_fwd_f32(const float* x, long long ldx, float* y, long long ldy, int M, int N,
                         const float* mean, const float* invstd, const float* gamma,
                         const float* beta, int act, const uint8_t* keep, long long ldkeep,
                         float keep_scale, const int* lens, int row_valid_T, void* stream);
/* The same, with y leaving ALSO as the bf16 halo image the next convolution's window product reads (round 6; y_img16:
 * [M/T (T + 2 pad) + 2 pad][N], t2amd_cast_halo_bf16's layout, halo rows zeroed here; same bits as the cast pass over y).
 * Needs N % 4 == 0, 16-byte-aligned rows, M a multiple of T, T >= 2 pad. */
int t2amd_bn_act_fwd_img_f32(const float* x, long long ldx, float* y, long long ldy, int M, int N, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, int act, const uint8_t* keep,
                             long long ldkeep, float keep_scale, void* y_img16, int T, int pad, void* stream);
/* backward of the above (train mode).  dy is overwritten with dx (grad wrt the conv output x).
 * y = forward output (post activation/dropout).  dgamma/dbeta are written (not accumulated).
 * ws: >= 2*64*N doubles. */
int t2amd_bn_act_bwd_f32(float* dy, long long lddy, const float* y, long long ldy, const float* x,
                         long long ldx, int M, int N, const float* mean, const float* invstd,
                         const float* gamma, int act, const uint8_t* keep, long long ldkeep,
                         float keep_scale, double* ws, float* dgamma, float* dbeta, void* stream);
/* The same backward with its two followers folded into stage 2 (round 6; the bf16 mode's convolution backward, reference
 * layers.py:37-39 + model.py:141-146 under autograd): dx leaves as the bf16 halo image the window products read
 * (dx_img16: [M/T (T + 2 pad) + 2 pad][N], t2amd_cast_halo_bf16's layout, halo rows zeroed here) and as its column sums
 * (dbias[N]: the convolution's bias gradient, bit-identical to t2amd_colsum_f32 over dx) -- one pass instead of three over the
 * slab.  keep_f32 = 0: dy is left holding stage 1's intermediate (no f32 dx is written); 1: dy = dx as above.
 * Needs N % 4 == 0, 16-byte-aligned rows, M a multiple of T, T >= 2 pad. */
int t2amd_bn_act_bwd_img_f32(float* dy, long long lddy, const float* y, long long ldy, const float* x, long long ldx, int M, int N,
                             const float* mean, const float* invstd, const float* gamma, int act, const uint8_t* keep,
                             long long ldkeep, float keep_scale, double* ws, float* dgamma, float* dbeta, void* dx_img16, int T,
                             int pad, float* dbias, int keep_f32, void* stream);
/* column sums: out[N] (+)= sum_m x[m][n]  (bias gradients). ws >= 64*N doubles */
int t2amd_colsum_f32(const float* x, long long ldx, int M, int N, double* ws, float* out,
                     int accumulate, void* stream);
/* The same sums over a bf16 slab x16[M][ldx] (round 6: the LSTM bias gradients of the bf16 mode, reference model.py:352-371 under
 * autograd, from the gate-gradient slabs the weight-gradient products read).  N, ldx multiples of 8; sums in double. */
int t2amd_colsum_bf16(const void* x16, long long ldx, int M, int N, double* ws, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * small data-movement kernels (reference model.py:503 embedding, :296-309 / :326-336
 * parse_decoder_inputs/outputs, :487-497 parse_output).
 * ------------------------------------------------------------------------------------ */
int t2amd_embedding_fwd_f32(const long long* ids, const float* table, float* out, long long rows,
                            int dim, int n_symbols, void* stream);
/* dtable[s] = sum of the dout rows whose id is s, in a fixed order (eight interleaved row lanes, no atomics).  ws is unused (may be
 * NULL); it stays in the signature for callers built against the earlier partial-sum version. */
int t2amd_embedding_bwd_f32(const long long* ids, const float* dout, float* dtable, float* ws, long long rows,
                            int dim, int n_symbols, void* stream);
/* Philox4x32-10 keep-mask: out[i] = uniform(seed, offset+i) >= p */
int t2amd_philox_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed,
                           unsigned long long offset, void* stream);
int t2amd_fill_f32(float* p, long long n, float v, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even); n % 4 == 0 */
int t2amd_cast_bf16_f32(const float* src, void* dst, long long n, void* stream);
/* Split-bf16 operand image (round 6, the 'bf16x3' mode of the LSTM tiles: t2amd_lstm_step.bf16 == 3).  src is f32 [rows][K]
 * (row stride lds floats), K % 16 == 0; dst takes 4 bytes per k (row stride ldd in k, ldd >= K): for every group of 16
 * consecutive k, 16 bf16 `hi = bf16(x)` followed by 16 bf16 `lo = bf16(x - hi)` (both round to nearest even), so that one
 * 64-byte group of a row feeds the 32x32x16 bf16 MFMA its hi and its lo fragment of the same 16 k.  x = hi + lo to ~2^-17. */
int t2amd_split_bf16x3_f32(const float* src, long long lds, void* dst, long long ldd, long long rows, int K, void* stream);
/* dst[r][c] = src[r][c] (+ src2[r][c] if src2) for r<rows, c<cols */
int t2amd_copy2d_f32(const float* src, long long lds, const float* src2, long long lds2, float* dst,
                     long long ldd, int rows, int cols, void* stream);
/* dst[c][r] = src[r][c], batched */
int t2amd_transpose_f32(const float* src, long long lds, float* dst, long long ldd, int rows,
                        int cols, int batch, long long sstride, long long dstride, void* stream);
/* teacher frames: mels [B][C][To] -> X0 [To][B][C] with X0[0]=0, X0[t]=mels[:,:,t-1] */
int t2amd_frames_to_time_major_f32(const float* mels, float* x0, int B, int C, int To, void* stream);
/* projection slab PG [To][B][C+1] -> mel_cl [B][To][C] (unmasked), gate [B][To] (1e3 where
 * t >= out_lens[b], if out_lens) */
int t2amd_split_projection_f32(const float* pg, float* mel_cl, float* gate, const int* out_lens,
                               int B, int C, int To, void* stream);
/* mel_cl,[post_cl] [B][To][C] -> mel,[mel_post = mel+post] [B][C][To], zeroed where t >= out_lens[b];
 * mel_cl itself is zeroed in place at padded frames (reference model.py:493: the in-place fill
 * reaches the tensor the first Postnet conv saved for backward). */
int t2amd_finalize_outputs_f32(float* mel_cl, const float* post_cl, float* mel, float* mel_post,
                               const int* out_lens, int B, int C, int To, void* stream);
/* backward entry: dmel,dmel_post [B][C][To] (either may be NULL) -> dpost_cl [B][To][C] = dmel_post^T,
 * dmel_cl [B][To][C] = dmel^T + dmel_post^T */
int t2amd_grads_to_channel_last_f32(const float* dmel, const float* dmel_post, float* dmel_cl,
                                    float* dpost_cl, int B, int C, int To, void* stream);
/* D_out [To][B][C+1] = [ dmel_cl[b][t][:] | dgate[b][t] ] */
int t2amd_gather_dout_f32(const float* dmel_cl, const float* dgate, float* dout, int B, int C,
                          int To, void* stream);
/* Prenet backward, elementwise part (reference model.py:99 F.dropout(F.relu(.)) under autograd):
 * dy <- (y > 0) ? dy*scale : 0, y being the forward output relu(pre)*keep*scale. */
int t2amd_relu_dropout_bwd_f32(float* dy, const float* y, float scale, long long n, void* stream);
/* alignments slab [B][To][Ti] passthrough needs no kernel. */

/* ------------------------------------------------------------------------------------
 * Recurrent cell kernels: batch x hidden "skinny" GEMM on v_mfma_f32_16x16x4_f32 with the
 * LSTM cell fused in the epilogue.  Replace torch.nn.LSTMCell / nn.LSTM + F.dropout at
 * reference model.py:352-356 (attention_rnn), :366-371 (decoder_rnn), :181-188 (encoder
 * bi-LSTM, packed-sequence semantics through `lens`/`t`).
 * ------------------------------------------------------------------------------------ */
typedef struct t2amd_seg {
    const float* p;  /* [B][width] rows, NULL = all zeros */
    long long ld;
    int width;       /* multiple of 64 */
} t2amd_seg;

typedef struct t2amd_lstm_step {
    t2amd_seg x[3];
    int nseg;
    const float* W;     /* [4H][Ktot], K contiguous, Ktot = sum widths; row g*H+j = gate g unit j */
    int Ktot, H, B;
    const float* gin;   /* [B][4H] pre-activation addend or NULL */
    long long ld_gin;
    const float* bias;  /* [4H] or NULL */
    const float* c_prev; /* [B][H] or NULL (zeros) */
    long long ld_cprev;
    float* gates_out;   /* [B][4H] activated gates i,f,g,o (may alias gin); NULL (round 6): not stored -- only a backward reads them */
    long long ld_gates;
    float* c_out;
    long long ld_c;
    float* h_out;       /* dropped-out hidden state */
    long long ld_h;
    const uint8_t* keep; /* [B][H] or NULL */
    long long ld_keep;
    float keep_scale;
    const int* lens;    /* NULL, or per-row valid length: rows with t >= lens[b] write h=c=gates=0 */
    int t;
    int tag;            /* kernel-symbol / profiling role: 0 generic, 1 attention LSTM, 2 decoder LSTM, 3 fused pair */
    /* bf16 operand mode: x[i].p and W point at bf16 (widths / ld / Ktot count ELEMENTS, widths multiples of 128);
     * the product runs on v_mfma_f32_16x16x32_bf16 with f32 accumulation; gin, bias, cell state and all f32
     * outputs are unchanged.  h16_out (optional) receives a bf16 copy of h, the next step's operand.
     * t2amd_lstm_step_small_f32 accepts 0 or 2: 2 = W alone is bf16, x[i].p stay f32 (matrix-vector path).
     * 3 (round 6, the engine's 'bf16x3' mode): x[i].p and W point at SPLIT-bf16 images (t2amd_split_bf16x3_f32: 4 bytes per k,
     * every 16 k as 16 hi then 16 lo bf16); widths / ld / Ktot count k as in the f32 form (widths multiples of 64); the
     * product is Xh.Wh + Xl.Wh + Xh.Wl on v_mfma_f32_32x32x16_bf16 with f32 accumulation (~2^-17 relative per product:
     * f32-class results at the f32 byte stream and 3/16 of the exact-f32 MFMA time); h16_out receives the split image of h
     * (ld_h16 in k). */
    int bf16;
    void* h16_out;
    long long ld_h16;
} t2amd_lstm_step;

int t2amd_lstm_step_fwd_f32(const t2amd_lstm_step* a, void* stream);
/* Two independent steps in ONE launch (b may be NULL): the time loops pair the decoder LSTM of step
 * t-1 with the attention LSTM of step t, neither depends on the other (reference model.py:352-371). */
int t2amd_lstm_step_fwd2_f32(const t2amd_lstm_step* a, const t2amd_lstm_step* b, void* stream);

/* Small-batch (B <= 8) variants for free-running decode (reference model.py:418-454 at B = 1, BASELINE
 * config 4): matrix-vector kernels bound by the weight stream instead of 64-row MFMA tiles. */
int t2amd_lstm_step_small_f32(const t2amd_lstm_step* a, void* stream);
typedef struct t2amd_small_linear {
    const float* X;      /* [B][ldx] */
    long long ldx;
    const float* W;      /* [N][ldw], K-contiguous (nn.Linear layout), 16-byte aligned rows */
    long long ldw;
    const float* bias;   /* [N] or NULL */
    float* Y;            /* [B][ldy] */
    long long ldy;
    int B, N, K;         /* B <= 8, K % 4 == 0 */
    int act;             /* 0 none, 1 relu */
    const uint8_t* keep; /* [B][ldkeep] or NULL */
    long long ldkeep;
    float keep_scale;
} t2amd_small_linear;
/* Y = act(X . W^T + bias) * keep * keep_scale   (Prenet linears model.py:99, projection + gate :373-378) */
int t2amd_linear_small_f32(const t2amd_small_linear* a, void* stream);

/* Y[s][B][N] = X[B][K-range s] . W[N][K]^T  (W K-contiguous), s < nsplit */
typedef struct t2amd_skinny_gemm {
    t2amd_seg x[3];
    int nseg;
    const float* W;   /* [N][Ktot] */
    int Ktot, N, B;
    float* Y;         /* [nsplit][B][ldy] */
    long long ldy;
    int nsplit;
    long long split_stride;
    int tag;          /* kernel-symbol role, as in t2amd_lstm_step */
    int bf16;         /* 1: x[i].p and W are bf16 (see t2amd_lstm_step) */
    /* optional epilogue (nsplit must be 1 when any of it is used): Y = keep ? act(Y + bias) * keep_scale : 0 --
     * nn.Linear + F.relu + F.dropout of the per-step prenet / projection at decode batch sizes 9..
     * (reference model.py:99, 373-378) */
    const float* bias;        /* [N] or NULL */
    int act;                  /* 0 none, 1 relu */
    const uint8_t* keep;      /* [B][ld_keep] keep mask or NULL */
    long long ld_keep;
    float keep_scale;
    void* Y16;                /* optional bf16 copy of Y (first split only), row stride ldy16 */
    long long ldy16;
    /* optional stop test of free-running decoding (reference model.py:439-444; nsplit must be 1), run by the thread that
     * holds the final value of column stop_col (the gate logit) of a row with stop_active[row] != 0:
     * sigmoid(Y) > stop_threshold (strict) or stop_t + 1 >= stop_max_steps  ->  stop_lengths[row] = stop_t + 1,
     * stop_active[row] = 0, ++*stop_done.  stop_active == NULL: no test. */
    uint8_t* stop_active;
    int* stop_lengths;
    int* stop_done;
    int stop_col, stop_t, stop_max_steps;
    float stop_threshold;
} t2amd_skinny_gemm;

int t2amd_skinny_gemm_f32(const t2amd_skinny_gemm* a, void* stream);
int t2amd_skinny_gemm2_f32(const t2amd_skinny_gemm* a, const t2amd_skinny_gemm* b, void* stream);

typedef struct t2amd_addend {
    const float* p;   /* NULL = absent */
    long long ld;
    int nsplit;       /* >=1 partial slabs summed */
    long long split_stride;
} t2amd_addend;

typedef struct t2amd_lstm_bwd {
    int B, H;
    t2amd_addend dh[3];   /* gradient wrt the dropped-out h, summed */
    const float* gates;   /* [B][4H] activated */
    long long ld_gates;
    const float* c_prev;  /* NULL = zeros */
    long long ld_cprev;
    const float* c;       /* [B][H] */
    long long ld_c;
    const uint8_t* keep;
    long long ld_keep;
    float keep_scale;
    float* dc;            /* [B][H] carry, in: dL/dc_t from step t+1, out: dL/dc_{t-1} */
    long long ld_dc;
    float* dgates;        /* [B][4H] out; may be NULL beside dgates16 (round 6): the operand copy is then the cell's only gate-gradient output */
    long long ld_dgates;
    const int* lens;
    int t;
    void* dgates16;       /* optional bf16 copy of dgates (bf16 operand mode of the dgrad GEMM) */
    long long ld_dgates16;
    int dgates16_x3;      /* 1: dgates16 receives the SPLIT-bf16 image of dgates instead (t2amd_split_bf16x3_f32 layout, ld_dgates16 in k) */
} t2amd_lstm_bwd;

int t2amd_lstm_pointwise_bwd_f32(const t2amd_lstm_bwd* a, void* stream);
int t2amd_lstm_pointwise_bwd2_f32(const t2amd_lstm_bwd* a, const t2amd_lstm_bwd* b, void* stream);

/* ------------------------------------------------------------------------------------
 * Location-sensitive attention, one decoder step (reference model.py:43-86 Attention.forward /
 * get_alignment_energies, :22-26 LocationLayer, :358-365 concat + cumulative update).
 * U[d][c*31+k] = sum_f Wdense[d][f]*Wconv[f][c][k] is the location conv and dense folded
 * into one 62-tap filter per attention dim (t2amd_fold_location_f32, U = 128*62 floats).
 * Each step is two launches of T2AMD_ATT_SLICES x B workgroups (energies over dim slices, then
 * softmax + context over channel slices); partial results cross between them through `ws`.
 * ------------------------------------------------------------------------------------ */
int t2amd_fold_location_f32(const float* wdense, const float* wconv, float* U, void* stream);
/* dWdense[d][f] = sum_ck dU[d][ck]*Wconv[f][ck]; dWconv[f][ck] = sum_d Wdense[d][f]*dU[d][ck];
 * dU = sum over nb per-utterance accumulators dU_acc[nb][128][62]; dv = sum_b dv_acc[nb][128] */
int t2amd_unfold_location_grads_f32(const float* dU_acc, const float* dv_acc, int nb,
                                    const float* wdense, const float* wconv, float* dwdense,
                                    float* dwconv, float* dv, void* stream);

typedef struct t2amd_attn_fwd {
    int B, Ti, E, Hq;        /* E = encoder dim (multiple of 16), Hq = attention_rnn_dim (multiple of 32) */
    const float* h;          /* [B][Hq] query source (dropped-out attention hidden), 16-byte aligned rows */
    long long ld_h;
    const float* Wq;         /* [128][Hq] query_layer weight, as stored by nn.Linear */
    const float* U;          /* [128][62] */
    const float* v;          /* [128] */
    const float* pm;         /* [B][Ti][128] processed memory */
    const float* memory;     /* [B][Ti][E] */
    const int* lens;         /* [B] or NULL (no mask: reference inference) */
    const float* w_prev;     /* previous weights, row b at w_prev + b*ld_wprev; NULL = zeros */
    long long ld_wprev;
    float* cum;              /* [B][Ti] running cumulative weights, updated in place */
    float* cum_save;         /* [B][Ti] copy of cum BEFORE the update, or NULL */
    float* w_out;            /* row b at w_out + b*ld_wout */
    long long ld_wout;
    float* ctx_out;          /* [B][E] */
    long long ld_ctx;
    float* q_out;            /* [B][128] or NULL */
    long long ld_q;
    const uint8_t* active;   /* [B] or NULL: rows with active[b]==0 are skipped (batched inference) */
    float* ws;               /* workspace, >= T2AMD_ATT_SLICES*B*Ti floats (partial energies) */
    void* ctx16_out;         /* optional bf16 copy of the context [B][ld_ctx16] (bf16 operand mode) */
    long long ld_ctx16;
    /* 1: the location convolution runs as a split-bf16 product (U = Uh + Ul, window = Wh + Wl in bf16;
     * Uh.Wh + Uh.Wl + Ul.Wh on v_mfma_f32_16x16x32_bf16, f32 accumulate: ~2^-17 relative per product) -- the
     * engine's bf16 compute mode.  0: exact-f32 MFMA (parity mode). */
    int loc_split_bf16;
    /* optional bf16 copy of `memory` ([B][Ti][E] bf16): the context product then streams it instead of the f32 rows
     * (the kernel is bound by the bytes of those rows); weights, accumulation and the context stay f32. */
    const void* memory16;
    /* optional bf16 copy of Wq ([128][Hq] bf16): the query product q = Wq h streams it (h and the sums stay f32) */
    const void* Wq16;
    /* Size of ws in floats, or 0 = "the minimum".  With at least t2amd_attn_fwd_ws_floats(B, Ti) floats -- the block
     * behind the partial energies zeroed once -- the step may run as ONE launch whose four workgroups per utterance hand
     * the partial energies to each other as 8-byte {launch token, value} granules (t2amd_set_attn_fwd_fused). */
    long long ws_floats;
    /* 1: ctx16_out receives the SPLIT-bf16 image of the context (t2amd_split_bf16x3_f32 layout, ld_ctx16 in k; E % 16 == 0) --
     * the operand of the 'bf16x3' LSTM tiles; everything else of the step stays in its exact-f32 form. */
    int ctx16_x3;
} t2amd_attn_fwd;

int t2amd_attention_step_fwd_f32(const t2amd_attn_fwd* a, void* stream);
/* floats of ws that every form of the call can use (a multiple of 4) */
long long t2amd_attn_fwd_ws_floats(int B, int Ti);
/* 1: K_e and K_c of a step run as one launch (energy granules between the four workgroups of an utterance), 0: two
 * launches, -1: library default / environment T2AMD_ATTN_FWD_FUSED.  Bit-identical results.  The one-launch form serves
 * launches of at most 512 workgroups (B <= 128) with Ti <= 512.  Its launch token is a kernel argument drawn from a
 * process-wide counter at enqueue time: do not capture these launches into a hipGraph (a replay would present the same
 * token again and accept the previous replay's granules); the same holds for the one-launch forms of the backward. */
int t2amd_set_attn_fwd_fused(int on);

typedef struct t2amd_attn_bwd {
    int B, Ti, E, Hq;
    t2amd_addend dctx[3];    /* gradient wrt the context vector, column offset pre-applied to p */
    float* dctx_total;       /* [B][E] out (saved for the deferred d_memory GEMM) */
    long long ld_dctx_total;
    const float* d_w_extra;  /* upstream grad wrt this step's weights (alignments output) or NULL */
    long long ld_dwextra;
    const float* q;          /* [B][128] saved */
    long long ld_q;
    const float* Wq;         /* [128][Hq] */
    const float* U;          /* [128][62] */
    const float* v;
    const float* pm;
    const float* memory;
    const int* lens;
    const float* w;          /* this step's weights, row stride ld_w */
    long long ld_w;
    const float* w_prev;     /* NULL = zeros */
    long long ld_wprev;
    const float* cum_before; /* [B][Ti] */
    /* Carries between steps, in partial form.  dwin_part[s][b][c][ti], s < T2AMD_ATT_SLICES, c = 0: grad
     * wrt the PREVIOUS weights (the next-processed step's w), c = 1: grad wrt the cumulative weights, both
     * from the location input of the step processed before this call.  In: partials of step t+1; out:
     * partials of this step.  dcum_acc[b][ti]: running gradient wrt the cumulative weights; on return it
     * includes the incoming c = 1 partials.  Zero both before the last time step. */
    float* dwin_part;        /* [T2AMD_ATT_SLICES][B][2][Ti] in/out */
    float* dcum_acc;         /* [B][Ti] in/out */
    float* d_pm;             /* [B][Ti][128] accumulated */
    float* dU_acc;           /* [B][128][62] accumulated */
    float* dv_acc;           /* [B][128] accumulated */
    float* dq_out;           /* [B][128] */
    long long ld_dq;
    float* dh_out;           /* [T2AMD_ATT_SLICES] partial slabs of Wq^T dq: slice s, row b at dh_out + s*dh_split_stride + b*ld_dh */
    long long ld_dh;
    long long dh_split_stride;
    float* ws;               /* workspace, >= B*Ti + 8*B floats (12*B with cell_q).  Words [B*Ti + 4*B, B*Ti + 8*B) (and
                              * the next 4*B with cell_q) receive the hand-off tokens of the fused kernel (a nonzero
                              * launch counter): zero them once before the first call and do not let other kernels
                              * write there between calls. */
    /* 1: the two gradient products of the location layer (dcol = U^T dpre, dU += dpre^T im2col) round their
     * operands to bf16 and run on v_mfma_f32_16x16x32_bf16 (f32 accumulate) -- the engine's bf16 compute mode.
     * 0: exact-f32 MFMA.  The recompute of the location conv uses the forward's split-bf16 form (see
     * t2amd_attn_fwd.loc_split_bf16) when this is 1, the exact-f32 MFMA otherwise.
     * 2 (round 6, the 'bf16x3' mode): the recompute in the split-bf16 form, the two gradient products exact f32. */
    int bf16;
    /* optional bf16 copy of `memory`: dw = dctx . memory streams it instead of the f32 rows */
    const void* memory16;
    /* Optional (both NULL = off): the LSTM cell backwards of a BPTT step (reference: nn.LSTMCell under autograd,
     * model.py:351-352, 366-370), run as the closing phase of THIS launch instead of a t2amd_lstm_pointwise_bwd2_f32
     * launch of their own.
     *   cell_q: the cell whose dL/dh contains this step's W_q^T dq -- the attention LSTM of the step.  Its dh[1] must
     *           describe dh_out's T2AMD_ATT_SLICES slabs (what a separate launch would read); H = Hq, same B.
     *   cell_x: a cell that does not depend on this launch (the decoder LSTM of step t-1), or NULL.
     * In the folded form the four workgroups of an utterance exchange their dq slices through dq_out behind a second
     * token block (ws must then hold B*Ti + 12*B floats, the last 8*B zeroed once), each forms W_q^T dq for a quarter
     * of the columns in the summation order of the dh_out slabs, and runs both cells for those units: results are
     * bit-identical to the separate launch; dh_out is not written.  Geometries the folded form does not cover
     * (Hq % 16 != 0, H > 1024, the two-launch form of this step) run the cells as a separate launch from inside the
     * call. */
    const t2amd_lstm_bwd* cell_q;
    const t2amd_lstm_bwd* cell_x;
    /* Size of ws in floats, or 0 = "the minimum".  With at least t2amd_attn_bwd_ws_floats(B, Ti) floats (zeroed once from
     * word B*Ti on) the first hand-off of the one-launch form may travel as 8-byte {launch token, value} granules in the
     * block behind the token words (t2amd_set_attn_bwd_granules). */
    long long ws_floats;
    /* optional bf16 copy of Wq ([128][Hq] bf16, 16-byte aligned, Hq % 8 == 0): the closing product dh = Wq^T dq streams it instead
     * of the f32 rows (dq and the sums stay f32) -- the engine's bf16 compute mode; the folded and the separate-launch forms
     * read the same copy in the same order and stay bit-identical to each other. */
    const void* Wq16;
} t2amd_attn_bwd;

int t2amd_attention_step_bwd_f32(const t2amd_attn_bwd* a, void* stream);
/* floats of ws that every form of the call can use: dw slab, slice partials, two token blocks, granule block */
long long t2amd_attn_bwd_ws_floats(int B, int Ti);
/* First hand-off of the one-launch form (dw slices between the four workgroups of an utterance): 1 = 8-byte
 * {token, value} granules polled by their consumers, 0 = write-through stores + drain + token + payload loads,
 * -1 = library default / environment T2AMD_ATTN_GRANULES.  Bit-identical results. */
int t2amd_set_attn_bwd_granules(int on);
/* Number of in-launch hand-offs of the one-launch attention forms that were ABANDONED since the last reset (their bounded
 * 50 ms spin ran out: the workgroups of an utterance were not co-resident, e.g. a shared or partitioned GPU).  Such a
 * launch poisons its outputs with NaN; a caller that finds a non-finite loss / gradient norm asks here whether that is
 * the reason and, if so, selects the separate-launch forms (t2amd_set_attn_fwd_fused(0), T2AMD_ATTN_FUSED_BWD=0 /
 * t2amd_set_attn_bwd_fused(0)).  Synchronises with the device; -1 on error. */
int t2amd_attn_handoff_timeouts(int reset);
/* 0: the attention backward of a step as two launches (K_b1, K_b2; no in-launch hand-off, the LSTM cell backwards in
 * their own launch); 1: one launch (default); -1: the T2AMD_ATTN_FUSED_BWD environment default */
int t2amd_set_attn_bwd_fused(int on);

/* ------------------------------------------------------------------------------------
 * Device-resident time loops.  One host call enqueues every step's kernels.
 * ------------------------------------------------------------------------------------ */
/* Teacher-forced decoder loop, forward: reference model.py:405-411 (the while loop) around
 * Decoder.decode :340-379.  Prenet, input projection, memory_layer and the mel/gate
 * projection are hoisted out of the loop by the host (dense GEMMs over all steps). */
typedef struct t2amd_dec_train {
    int B, Ti, To, E, Ha, Hd;
    /* weights */
    const float* Wa_rec;   /* [4Ha][E+Ha] = [W_ih_att[:, P:P+E] | W_hh_att] */
    const float* Wd_cat;   /* [4Hd][Ha+E+Hd] = [W_ih_dec | W_hh_dec] */
    const float* bias_d;   /* [4Hd] = b_ih + b_hh */
    const float* Wq;       /* [128][Ha] */
    const float* U;        /* [128][62] */
    const float* v;        /* [128] */
    /* inputs */
    float* GA;             /* [To][B][4Ha] in: prenet input projection + att biases; out: activated gates */
    const float* memory;   /* [B][Ti][E] */
    const float* pm;       /* [B][Ti][128] */
    const int* lens;       /* [B] */
    const uint8_t* keep_att; /* [To][B][Ha] */
    const uint8_t* keep_dec; /* [To][B][Hd] */
    float scale_att, scale_dec;
    /* saved slabs (outputs) */
    float* HA;  /* [To][B][Ha] */
    float* CA;  /* [To][B][Ha] */
    float* GD;  /* [To][B][4Hd] */
    float* HD;  /* [To][B][Hd] */
    float* CD;  /* [To][B][Hd] */
    float* CTX; /* [To][B][E] */
    float* Q;   /* [To][B][128] */
    float* ALIGN; /* [B][To][Ti] */
    float* CUM;   /* [To][B][Ti] cumulative weights before each step */
    float* cum_work; /* [B][Ti] scratch (zeroed by the call) */
    float* attn_ws;  /* >= T2AMD_ATT_SLICES*B*Ti + B*Ti + 8*B floats: attention workspace (the backward loop uses the
                      * part behind the forward's T2AMD_ATT_SLICES*B*Ti) */
    /* bf16 operand mode (all NULL / 0 for f32): bf16 copies of the packed weights and of the three recurrent
     * operand slabs; the LSTM products then run on the bf16 MFMA, state and slabs above stay f32. */
    int bf16;
    const void* Wa_rec16;  /* [4Ha][E+Ha] bf16 */
    const void* Wd_cat16;  /* [4Hd][Ha+E+Hd] bf16 */
    void* HA16;            /* [To][B][Ha] bf16 */
    void* HD16;            /* [To][B][Hd] bf16 */
    void* CTX16;           /* [To][B][E] bf16 */
    const void* memory16;  /* [B][Ti][E] bf16 copy of memory, or NULL: attention context / its backward stream it */
    const void* Wq16;      /* [128][Ha] bf16 copy of Wq, or NULL: the forward query product streams it */
} t2amd_dec_train;

int t2amd_decoder_train_fwd_loop_f32(const t2amd_dec_train* p, void* stream);

/* The same loop as ONE persistent launch (reference model.py:405-411 around Decoder.decode :340-379; BASELINE north_star:
 * "the decode loop is fused into a persistent wavefront-resident kernel").  max(Ha/8 + Hd/8, 4 B) co-resident 512-thread
 * workgroups alternate the LSTM-tile role and the attention role of every time step; the two all-to-all edges of a step are
 * flag + data hand-offs (write-through payload, one step counter per workgroup in `flags`:
 * t2amd_decoder_train_fwd_persistent_flag_bytes(B, Ha) bytes, zeroed by the call).  Same arithmetic in the same order as the
 * launch chain above: every output is bit-identical to it.  Both operand modes since round 5: p->bf16 with every bf16 copy
 * present (bf16 MFMA tiles, bf16 attention streams), or p->bf16 == 0 -- the fp32 parity mode: the same loop over the f32 slabs,
 * tiles on the exact-f32 MFMA, which the chain above then runs as well.  Inside the launch a workgroup fetches the first four
 * k-tiles of its NEXT LSTM tile while it sits in its attention step (T2AMD_DTP_PREFETCH=0 switches that off).
 * B <= 64, Ti <= 512; `_supported` returns 0 when the geometry fits a device of `cus` compute units (every workgroup must be
 * resident at once), else T2AMD_ERR_ARG with the reason in t2amd_last_error().  Spins are bounded (50 ms of the wall clock):
 * a give-up sets *status != 0 and every workgroup leaves; with `poison` non-NULL a one-thread launch behind the kernel
 * turns poison[0] into NaN in that case (for callers that do not read status back) and counts it for
 * t2amd_attn_handoff_timeouts(). */
long long t2amd_decoder_train_fwd_persistent_flag_bytes(int B, int Ha);
int t2amd_decoder_train_fwd_persistent_supported(const t2amd_dec_train* p, int cus);
int t2amd_decoder_train_fwd_persistent_f32(const t2amd_dec_train* p, unsigned* flags, int* status, float* poison, void* stream);

/* BPTT through the same loop (what autograd replays for reference train.py:226). */
typedef struct t2amd_dec_train_bwd {
    t2amd_dec_train f;       /* the forward description (slabs now inputs) */
    const float* Wa_recT;    /* [E+Ha][4Ha] */
    const float* Wd_catT;    /* [Ha+E+Hd][4Hd] */
    const float* DHC;        /* [To][B][Hd+E] grad wrt [h_dec | ctx] from the projection */
    const float* d_align;    /* [B][To][Ti] or NULL */
    int nsplit;              /* split-K factor of the two backward skinny GEMMs */
    /* outputs */
    float* DGA;  /* [To][B][4Ha] */
    float* DGD;  /* [To][B][4Hd]; round 6: DGA and DGD may both be NULL in the bf16 mode when DGA16 / DGD16 below are whole-sequence
                  * slabs (dg16_step_* > 0): nothing downstream reads the f32 slabs then (weight and bias gradients come from the
                  * bf16 slabs, t2amd_gemm16_kk_group / t2amd_colsum_bf16) and the cells stop writing them -- 2 MB per time step */
    float* DCTX; /* [To][B][E] */
    float* DQ;   /* [To][B][128] */
    float* d_pm; /* [B][Ti][128] (zeroed by the call) */
    float* dU_acc; /* [B][128][62] (zeroed by the call) */
    float* dv_acc; /* [B][128] (zeroed by the call) */
    /* workspaces */
    float* dXd;  /* [To][nsplit][B][Ha+E+Hd]: per step, the decoder-LSTM chain runs ahead of its consumers */
    float* dXa;  /* [nsplit][B][E+Ha] */
    float* dc_a; /* [B][Ha] */
    float* dc_d; /* [B][Hd] */
    float* dwin_part;  /* [T2AMD_ATT_SLICES][B][2][Ti] (zeroed by the call) */
    float* dcum_acc;   /* [B][Ti] (zeroed by the call) */
    float* dq_h;       /* [T2AMD_ATT_SLICES][B][Ha] */
    /* bf16 operand mode (f.bf16 != 0) */
    const void* Wa_recT16; /* [E+Ha][4Ha] bf16 */
    const void* Wd_catT16; /* [Ha+E+Hd][4Hd] bf16 */
    void* DGA16;           /* [B][4Ha] bf16 scratch: this step's attention-LSTM gate gradients */
    void* DGD16;           /* [B][4Hd] bf16 scratch */
    /* 0: DGA16 / DGD16 are one step's scratch, reused by every step; B*4Ha / B*4Hd: they are [To][B][4H] slabs and step t
     * writes (and its dgrad reads) its own rows -- the engine then builds the weight-gradient operands from them */
    long long dg16_step_a, dg16_step_d;
    /* (round 6) 0: dXd holds To step slabs (needed when the decoder-LSTM chain runs ahead on its own stream,
     * t2amd_set_decoder_streams(2)); R >= 3: dXd holds R step slabs used as a ring (step t -> slab t % R) -- a slab is read
     * one step after it is written and never again, and a ring stays in L2 / MALL instead of streaming To slabs to HBM.
     * Single-stream loop only. */
    int dXd_ring;
} t2amd_dec_train_bwd;

int t2amd_decoder_train_bwd_loop_f32(const t2amd_dec_train_bwd* p, void* stream);

/* (Round 6: the BPTT loop as ONE persistent launch -- t2amd_decoder_train_bwd_persistent_f32 and its _supported / _flag_bytes /
 * _desc_bytes queries, opt-in since round 4 -- was removed: two rounds at 1 ms behind the chain above, profiles/r04_j_ab_train_bwd_
 * persistent.json) */
/* 1 (default): everything on the caller's stream; the decoder-LSTM chain (off the attention recurrence under
 * teacher forcing) shares fused launches with the attention-LSTM chain.  2: the decoder-LSTM chain of both
 * training loops runs on an internal side stream, ordered against the caller's stream with one hipEvent per
 * 8 steps. */
int t2amd_set_decoder_streams(int n);
/* 1: the two LSTM cell backwards of a decoder BPTT step (attention LSTM of step t, decoder LSTM of step t-1) run as the
 * closing phase of the step's attention-backward launch (t2amd_attn_bwd.cell_q / cell_x): 5 dependent launches per
 * decoder time step instead of 6.  0: a launch of their own.  Gradients are bit-identical either way.  Start-up value:
 * environment T2AMD_CELL_FOLD, else the library default.  Single-stream loop only (t2amd_set_decoder_streams(1)). */
int t2amd_set_bptt_cell_fold(int on);
int t2amd_get_bptt_cell_fold(void);   /* the current value (0 / 1) */
/* Free-running decoder (t2amd_decoder_infer_steps_f32, reference model.py:418-454): the largest batch served by the matrix-vector
 * kernels; above it the step runs on the 64-row MFMA tiles, whose cost does not depend on B <= 64.  -1 (default): by operand mode
 * -- 3 rows with bf16 operands (t2amd_dec_infer.bf16 == 1), 4 otherwise (measured: profiles/r06_s_bench_infer_small_batches.txt);
 * 0 .. 8: that many rows (the matrix-vector kernels hold at most 8; 0 = tiles at every B).  Start-up value: environment
 * T2AMD_SMALL_BATCH_MAX.  A caller that sets up t2amd_dec_infer asks t2amd_dec_infer_uses_tiles (below) to know which operand copies the descriptor needs (bf16 / split-bf16 images and the folded prenet matrix Wf on the tile
 * path); bf16 == 3 with B at or below the boundary is refused (the split images belong to the tiles: pass 0 there). */
int t2amd_set_small_batch_max(int n);
int t2amd_get_small_batch_max(int bf16);
/* 1 when t2amd_decoder_infer_steps_f32 runs a batch of B rows in operand mode `bf16` on the tiles, 0 for the matrix-vector kernels:
 * B above the boundary -- except that bf16 operands on widths that are not multiples of 128 (which the bf16 tiles refuse) stay on
 * the matrix-vector kernels up to their limit of 8 rows.  What a caller asks before it fills t2amd_dec_infer. */
int t2amd_dec_infer_uses_tiles(int B, int bf16, int E, int Ha, int Hd, int P);

/* Encoder bi-LSTM over [B][T][.] with packed-sequence semantics (reference model.py:180-188).
 * GX [B][T][4H] holds x.W_ih^T + b_ih + b_hh (hoisted dense GEMM) and is overwritten with the
 * activated gates.  out: [B][T][ld_out] slice (direction offset pre-applied). */
typedef struct t2amd_lstm_seq {
    int B, T, H, reverse;
    const float* Whh;   /* [4H][H] */
    const float* WhhT;  /* [H][4H] (backward only) */
    float* GX;          /* [B][T][4H] */
    float* out;         /* h outputs, element (b,t,j) at out[(b*T+t)*ld_out + j] */
    long long ld_out;
    float* C;           /* [T][B][H] cell states */
    const int* lens;
    /* backward only */
    const float* dout;  /* grad wrt out, same indexing with ld_dout */
    long long ld_dout;
    float* DG;          /* [B][T][4H] out: gate pre-activation grads */
    float* dX;          /* [max(dx_splits, 1)][B][H] workspace */
    float* dc;          /* [B][H] workspace */
    /* backward: 2..4 = the recurrent data-gradient product dh_prev = dgates . Whh of every step splits its K = 4H over
     * that many groups of workgroups, which write partial slabs of dX that the next step's cell backward adds in index
     * order (H/16 column tiles alone are 16 workgroups per direction at H = 256: a 16-tile dependent k loop on a
     * handful of CUs).  0 / 1: one slab. */
    int dx_splits;
} t2amd_lstm_seq;

int t2amd_lstm_seq_fwd_f32(const t2amd_lstm_seq* p, void* stream);
int t2amd_lstm_seq_bwd_f32(const t2amd_lstm_seq* p, void* stream);
/* Both directions of the bi-LSTM in lockstep, one launch per step for the pair (q may be NULL). */
int t2amd_lstm_seq_fwd2_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, void* stream);
int t2amd_lstm_seq_bwd2_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, void* stream);
/* The same forward recurrence for ONE utterance (B == 1, H <= 256; Encoder.inference, reference model.py:192-201) as ONE
 * persistent launch of 2 x H/4 co-resident workgroups: W_hh rows in registers, h exchanged as {step + 1, f32} granules in
 * `mailbox` (t2amd_lstm_seq_persistent_mailbox_bytes(H, ndir) bytes, zeroed by the call).  Inference only: writes `out`,
 * leaves GX / C untouched.  Bounded spins: *status != 0 afterwards = a workgroup gave up, run t2amd_lstm_seq_fwd2_f32. */
long long t2amd_lstm_seq_persistent_mailbox_bytes(int H, int ndir);
int t2amd_lstm_seq_persistent_supported(const t2amd_lstm_seq* p);
int t2amd_lstm_seq_fwd2_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned long long* mailbox,
                                       int* status, void* stream);
/* The same recurrence for a BATCH (B > 1: the training forward, batched inference) as ONE persistent launch of
 * ndir x ceil(B / 32) x H/4 co-resident workgroups: each keeps its 16 gate rows of W_hh in registers as exact-f32 MFMA
 * fragments and the cell state of its 32 x 4 cells; h(s) is handed on through the output slab itself (write-through stores,
 * one step counter per workgroup in `flags` -- t2amd_lstm_seq_batch_persistent_flag_bytes(B, H, ndir) bytes, zeroed by the
 * call -- polled by the H/4 workgroups of the same direction and row group).  Writes everything the launch chain writes
 * (out, C, GX overwritten with the activated gates): the backward is unchanged.  Exact f32 like t2amd_lstm_seq_fwd2_f32 (other
 * summation order: ~2e-7).  Bounded spins: *status != 0 afterwards = a workgroup gave up (GPU shared?): recompute GX and run
 * t2amd_lstm_seq_fwd2_f32.  `cus` = the device's CU count (at most 4 workgroups per CU are assumed co-resident).
 * `poison` (may be NULL): for callers that do not read `status` back -- the training step, where a host sync per step would
 * tie the step time to the host's enqueue speed -- a give-up writes NaN to *poison behind the launch (the step goes non-finite
 * and is skipped like an abandoned attention hand-off) and t2amd_encoder_handoff_timeouts() counts it. */
long long t2amd_lstm_seq_batch_persistent_flag_bytes(int B, int H, int ndir);
int t2amd_lstm_seq_batch_persistent_supported(const t2amd_lstm_seq* p, int ndir, int cus);
int t2amd_lstm_seq_fwd2_batch_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned* flags, int* status,
                                             float* poison, void* stream);
/* BPTT of the same recurrence for a batch as ONE persistent launch (reference model.py:181-188 under autograd) instead of
 * 2 T dependent launches (t2amd_lstm_seq_bwd2_f32): same workgroup layout, flag buffer size and hand-off as the forward launch
 * above; the gate gradients go straight into p->DG (the slab the weight-gradient GEMMs read), the recurrent data gradient is a
 * split-bf16 (hi + lo, three products, f32 accumulate: ~2^-17 relative) MFMA product against W_hh^T rows held in registers.
 * Reads WhhT, GX (activated gates), C, lens, dout; dX / dc of the descriptor are not used.  Results equal the chain's to that
 * product's rounding.  `_supported` / status / poison as for the forward launch. */
int t2amd_lstm_seq_bwd2_batch_persistent_supported(const t2amd_lstm_seq* p, int ndir, int cus);
int t2amd_lstm_seq_bwd2_batch_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned* flags, int* status,
                                             float* poison, void* stream);
/* give-ups of that launch since the last reset (synchronises the device); negative on a runtime error */
int t2amd_encoder_handoff_timeouts(int reset);

/* Free-running decoder (reference model.py:418-454 Decoder.inference), any B: per-utterance
 * stop flags on the device, stop test sigmoid(gate) > threshold (strict) after the frame is
 * emitted.  Runs `n_steps` steps starting at step `t0` (host polls `done_count` between calls). */
typedef struct t2amd_dec_infer {
    int B, Ti, E, Ha, Hd, P, C;   /* P = prenet dim, C = n_mel_channels */
    int t0, n_steps, max_steps;
    float gate_threshold;
    const float* W1;       /* prenet layer 0 [P][C] */
    const float* W2;       /* prenet layer 1 [P][P] */
    const float* Wa_cat;   /* [4Ha][P+E+Ha] = [W_ih_att | W_hh_att] */
    const float* bias_a;   /* [4Ha] */
    const float* Wd_cat;   /* [4Hd][Ha+E+Hd] */
    const float* bias_d;
    const float* Wq;       /* [128][Ha] */
    const float* U;
    const float* v;
    const float* Wpg;      /* [C+1][Hd+E] = [linear_projection ; gate_layer] rows */
    const float* bias_pg;  /* [C+1] */
    const float* memory;
    const float* pm;
    const int* lens;       /* NULL = unmasked (reference B==1 path) */
    const uint8_t* keep_prenet; /* [max_steps][2][B][P] */
    /* state (persist across calls; zeroed by the caller before t0 == 0) */
    float* h_a;            /* [2][B][Ha] ping-pong: step t reads slot t&1, writes slot (t+1)&1 */
    float* c_a;            /* [2][B][Ha] */
    float* c_d;            /* [2][B][Hd] */
    float* hc;             /* [2][B][Hd+E] = [h_dec | ctx] ping-pong */
    float* cum;            /* [B][Ti] */
    float* x_prenet;       /* [2][B][P] scratch */
    float* gates;          /* unused since round 6 (was: [B][4*max(Ha,Hd)] scratch for gate activations nobody read; may be NULL) */
    const float* zero_frame; /* [B][C] zeros (go frame) */
    float* attn_ws;        /* >= T2AMD_ATT_SLICES*B*Ti floats */
    /* outputs */
    float* PG;             /* [max_steps][B][C+1] mel frame + gate logit per step */
    float* ALIGN;          /* [B][max_steps][Ti] */
    int* out_lengths;      /* [B] frames emitted incl. the stopping frame (0 while running) */
    uint8_t* active;       /* [B] 1 while the utterance is still decoding */
    int* done_count;       /* [1] number of finished utterances */
    /* bf16 operand mode for the two LSTM products (all NULL / 0 for f32): bf16 copies of the packed weights and,
     * for B > M (MFMA tile path; M = t2amd_get_small_batch_max(bf16): 3 with bf16 operands, 4 otherwise, unless set), of the recurrent operands, written by their producers next to the f32 values; the
     * matrix-vector kernels of B <= M read the bf16 weight rows against f32 inputs.  State, gates, attention and
     * outputs stay f32. */
    /* (3, round 6, B > M only: the 'bf16x3' mode -- Wa_cat16 / Wd_cat16 / x_prenet16 / h_a16 / hc16 are SPLIT-bf16 images
     * (t2amd_split_bf16x3_f32: two bf16 per k), the two LSTM steps run on the wide tile's split form; the optional pointers below
     * stay NULL: prenet, projection and attention keep their f32 operands) */
    int bf16;
    const void* Wa_cat16;  /* [4Ha][P+E+Ha] bf16 */
    const void* Wd_cat16;  /* [4Hd][Ha+E+Hd] bf16 */
    void* x_prenet16;      /* [B][P] bf16: prenet output of the current step */
    void* h_a16;           /* [2][B][Ha] bf16 ping-pong (zeroed before t0 == 0) */
    void* hc16;            /* [2][B][Hd+E] bf16 ping-pong (zeroed before t0 == 0) */
    /* B > M, optional: prenet layer 0 folded through the frame projection, Wf = W1 . Wp [P][Hd+E], bias_f = W1 . bp [P]
     * (p1 = relu(W1 (Wp hc + bp)) = relu(Wf hc + bias_f)): layer 0 of step t+1 then rides in step t's projection launch
     * instead of being a B x 256 x 80 tiled GEMM of its own.  NULL: the unfolded form. */
    const float* Wf;
    const float* bias_f;
    /* bf16 mode, B > M, optional: bf16 copies of the encoder memory [B][Ti][E] and of W_q [128][Ha] for the attention
     * kernels (they are bound by those streams); NULL: the f32 arrays are read. */
    const void* memory16;
    const void* Wq16;
    /* bf16 mode, B > M, optional (all four or none): bf16 copies of Wf [P][Hd+E], Wpg [C+1][Hd+E] and W2 [P][P] and a
     * [B][P] bf16 buffer for prenet layer 0's output -- the per-step prenet and frame/gate projection then run on the
     * bf16 MFMA path from the bf16 state copies the LSTM / attention kernels already write (f32 accumulation, f32
     * outputs; as the training loop's bf16 mode computes them) */
    const void* Wf16;
    const void* Wpg16;
    const void* W2_16;
    void* x_prenet1_16;
    /* floats in attn_ws, or 0 = "the minimum".  With t2amd_attn_fwd_ws_floats(B, Ti) floats, zeroed by the caller before the
     * first call, the attention step of launches of at most 512 workgroups runs as ONE launch (t2amd_attn_fwd.ws_floats). */
    long long attn_ws_floats;
} t2amd_dec_infer;

int t2amd_decoder_infer_steps_f32(const t2amd_dec_infer* p, void* stream);

/* Persistent, weight-stationary decode loop for ONE utterance (reference model.py:435-449 Decoder.inference's loop
 * around Decoder.decode :340-379; Prenet :97-100; Attention :43-86; LocationLayer :22-26): ONE launch of H/4
 * co-resident workgroups runs every step until the gate fires or max_steps is reached.  Each workgroup keeps its 16 rows
 * of both LSTM matrices (bf16 in LDS, or f32 split between LDS and registers: weights_f32) for the whole utterance; the six per-step vectors (p2, h_att, partial energies,
 * context, h_dec, p1 + stop flag) travel between workgroups as 8-byte {step + 1, f32} granules (agent-scope relaxed
 * stores / polled loads, no fences).  Prenet layer 0 is folded through the frame projection (Wf below).  Requires
 * attention_rnn_dim == decoder_rnn_dim = H, H/4 <= number of CUs, Ti <= 256 (t2amd_decoder_persist_supported).  Spins
 * are bounded (30 ms): on a timeout *status = T2AMD_PERSIST_TIMEOUT, every workgroup leaves and the caller must use
 * t2amd_decoder_infer_steps_f32 instead. */
#define T2AMD_PERSIST_TIMEOUT 7
typedef struct t2amd_dec_persist {
    int Ti, E, H, P, C;
    int max_steps;
    float gate_threshold;
    const void* Wa16;      /* [4H][P+E+H] bf16 = [W_ih_att | W_hh_att] */
    const void* Wd16;      /* [4H][H+E+H] bf16 = [W_ih_dec | W_hh_dec] */
    const float* bias_a;   /* [4H] b_ih + b_hh */
    const float* bias_d;   /* [4H] */
    const float* Wq;       /* [128][H] */
    const float* U;        /* [128][62] folded location filter (t2amd_fold_location_f32) */
    const float* v;        /* [128] */
    const float* Wf;       /* [P + C + 1][H+E]: rows 0..P-1 = W_prenet0 . W_proj, rows P..P+C-1 = W_proj, row P+C = W_gate */
    const float* bias_f;   /* [P + C + 1]: W_prenet0 . b_proj, b_proj, b_gate */
    const float* W2;       /* [P][P] prenet layer 1 */
    const float* memory;   /* [Ti][E] encoder outputs */
    const float* pm;       /* [Ti][128] processed memory */
    const uint8_t* keep_prenet; /* [max_steps][2][1][P] */
    float* PG;             /* [max_steps][C+1] frame + gate logit per step */
    float* ALIGN;          /* [max_steps][Ti] */
    int* out_length;       /* [1] frames emitted incl. the stopping frame */
    int* status;           /* [1] 0 = ok */
    int* steps_done;       /* [1] */
    unsigned long long* mailbox;  /* t2amd_decoder_persist_mailbox_bytes() bytes, zeroed by the call */
    float* trace;          /* NULL, or [max_steps][H + E + H + P + P]: h_att, ctx, h_dec, p1(t+1), p2(t+1) per step (tests) */
    unsigned long long* timing; /* NULL, or [32] zeroed by the caller: 100 MHz ticks per phase, summed over the steps, of
                                 * thread 0 of the first ([0..15]) and of the last ([16..31]) workgroup (tools) */
    int weights_f32;       /* 0: Wa16 / Wd16 are bf16 (throughput mode).  1: they point at the SAME matrices in f32 (the fp32
                            * parity mode: exact f32 products, the arithmetic of the fp32 launch chain): a workgroup then
                            * keeps its 16 attention-LSTM rows in LDS (16 x 1792 x 4 = 112 KB) and its 16 decoder-LSTM rows in
                            * REGISTERS (160 per lane: the CU's 512 KB register file holds what its 160 KB of LDS cannot);
                            * needs H and E multiples of 256 */
} t2amd_dec_persist;

long long t2amd_decoder_persist_mailbox_bytes(int Ti, int E, int H, int P);
int t2amd_decoder_persist_supported(const t2amd_dec_persist* p);
int t2amd_decoder_infer_persistent_f32(const t2amd_dec_persist* p, void* stream);

/* ------------------------------------------------------------------------------------
 * Mel front end: element passes around the two GEMMs of
 * TacotronSTFT.mel_spectrogram (reference layers.py:63-80, stft.py:77-105,
 * audio_processing.py:78-84).  The contractions themselves are t2amd_gemm_f32 calls:
 *   spec[n][2F] = frames . basis^T with frames = the padded signal viewed as [n][L] with
 *   row stride hop (lda = hop < K: overlapping rows, no frame matrix), then
 *   mel[n][n_mel] = mag[n][Fpad] . mel_basis[n_mel][Fpad]^T.
 * ------------------------------------------------------------------------------------ */
/* out[b][i] = y[b][reflect(i - pad)] for i < T + 2*pad (torch 'reflect': the edge sample is not repeated;
 * needs pad < T), 0 for T + 2*pad <= i < Tout (row slack so that ldo can be a multiple of 4). */
int t2amd_reflect_pad_f32(const float* y, long long ldy, float* out, long long ldo, int B, int T, int pad,
                          int Tout, void* stream);
/* The index rule of the kernel above, callable on the host: position in y[0..T) of offset i in [-T+1, 2T-2]. */
long long t2amd_reflect_index(long long i, long long T);
/* mag[r][f] = sqrt(spec[r][f]^2 + spec[r][F+f]^2) for f < F, 0 for F <= f < Fpad; rows <= 65535 per call. */
int t2amd_stft_magnitude_f32(const float* spec, long long lds, float* mag, long long ldm, long long rows,
                             int F, int Fpad, void* stream);
/* out[b][m][j] = log(max(mel[(b*n + j)*ld + m], clip)): dynamic range compression + (frame, channel) transpose
 * into the (B, n_mel, n) layout TextMelCollate pads. */
int t2amd_mel_log_compress_f32(const float* mel, long long ld, float* out, int B, int n, int n_mel,
                               float clip, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser step: global-norm clipping + Adam over all parameter
 * tensors in two launches.  Replaces reference train.py:233-236
 *   grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_thresh)
 *   optimizer.step()                      # torch.optim.Adam(lr, weight_decay), train.py:170-171
 * with torch's arithmetic (L2 weight decay folded into the gradient, bias-corrected moments,
 * no amsgrad).  The state tensors (exp_avg, exp_avg_sq) are the ones torch.optim.Adam keeps, so
 * the reference's checkpoint format (train.py:112-118) is unchanged.
 * ------------------------------------------------------------------------------------ */
/* Tacotron2Loss (reference loss_function.py:8-19): loss = mean((mel - y)^2) + mean((mel_post - y)^2) +
 * mean(bce_with_logits(gate, g)) over the padded tensors, one reduction pass (fixed summation order, double partial sums:
 * bit-reproducible) and one gradient pass.  out4 = {total, mel term, postnet term, gate term}; ws holds
 * t2amd_loss_workspace_doubles() doubles; `upstream` is the device scalar d(total)/d(loss) (no host read). */
int t2amd_loss_workspace_doubles(void);
int t2amd_tacotron2_loss_fwd_f32(const float* mel, const float* mel_post, const float* mel_target, long long n_mel,
                                 const float* gate, const float* gate_target, long long n_gate, double* ws, float* out4,
                                 void* stream);
int t2amd_tacotron2_loss_bwd_f32(const float* mel, const float* mel_post, const float* mel_target, long long n_mel,
                                 const float* gate, const float* gate_target, long long n_gate, const float* upstream,
                                 float* d_mel, float* d_mel_post, float* d_gate, void* stream);

#define T2AMD_MAX_TENSORS 64
typedef struct t2amd_tensor_list {
    void* param[T2AMD_MAX_TENSORS];        /* f32, updated in place (unused by t2amd_grad_norm_f32) */
    const void* grad[T2AMD_MAX_TENSORS];   /* f32, read only: the clip factor is applied on the fly, p.grad is NOT rescaled */
    void* exp_avg[T2AMD_MAX_TENSORS];      /* f32 first moment */
    void* exp_avg_sq[T2AMD_MAX_TENSORS];   /* f32 second moment */
    long long numel[T2AMD_MAX_TENSORS];
    int first_block[T2AMD_MAX_TENSORS];    /* running sum of ceil(numel / t2amd_optim_chunk()) */
    int count;
} t2amd_tensor_list;

typedef struct t2amd_adam_hyper {
    float step_size;        /* lr / (1 - beta1^step) */
    float bc2_sqrt;         /* sqrt(1 - beta2^step) */
    float one_minus_beta1;
    float beta2;
    float one_minus_beta2;
    float eps;
    float weight_decay;     /* L2: g += weight_decay * p */
} t2amd_adam_hyper;

/* elements one workgroup covers (4096): callers size first_block[] and the workspace with it */
int t2amd_optim_chunk(void);
/* norm_and_coef[0] = ||all gradients||_2, [1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0).
 * ws: >= total blocks doubles.  Deterministic (fixed summation order). */
int t2amd_grad_norm_f32(const t2amd_tensor_list* L, float max_norm, double* ws, float* norm_and_coef, void* stream);
/* One Adam step on every tensor of the list; gradients are scaled by norm_and_coef[1] when the pointer is non-NULL. */
int t2amd_adam_step_f32(const t2amd_tensor_list* L, const t2amd_adam_hyper* h, const float* norm_and_coef,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TACOTRON2_AMD_H */
