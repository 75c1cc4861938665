// bf16-resident GEMM for gfx950: C[M][N] (f32) = A[M][K] . B[N][K]^T with BOTH operands bf16 and K-contiguous in HBM.
//
// gemm.hip's bf16 kernels read f32 operands and round them on their way into LDS (register staging): they are bound by the
// operand bytes they pull through the CU's vector-memory path (64 KB per 256 x 256 x 32 tile step) and saturate near
// 490 TF/s on the 55 k-deep weight gradients.  Here the operands are already bf16 in HBM (the engine's bf16 mode keeps bf16
// images of every slab it multiplies more than once; weight gradients get K-contiguous images from the transposing cast
// below), so a tile step of K = 64 is 64 KB for twice the MFMA work, and it never touches a register on its way in:
//
//   * 256 x 256 x 64 tile, 512 threads = 8 waves as 2 (m) x 4 (n), each wave 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16
//     tiles (128 accumulator registers), 4 k-steps of 16 per tile step;
//   * LDS-DMA staging (global_load_lds_dwordx4): every wave instruction copies 8 rows x 128 B; two stages of
//     (32 KB A + 32 KB B), tile t+1 in flight while tile t is multiplied; one vmcnt(0) + barrier per tile step;
//   * the DMA writes lane-linear, so bank conflicts are removed on the SOURCE: LDS slot s of row r holds global 16-byte
//     chunk s ^ ((r >> 1) & 7); a fragment read (ds_read_b128, lane -> row l31, chunk 2 ks + lhi) of 16 consecutive rows
//     then covers all 16 slots of the 256-byte bank row;
//   * fragment reads are inline asm (hipcc orders every ds_read it can see behind ALL pending LDS-DMA);
//   * XCD-aware tile order as in gemm.hip; split-K writes partial slabs (reduced by t2amd_splitk_reduce_f32).
//
// Replaces, in the bf16 compute mode: the deferred weight-gradient products dW = dG^T . X of the two decoder LSTMs and the
// hoisted input projection (reference model.py:352-371 under autograd), and nn.Conv1d / nn.Linear products whose operands
// already exist as bf16 images.
#include "common.h"
#include <stdlib.h>

typedef __bf16 g16_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* g16_lptr;
typedef __attribute__((address_space(1))) const void* g16_gptr;

#define G16_T 256            // tile edge (rows of A / rows of B)
#define G16_BK 64            // k per tile step: 128-byte rows
#define G16_IMG (G16_T * 128)        // bytes of one operand image
#define G16_NT 512

struct Gemm16Params {
    t2amd_gemm16_desc d;
    int ktiles_per_split;
};

struct G16Tile { int bx, by, bz; };
// same order as gemm.hip's xcd_tile_id(): XCD j takes a contiguous run of tile ids, ids run through groups of 8 row tiles x
// all column tiles, so the workgroups an XCD runs at once share a few A and B tiles per k step in its own L2
__device__ __forceinline__ G16Tile g16_tile_id() {
    const int nbx = gridDim.x, nby = gridDim.y;
    const int total = nbx * nby * (int)gridDim.z;
    const int L = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
    const int xcd = L & 7, q = L >> 3;
    const int base = total >> 3, rem = total & 7;
    const int id = xcd * base + (xcd < rem ? xcd : rem) + q;
    const int per_z = nbx * nby;
    G16Tile t;
    t.bz = id / per_z;
    const int r = id - t.bz * per_z;
    const int G = 8;
    const int group = r / (G * nbx);
    const int first = group * G;
    const int gsz = (nby - first) < G ? (nby - first) : G;
    const int w = r - group * (G * nbx);
    t.by = first + w % gsz;
    t.bx = w / gsz;
    return t;
}

__global__ __launch_bounds__(G16_NT) void gemm16_tn_kernel(Gemm16Params p) {
    // [A stage 0 | A stage 1 | B stage 0 | B stage 1], 32 KB each: a stage switch is a 16-bit immediate offset
    __shared__ __attribute__((aligned(16))) char smem[4 * G16_IMG];
    const t2amd_gemm16_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;

    const G16Tile tile = g16_tile_id();
    const int split = tile.bz;
    const int row0 = tile.by * G16_T, col0 = tile.bx * G16_T;
    const int M = d.M, N = d.N;
    const int nkt_all = d.K / G16_BK;
    const int kt_beg = split * p.ktiles_per_split;
    int kt_end = kt_beg + p.ktiles_per_split;
    if (kt_end > nkt_all) kt_end = nkt_all;
    const int nk = kt_end > kt_beg ? kt_end - kt_beg : 0;

    // ---- DMA sources: instruction i of this wave fills image rows 64 i + 8 wave + (lane >> 3), slot lane & 7 ----
    const char* asrc[4];
    const char* bsrc[4];
    {
        const int rin = 8 * wave + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rin >> 1) & 7);          // (64 i) >> 1 adds nothing to bits 0..2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ga = row0 + 64 * i + rin, gb = col0 + 64 * i + rin;
            ga = ga < M ? ga : M - 1;                              // clamped rows: their products are never stored
            gb = gb < N ? gb : N - 1;
            asrc[i] = reinterpret_cast<const char*>(d.A) + ((long long)ga * d.lda + (long long)kt_beg * G16_BK) * 2 + chunk * 16;
            bsrc[i] = reinterpret_cast<const char*>(d.B) + ((long long)gb * d.ldb + (long long)kt_beg * G16_BK) * 2 + chunk * 16;
        }
    }
#define G16_ISSUE(STAGE)                                                                                         \
    {                                                                                                            \
        char* ad_ = smem + (STAGE) * G16_IMG + wave * 1024;                                                      \
        char* bd_ = smem + (2 + (STAGE)) * G16_IMG + wave * 1024;                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
            __builtin_amdgcn_global_load_lds((g16_gptr)(asrc[i]), (g16_lptr)(ad_ + i * 8192), 16, 0, 0);         \
            __builtin_amdgcn_global_load_lds((g16_gptr)(bsrc[i]), (g16_lptr)(bd_ + i * 8192), 16, 0, 0);         \
            asrc[i] += 128; bsrc[i] += 128;                                                                      \
        }                                                                                                        \
    }

    // ---- fragment addresses: k-step ks of row r reads slot ((2 ks + lhi) ^ ((r >> 1) & 7)); one VGPR per (tile, ks) ----
    unsigned aaddr[4][4], baddr[2][4];
    {
        const unsigned base = (unsigned)reinterpret_cast<size_t>((g16_lptr)(smem));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = wm * 128 + t * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) aaddr[t][ks] = base + (unsigned)(r * 128 + (((2 * ks + lhi) ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = wn * 64 + t * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                baddr[t][ks] = base + 2 * G16_IMG + (unsigned)(r * 128 + (((2 * ks + lhi) ^ ((r >> 1) & 7)) << 4));
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 fa[4], fb[2], ga[4], gb[2];
#define G16_READ(STAGE, KS, A_, B_)                                                                              \
    asm volatile(                                                                                                \
        "ds_read_b128 %0, %6 offset:%12\n\t"                                                                     \
        "ds_read_b128 %4, %10 offset:%12\n\t"                                                                    \
        "ds_read_b128 %1, %7 offset:%12\n\t"                                                                     \
        "ds_read_b128 %5, %11 offset:%12\n\t"                                                                    \
        "ds_read_b128 %2, %8 offset:%12\n\t"                                                                     \
        "ds_read_b128 %3, %9 offset:%12"                                                                         \
        : "=&v"(A_[0]), "=&v"(A_[1]), "=&v"(A_[2]), "=&v"(A_[3]), "=&v"(B_[0]), "=&v"(B_[1])                     \
        : "v"(aaddr[0][KS]), "v"(aaddr[1][KS]), "v"(aaddr[2][KS]), "v"(aaddr[3][KS]), "v"(baddr[0][KS]),          \
          "v"(baddr[1][KS]), "i"((STAGE) * G16_IMG)                                                              \
        : "memory");
#define G16_WAIT(A_, B_)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                 : "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]), "+v"(A_[3]), "+v"(B_[0]), "+v"(B_[1]) : : "memory");
#define G16_MFMA(A_, B_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g16_bf16x8, A_[i]),           \
                                                                __builtin_bit_cast(g16_bf16x8, B_[j]), acc[i][j], 0, 0, 0);
    // one tile step from stage STAGE: fragments of k-step ks+1 load while k-step ks is multiplied
#define G16_TILE(STAGE)                                                                                          \
    {                                                                                                            \
        G16_READ(STAGE, 0, fa, fb) G16_WAIT(fa, fb)                                                              \
        G16_READ(STAGE, 1, ga, gb) __builtin_amdgcn_sched_barrier(0); G16_MFMA(fa, fb) __builtin_amdgcn_sched_barrier(0); G16_WAIT(ga, gb) \
        G16_READ(STAGE, 2, fa, fb) __builtin_amdgcn_sched_barrier(0); G16_MFMA(ga, gb) __builtin_amdgcn_sched_barrier(0); G16_WAIT(fa, fb) \
        G16_READ(STAGE, 3, ga, gb) __builtin_amdgcn_sched_barrier(0); G16_MFMA(fa, fb) __builtin_amdgcn_sched_barrier(0); G16_WAIT(ga, gb) \
        G16_MFMA(ga, gb)                                                                                         \
    }

    if (nk > 0) {
        G16_ISSUE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            G16_ISSUE(1)                               // tile kt + 1 in flight while tile kt is multiplied
            G16_TILE(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) G16_ISSUE(0)
            G16_TILE(1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (nk & 1) G16_TILE(0)
    }
#undef G16_ISSUE
#undef G16_READ
#undef G16_WAIT
#undef G16_MFMA
#undef G16_TILE

    // ---- epilogue: D layout of the 32 x 32 MFMA: lane -> column l31, register r -> row (r & 3) + 8 (r >> 2) + 4 lhi ----
    float* __restrict__ C = d.C + (long long)split * d.strideSplitC;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int gn = col0 + wn * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = row0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                long long crow = gm;
                bool live = gm < M;
                if (d.win_Tp > 0) {              // window rows (b, t) of pitch Tp -> compact output rows b T + t, t < T
                    const int b_ = gm / d.win_Tp, t_ = gm - b_ * d.win_Tp;
                    live = live && t_ < d.win_T;
                    crow = (long long)b_ * d.win_T + t_;
                }
                if (live && gn < N) {
                    float* cp = C + crow * d.ldc + gn;
                    float val = acc[tm][tn][r];
                    if (d.bias) val += d.bias[gn];
                    if (d.accumulate) val += *cp;
                    *cp = val;
                }
            }
        }
    }
}

extern "C" int t2amd_gemm16_tn(const t2amd_gemm16_desc* dp, void* stream) {
    T2_REQUIRE(dp != nullptr, "gemm16: null descriptor");
    Gemm16Params p;
    p.d = *dp;
    t2amd_gemm16_desc& d = p.d;
    T2_REQUIRE(d.A && d.B && d.C, "gemm16: null operand");
    T2_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.K % G16_BK == 0, "gemm16: K must be a positive multiple of 64");
    T2_REQUIRE(t2_aligned16(d.A) && t2_aligned16(d.B) && d.lda % 8 == 0 && d.ldb % 8 == 0 && d.ldb >= d.K,
               "gemm16: operands must be 16-byte aligned with row strides that are multiples of 8 elements (B's >= K)");
    // A's rows may overlap (lda < K): the sliding windows of a 1-d convolution over a channel-last image with zero halos
    T2_REQUIRE(d.win_Tp == 0 || (d.win_T > 0 && d.win_T <= d.win_Tp && d.splitk == 1), "gemm16: bad window geometry");
    T2_REQUIRE(d.win_Tp > 0 || d.lda >= d.K, "gemm16: lda must be >= K unless the rows are windows");
    if (d.splitk < 1) d.splitk = 1;
    T2_REQUIRE(d.splitk == 1 || (!d.bias && !d.accumulate), "gemm16: split-K needs a plain epilogue");
    const int nkt = d.K / G16_BK;
    p.ktiles_per_split = t2_cdiv(nkt, d.splitk);
    dim3 grid(t2_cdiv(d.N, G16_T), t2_cdiv(d.M, G16_T), d.splitk);
    T2_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm16: grid too large");
    T2_LAUNCH(gemm16_tn_kernel, grid, dim3(G16_NT), 0, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// K-MAJOR operands: C[M][N] (f32) = sum_k A[k][m] . B[k][n], A = [K][lda] and B = [K][ldb] bf16 with m / n contiguous -- the
// layout every slab of the time loops and every channel-last activation image already has ([To.B][.], [(b, t)][channel]),
// so the weight-gradient products dW = dG^T . X need no transposed copies (round 3; before, transpose_cast16_kernel made
// K-contiguous images of both operands first: ~0.5 ms per training step of pure copying).
//
// Same tile (256 x 256 x 64, 8 waves 2 x 4, 32x32x16 MFMA, two LDS stages, LDS-DMA) as gemm16_tn_kernel; what differs is the
// LDS image and the fragment read:
//   * an operand tile is 64 k-rows x 512 B (256 m); a wave DMA instruction copies 2 k-rows;
//   * fragments come from gfx950's transposing LDS read: ds_read_b64_tr_b16 -- each 16-lane group reads a [4 k][16 m] block
//     (lane i: 8 bytes = m 4 (i & 3) .. +3 of row k + (i >> 2)) and lane i receives the four k values of column i.  Two of them
//     (k + 0..3, k + 4..7) are the 8 bf16 of a v_mfma_f32_32x32x16_bf16 operand (lanes 0-31: k 0..7, lanes 32-63: k 8..15);
//   * bank conflicts are removed on the DMA SOURCE: slot s (16 B) of row k holds global chunk s ^ (4 (k & 3)); the 32 lanes a
//     transposing read services together (4 rows x 64 B) then cover all 16 slots of a 256-byte bank row;
//   * rows may overlap (ldb < N): B[k][n] = img[k Ci + n], n < taps Ci, is the sliding window of a 1-d convolution over a
//     channel-last bf16 image -- the weight gradient of nn.Conv1d is this product against the output gradient's image;
//   * K need not be a multiple of 64: the rows of the last tile past K are read from row K - 1 (finite) and A's are zeroed
//     in LDS by the wave that fetched them.
// Replaces (bf16 mode): autograd's dW of the two decoder nn.LSTMCell (reference model.py:352-371) and of every nn.Conv1d of
// encoder and postnet whose channel count is a multiple of 8 (model.py:141-146, 174-175; layers.py:37-39).
// ---------------------------------------------------------------------------------------
typedef float g16_f32x2 __attribute__((ext_vector_type(2)));

// Up to four products share one launch (same M, same split count): the column tiles of problem j follow those of problem
// j - 1 in the grid, so that the workgroups an XCD runs at once multiply the SAME rows of A against different B's -- the
// three input blocks of an LSTM's weight gradient read dG once instead of three times (a 256-column block alone is bound
// by streaming its 456 MB of dG, not by the MFMA).
#define G16_GROUP 4
struct Gemm16GroupParams {
    t2amd_gemm16_desc d[G16_GROUP];
    int coltile_end[G16_GROUP];      // running count of column tiles
    int count;
};

__global__ __launch_bounds__(G16_NT) void gemm16_kk_kernel(Gemm16GroupParams p) {
    // [A stage 0 | A stage 1 | B stage 0 | B stage 1], 32 KB each = 64 k-rows x 512 B
    __shared__ __attribute__((aligned(16))) char smem[4 * G16_IMG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;

    const G16Tile tile = g16_tile_id();
    const int split = tile.bz;
    int prob = 0;
#pragma unroll
    for (int j = 0; j + 1 < G16_GROUP; ++j)
        if (j + 1 < p.count && tile.bx >= p.coltile_end[j]) prob = j + 1;
    const int bx = tile.bx - (prob > 0 ? p.coltile_end[prob - 1] : 0);
    // the problem's fields, read once (a reference into the kernel-argument array would reload them at every use)
    struct { const void* A; const void* B; float* C; int M, N, K; long long lda, ldb, ldc; long long strideSplitC; int accumulate;
             const float* bias; int splitk; } d;
    d.A = p.d[prob].A; d.B = p.d[prob].B; d.C = p.d[prob].C; d.M = p.d[prob].M; d.N = p.d[prob].N; d.K = p.d[prob].K;
    d.lda = p.d[prob].lda; d.ldb = p.d[prob].ldb; d.ldc = p.d[prob].ldc; d.strideSplitC = p.d[prob].strideSplitC;
    d.accumulate = p.d[prob].accumulate; d.bias = p.d[prob].bias; d.splitk = p.d[prob].splitk;
    const int row0 = tile.by * G16_T, col0 = bx * G16_T;
    const int M = d.M, N = d.N;
    const int nkt_all = (d.K + G16_BK - 1) / G16_BK;
    const int per_split = (nkt_all + d.splitk - 1) / d.splitk;
    const int kt_beg = split * per_split;
    int kt_end = kt_beg + per_split;
    if (kt_end > nkt_all) kt_end = nkt_all;
    const int nk = kt_end > kt_beg ? kt_end - kt_beg : 0;
    const int rem = d.K - (nkt_all - 1) * G16_BK;                       // rows of the last k tile (1..64)
    const int ztile = (kt_end == nkt_all && rem < G16_BK) ? nk - 1 : -1; // this split's tile with rows past K, if any

    // ---- DMA sources: instruction i of this wave fills rows 16 i + 2 wave + (lane >> 5), slot lane & 31 ----
    const int rsub = 2 * wave + lhi;
    const long long arow = d.lda * 2, brow = d.ldb * 2;                 // bytes per k-row
    const char* asrc[4];
    const char* bsrc[4];
    {
        const int cl = l31 ^ ((rsub & 3) << 2);                         // logical 16-byte chunk held by slot lane & 31
        int ma = row0 + 8 * cl, nb = col0 + 8 * cl;
        ma = ma + 8 <= M ? ma : M - 8;                                  // clamped columns: their products are never stored
        nb = nb + 8 <= N ? nb : N - 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long k = (long long)kt_beg * G16_BK + 16 * i + rsub;
            asrc[i] = reinterpret_cast<const char*>(d.A) + k * arow + (long long)ma * 2;
            bsrc[i] = reinterpret_cast<const char*>(d.B) + k * brow + (long long)nb * 2;
        }
    }
#define KK_ISSUE(STAGE, KT)                                                                                      \
    {                                                                                                            \
        char* ad_ = smem + (STAGE) * G16_IMG + wave * 1024;                                                      \
        char* bd_ = smem + (2 + (STAGE)) * G16_IMG + wave * 1024;                                                \
        if ((KT) != ztile) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
                __builtin_amdgcn_global_load_lds((g16_gptr)(asrc[i]), (g16_lptr)(ad_ + i * 8192), 16, 0, 0);     \
                __builtin_amdgcn_global_load_lds((g16_gptr)(bsrc[i]), (g16_lptr)(bd_ + i * 8192), 16, 0, 0);     \
                asrc[i] += 64 * arow; bsrc[i] += 64 * brow;                                                      \
            }                                                                                                    \
        } else {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
                const int over = 16 * i + rsub - (rem - 1);          /* rows past the last one: read that one */ \
                const char* a_ = over > 0 ? asrc[i] - over * arow : asrc[i];                                     \
                const char* b_ = over > 0 ? bsrc[i] - over * brow : bsrc[i];                                     \
                __builtin_amdgcn_global_load_lds((g16_gptr)(a_), (g16_lptr)(ad_ + i * 8192), 16, 0, 0);          \
                __builtin_amdgcn_global_load_lds((g16_gptr)(b_), (g16_lptr)(bd_ + i * 8192), 16, 0, 0);          \
            }                                                                                                    \
        }                                                                                                        \
    }
    // the wave that fetched a row past K zeroes A's copy of it once its own DMA has landed (B's is finite: 0 x finite = 0)
#define KK_LANDED(STAGE, KT)                                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    if ((KT) == ztile) {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            if (16 * i + rsub >= rem)                                                                            \
                *reinterpret_cast<f32x4*>(smem + (STAGE) * G16_IMG + i * 8192 + wave * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f}; \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    }                                                                                                            \
    __builtin_amdgcn_s_barrier();

    // ---- fragment addresses.  Lane = (group g = lane >> 4, i = lane & 15): row 8 (g >> 1) + (i >> 2) of a k-step, logical
    //      chunk (tile's first) + 2 (g & 1) + ((i & 3) >> 1), byte 8 (i & 1) in it; physical slot = chunk ^ 4 (row & 3).
    //      k-step ks adds 16 rows (8192 B), the second read of a fragment 4 rows (2048 B), the stage 32768 B: immediates. ----
    unsigned aaddr[4], baddr[2];
    {
        const unsigned base = (unsigned)reinterpret_cast<size_t>((g16_lptr)(smem));
        const int g = lane >> 4, i = lane & 15;
        const int r = 8 * (g >> 1) + (i >> 2);
        const int sw = (i >> 2) << 2;
        const int sub = 2 * (g & 1) + ((i & 3) >> 1);
        const unsigned in = (unsigned)(r * 512 + (i & 1) * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) aaddr[t] = base + in + (unsigned)((((wm * 16 + 4 * t + sub) ^ sw)) << 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) baddr[t] = base + 2 * G16_IMG + in + (unsigned)((((wn * 8 + 4 * t + sub) ^ sw)) << 4);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment halves: [tile][0] = k + 0..3, [tile][1] = k + 4..7
    g16_f32x2 fa[4][2], fb[2][2], ga[4][2], gb[2][2];
#define KK_READ(STAGE, KS, A_, B_)                                                                               \
    asm volatile(                                                                                                \
        "ds_read_b64_tr_b16 %0, %12 offset:%18\n\t"                                                              \
        "ds_read_b64_tr_b16 %1, %12 offset:%19\n\t"                                                              \
        "ds_read_b64_tr_b16 %8, %16 offset:%18\n\t"                                                              \
        "ds_read_b64_tr_b16 %9, %16 offset:%19\n\t"                                                              \
        "ds_read_b64_tr_b16 %2, %13 offset:%18\n\t"                                                              \
        "ds_read_b64_tr_b16 %3, %13 offset:%19\n\t"                                                              \
        "ds_read_b64_tr_b16 %10, %17 offset:%18\n\t"                                                             \
        "ds_read_b64_tr_b16 %11, %17 offset:%19\n\t"                                                             \
        "ds_read_b64_tr_b16 %4, %14 offset:%18\n\t"                                                              \
        "ds_read_b64_tr_b16 %5, %14 offset:%19\n\t"                                                              \
        "ds_read_b64_tr_b16 %6, %15 offset:%18\n\t"                                                              \
        "ds_read_b64_tr_b16 %7, %15 offset:%19"                                                                  \
        : "=&v"(A_[0][0]), "=&v"(A_[0][1]), "=&v"(A_[1][0]), "=&v"(A_[1][1]), "=&v"(A_[2][0]), "=&v"(A_[2][1]),  \
          "=&v"(A_[3][0]), "=&v"(A_[3][1]), "=&v"(B_[0][0]), "=&v"(B_[0][1]), "=&v"(B_[1][0]), "=&v"(B_[1][1])   \
        : "v"(aaddr[0]), "v"(aaddr[1]), "v"(aaddr[2]), "v"(aaddr[3]), "v"(baddr[0]), "v"(baddr[1]),              \
          "i"((STAGE) * G16_IMG + (KS) * 8192), "i"((STAGE) * G16_IMG + (KS) * 8192 + 2048)                      \
        : "memory");
#define KK_WAIT(A_, B_)                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                 : "+v"(A_[0][0]), "+v"(A_[0][1]), "+v"(A_[1][0]), "+v"(A_[1][1]), "+v"(A_[2][0]), "+v"(A_[2][1]), \
                   "+v"(A_[3][0]), "+v"(A_[3][1]), "+v"(B_[0][0]), "+v"(B_[0][1]), "+v"(B_[1][0]), "+v"(B_[1][1]) : : "memory");
#define KK_OP(X_) __builtin_bit_cast(g16_bf16x8, __builtin_shufflevector(X_[0], X_[1], 0, 1, 2, 3))
#define KK_MFMA(A_, B_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KK_OP(A_[i]), KK_OP(B_[j]), acc[i][j], 0, 0, 0);
#define KK_TILE(STAGE)                                                                                           \
    {                                                                                                            \
        KK_READ(STAGE, 0, fa, fb) KK_WAIT(fa, fb)                                                                \
        KK_READ(STAGE, 1, ga, gb) __builtin_amdgcn_sched_barrier(0); KK_MFMA(fa, fb) __builtin_amdgcn_sched_barrier(0); KK_WAIT(ga, gb) \
        KK_READ(STAGE, 2, fa, fb) __builtin_amdgcn_sched_barrier(0); KK_MFMA(ga, gb) __builtin_amdgcn_sched_barrier(0); KK_WAIT(fa, fb) \
        KK_READ(STAGE, 3, ga, gb) __builtin_amdgcn_sched_barrier(0); KK_MFMA(fa, fb) __builtin_amdgcn_sched_barrier(0); KK_WAIT(ga, gb) \
        KK_MFMA(ga, gb)                                                                                          \
    }

    if (nk > 0) {
        KK_ISSUE(0, 0)
        KK_LANDED(0, 0)
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            KK_ISSUE(1, kt + 1)                         // tile kt + 1 in flight while tile kt is multiplied
            KK_TILE(0)
            KK_LANDED(1, kt + 1)
            if (kt + 2 < nk) KK_ISSUE(0, kt + 2)
            KK_TILE(1)
            KK_LANDED(0, kt + 2)
        }
        if (nk & 1) KK_TILE(0)
    }
#undef KK_ISSUE
#undef KK_LANDED
#undef KK_READ
#undef KK_WAIT
#undef KK_OP
#undef KK_MFMA
#undef KK_TILE

    // ---- epilogue: D layout of the 32 x 32 MFMA: lane -> column l31, register r -> row (r & 3) + 8 (r >> 2) + 4 lhi ----
    float* __restrict__ C = d.C + (long long)split * d.strideSplitC;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int gn = col0 + wn * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = row0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (gm < M && gn < N) {
                    float* cp = C + (long long)gm * d.ldc + gn;
                    float val = acc[tm][tn][r];
                    if (d.bias) val += d.bias[gn];
                    if (d.accumulate) val += *cp;
                    *cp = val;
                }
            }
        }
    }
}

extern "C" int t2amd_gemm16_kk_group(const t2amd_gemm16_desc* dp, int count, void* stream) {
    T2_REQUIRE(dp != nullptr && count >= 1 && count <= G16_GROUP, "gemm16_kk: 1..4 descriptors");
    Gemm16GroupParams p;
    p.count = count;
    int coltiles = 0;
    for (int j = 0; j < G16_GROUP; ++j) {
        p.d[j] = dp[j < count ? j : count - 1];
        if (j >= count) { p.coltile_end[j] = coltiles; continue; }
        t2amd_gemm16_desc& d = p.d[j];
        T2_REQUIRE(d.A && d.B && d.C, "gemm16_kk: null operand");
        T2_REQUIRE(d.M >= 8 && d.N >= 8 && d.K > 0 && d.M % 8 == 0 && d.N % 8 == 0, "gemm16_kk: M and N must be positive multiples of 8, K positive");
        T2_REQUIRE(t2_aligned16(d.A) && t2_aligned16(d.B) && d.lda > 0 && d.ldb > 0 && d.lda % 8 == 0 && d.ldb % 8 == 0,
                   "gemm16_kk: operands must be 16-byte aligned with row strides that are multiples of 8 elements");
        T2_REQUIRE(d.win_Tp == 0 && d.win_T == 0, "gemm16_kk: no window epilogue (overlapping rows are expressed by lda / ldb < M / N)");
        if (d.splitk < 1) d.splitk = 1;
        T2_REQUIRE(d.splitk == 1 || (!d.bias && !d.accumulate), "gemm16_kk: split-K needs a plain epilogue");
        T2_REQUIRE(d.M == p.d[0].M && d.splitk == p.d[0].splitk, "gemm16_kk: the products of one launch share M and the split count");
        coltiles += t2_cdiv(d.N, G16_T);
        p.coltile_end[j] = coltiles;
    }
    dim3 grid(coltiles, t2_cdiv(p.d[0].M, G16_T), p.d[0].splitk);
    T2_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm16_kk: grid too large");
    T2_LAUNCH(gemm16_kk_kernel, grid, dim3(G16_NT), 0, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_gemm16_kk(const t2amd_gemm16_desc* dp, void* stream) { return t2amd_gemm16_kk_group(dp, 1, stream); }

// ---------------------------------------------------------------------------------------
// Transposing cast: dst[c][r] (bf16, row stride ldd >= rows, columns rows..rpad-1 zeroed) = src[r][c], src f32 or bf16.
// The K-contiguous bf16 image of a [To.B][C] slab for the weight-gradient products above (K = To.B rounded up to 64).
// 64 x 64 tiles through LDS: 256-byte reads along c, 128-byte writes along r.
// ---------------------------------------------------------------------------------------
template <bool SRC16>
__global__ __launch_bounds__(256) void transpose_cast16_kernel(const void* __restrict__ src, long long lds_, unsigned short* __restrict__ dst,
                                                               long long ldd, int rows, int cols, int rpad) {
    // tile[r][c], row stride 66 shorts = 33 dwords: the column reads of the write pass fall on distinct banks
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    // read pass: thread -> (row i = tid >> 4 (+16 per pass), 4 consecutive columns 4 (tid & 15)): 8 bytes of a bf16 source,
    // 16 of an f32 one; a row of the tile is one 128 / 256-byte segment
    const bool vec_in = SRC16 ? (lds_ % 4 == 0 && (reinterpret_cast<size_t>(src) & 7) == 0) : (lds_ % 4 == 0 && (reinterpret_cast<size_t>(src) & 15) == 0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = pass * 16 + (tid >> 4), j = 4 * (tid & 15);
        const int r = r0 + i, c = c0 + j;
        unsigned short v[4] = {0, 0, 0, 0};
        if (r < rows) {
            if (vec_in && c + 3 < cols) {
                if (SRC16) {
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(src) + (long long)r * lds_ + c);
                    v[0] = (unsigned short)(u.x & 0xffffu); v[1] = (unsigned short)(u.x >> 16);
                    v[2] = (unsigned short)(u.y & 0xffffu); v[3] = (unsigned short)(u.y >> 16);
                } else {
                    const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (long long)r * lds_ + c);
                    v[0] = t2_f32_to_bf16(f.x); v[1] = t2_f32_to_bf16(f.y); v[2] = t2_f32_to_bf16(f.z); v[3] = t2_f32_to_bf16(f.w);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols)
                        v[e] = SRC16 ? reinterpret_cast<const unsigned short*>(src)[(long long)r * lds_ + c + e]
                                     : t2_f32_to_bf16(reinterpret_cast<const float*>(src)[(long long)r * lds_ + c + e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[i][j + e] = v[e];
    }
    __syncthreads();
    // write pass: thread -> (column i of the tile = row of dst, 4 consecutive r): one 8-byte store, 128 bytes per dst row
    const bool vec_out = ldd % 4 == 0 && (reinterpret_cast<size_t>(dst) & 7) == 0;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = pass * 16 + (tid >> 4), j = 4 * (tid & 15);
        const int c = c0 + i, r = r0 + j;
        if (c >= cols || r >= rpad) continue;
        unsigned short* o = dst + (long long)c * ldd + r;
        if (vec_out && r + 3 < rpad) {
            uint2 u;
            u.x = (unsigned)tile[j][i] | ((unsigned)tile[j + 1][i] << 16);
            u.y = (unsigned)tile[j + 2][i] | ((unsigned)tile[j + 3][i] << 16);
            *reinterpret_cast<uint2*>(o) = u;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r + e < rpad) o[e] = tile[j + e][i];
        }
    }
}

// dst[(b Tp + pad + t)][c] (bf16) = src[(b T + t)][c] (f32): the channel-last image with `pad` zero rows around every utterance
// that the window mode above reads.  The kernel walks the IMAGE rows -- all (rows / T) * Tp of them plus the 2 * pad rows
// the last utterance's windows reach into -- and writes the zero halos itself (round 3: the image no longer has to be
// zero-filled first, one pass instead of two over it).
__global__ __launch_bounds__(256) void cast_halo16_kernel(const float* __restrict__ src, long long lds_, unsigned short* __restrict__ dst,
                                                          long long img_rows, long long nb, int C4, int T, int Tp, int pad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= img_rows * C4) return;
    const long long R = i / C4;
    const int c4 = (int)(i - R * C4);
    const long long b = R / Tp;
    const int t = (int)(R - b * Tp) - pad;
    uint2 o = make_uint2(0u, 0u);
    if (b < nb && t >= 0 && t < T) {
        const float4 v = *reinterpret_cast<const float4*>(src + (b * T + t) * lds_ + c4 * 4);
        o.x = (unsigned)t2_f32_to_bf16(v.x) | ((unsigned)t2_f32_to_bf16(v.y) << 16);
        o.y = (unsigned)t2_f32_to_bf16(v.z) | ((unsigned)t2_f32_to_bf16(v.w) << 16);
    }
    *reinterpret_cast<uint2*>(dst + (R * (long long)(C4 * 4)) + c4 * 4) = o;
}

// bf16 weight images of nn.Conv1d (W f32 [Co][Ci][k], torch layout) for the window products above, one launch each, columns
// k C .. Kp - 1 zero (Kp = the row length rounded up to the 64-deep k-steps; the windows' overhang meets zeros):
//   forward   out[co][tap Ci + ci]           = W[co][ci][tap]          (reversed = 0, rows = Co, inner = Ci)
//   dgrad     out[ci][(k - 1 - tap) Co + co] = W[co][ci][tap]          (reversed = 1, rows = Ci, inner = Co)
// Replaces a transpose + cast (forward) and an ATen flip / permute / contiguous / cast chain (data gradient) per layer and
// optimiser step.
__global__ __launch_bounds__(256) void pack_conv16_kernel(const float* __restrict__ W, unsigned short* __restrict__ out, int Co, int Ci,
                                                          int k, int Kp, int reversed) {
    const int rows = reversed ? Ci : Co, inner = reversed ? Co : Ci;
    const long long n = (long long)rows * Kp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / Kp), c = (int)(i - (long long)r * Kp);
        unsigned short v = 0;
        if (c < k * inner) {
            const int tap = c / inner, j = c - tap * inner;
            const int co = reversed ? j : r, ci = reversed ? r : j, t = reversed ? k - 1 - tap : tap;
            v = t2_f32_to_bf16(W[((long long)co * Ci + ci) * k + t]);
        }
        out[i] = v;
    }
}

extern "C" int t2amd_pack_conv_bf16(const float* W, void* out, int Co, int Ci, int k, int Kp, int reversed, void* stream) {
    T2_REQUIRE(W && out && Co > 0 && Ci > 0 && k > 0 && Kp >= k * (reversed ? Co : Ci), "pack_conv: bad arguments");
    const long long n = (long long)(reversed ? Ci : Co) * Kp;
    int blocks = t2_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(pack_conv16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, (unsigned short*)out, Co, Ci, k, Kp, reversed);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_cast_halo_bf16(const float* src, long long lds_, void* dst, long long rows, int C, int T, int pad, void* stream) {
    T2_REQUIRE(src && dst && rows > 0 && C > 0 && C % 4 == 0 && T > 0 && rows % T == 0 && pad >= 0 && lds_ % 4 == 0 &&
                   t2_aligned16(src) && t2_aligned16(dst),
               "cast_halo: rows must be whole utterances of T, C a multiple of 4, 16-byte aligned operands");
    const long long nb = rows / T, img_rows = nb * (T + 2 * pad) + 2 * pad;     // dst holds img_rows rows of C bf16
    const long long n = img_rows * (C / 4);
    T2_REQUIRE((n + 255) / 256 < (1ll << 31), "cast_halo: too large");
    T2_LAUNCH(cast_halo16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds_, (unsigned short*)dst,
              img_rows, nb, C / 4, T, T + 2 * pad, pad);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_transpose_cast_bf16(const void* src, int src_is_bf16, long long lds_, void* dst, long long ldd, int rows, int cols,
                                         int rows_padded, void* stream) {
    T2_REQUIRE(src && dst && rows > 0 && cols > 0 && rows_padded >= rows && ldd >= rows_padded, "transpose_cast: bad arguments");
    dim3 grid(t2_cdiv(rows_padded, 64), t2_cdiv(cols, 64));
    T2_REQUIRE(grid.y <= 65535, "transpose_cast: too many columns");
    if (src_is_bf16)
        T2_LAUNCH((transpose_cast16_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, src, lds_, (unsigned short*)dst, ldd, rows, cols, rows_padded);
    else
        T2_LAUNCH((transpose_cast16_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, src, lds_, (unsigned short*)dst, ldd, rows, cols, rows_padded);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
