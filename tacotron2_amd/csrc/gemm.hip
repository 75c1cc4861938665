// Dense / implicit-conv GEMM for gfx950 on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Tile: 128 x 128 x 16, 256 threads = 4 waves (2 x 2), each wave a 64 x 64 block of
// 2 x 2 MFMA tiles (4 x 16 accumulator registers).  Both operands are staged k-major in
// LDS ([k][m], row stride 132 floats) so that the MFMA fragment read (lane -> one f32 at
// [k = lane>>5][i = lane&31]) is 32 consecutive floats per half-wave: conflict-free.
// K-contiguous operands are transposed on the LDS store (2-way ds_write_b32 conflicts are
// free on CDNA4), M/N-contiguous operands go straight in with ds_write_b128.
// Register double-buffering: tile k+1 is in flight from HBM while tile k is multiplied.
//
// Replaces: every nn.Linear / nn.Conv1d forward + dgrad + wgrad on the Tacotron 2 hot path
// (reference layers.py:17-18, 37-39 and their call sites in model.py; see include/tacotron2_amd.h).
#include "common.h"

#define GBM 128
#define GBN 128
#define GBK 16
#define GLD 132

struct GemmParams {
    t2amd_gemm_desc d;
    int avec, bvec;
    int ktiles_per_split;
};

// XCD-aware tile order.  Workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest), and each
// XCD has its own 4 MB L2: with the plain blockIdx -> tile map every XCD walks ALL row tiles and re-fetches the whole
// A operand (measured on the 4096x2560x55k wgrad: L2 hit rate 60 %, 14 GB of fabric reads for 1.5 GB of operands).
// Here XCD j takes a contiguous run of logical tile ids, and ids are laid out in groups of 8 row tiles x all
// column tiles (column-major inside the group), so the ~96 workgroups an XCD runs at once form an ~8 x 12 block of
// tiles that shares 8 A tiles and 12 B tiles per k step.
struct TileId { int bx, by, bz; };
__device__ __forceinline__ TileId xcd_tile_id() {
    const int nbx = gridDim.x, nby = gridDim.y;
    const int total = nbx * nby * (int)gridDim.z;
    const int L = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
    const int xcd = L & 7, q = L >> 3;
    const int base = total >> 3, rem = total & 7;
    const int id = xcd * base + (xcd < rem ? xcd : rem) + q;      // bijection for any total
    const int per_z = nbx * nby;
    TileId t;
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

template <bool AK, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK][GLD];

    const t2amd_gemm_desc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const TileId tile = xcd_tile_id();
    const int z = tile.bz;
    const int bidx = z / d.splitk;
    const int split = z - bidx * d.splitk;
    const float* __restrict__ A = d.A + (long long)bidx * d.strideA;
    const float* __restrict__ B = d.B + (long long)bidx * d.strideB;
    float* __restrict__ C = d.C + (long long)bidx * d.strideC + (long long)split * d.strideSplitC;

    const int row0 = tile.by * GBM;
    const int col0 = tile.bx * GBN;
    const int M = d.M, N = d.N;
    const int kbeg = split * p.ktiles_per_split * GBK;
    int kend = kbeg + p.ktiles_per_split * GBK;
    if (kend > d.K) kend = d.K;
    const int nk = (kend > kbeg) ? (kend - kbeg + GBK - 1) / GBK : 0;

    // ---- per-thread load coordinates -------------------------------------------------
    // K-contiguous operand: 128 rows x 16 k = 512 float4, f = tid + 256*i -> (row = f>>2, kq = f&3)
    // M-contiguous operand: 16 k x 128 m  = 512 float4, f -> (kk = f>>5, m4 = f&31)
    int a_t[2] = {0, 0};          // conv: t index of the thread's rows
    if (AK && d.convA_T > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = row0 + ((tid + 256 * i) >> 2);
            a_t[i] = r % d.convA_T;
        }
    }
    int b_tap[2] = {0, 0}, b_ci[2] = {0, 0};
    if (!BKC && d.convB_T > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gn = col0 + ((tid + 256 * i) & 31) * 4;
            b_tap[i] = gn / d.convB_C;
            b_ci[i] = gn - b_tap[i] * d.convB_C;
        }
    }

    float4 ra[2], rb[2];

    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AK) {
                const int r = f >> 2, kq = f & 3;
                const int gm = row0 + r;
                const int gk = k0 + kq * 4;
                if (gm < M && gk < kend) {
                    long long srow = gm;
                    int col = gk;
                    bool ok = true;
                    if (d.convA_T > 0) {
                        const int tap = k0 / d.convA_C;
                        col = (k0 - tap * d.convA_C) + kq * 4;
                        const int sh = (tap - d.convA_pad) * d.convA_sign;
                        const int tt = a_t[i] + sh;
                        ok = (tt >= 0) && (tt < d.convA_T);
                        srow = (long long)gm + sh;
                    }
                    if (ok) {
                        const float* src = A + srow * d.lda + col;
                        if (p.avec) {
                            v = *reinterpret_cast<const float4*>(src);
                        } else {
                            v.x = src[0];
                            if (gk + 1 < kend) v.y = src[1];
                            if (gk + 2 < kend) v.z = src[2];
                            if (gk + 3 < kend) v.w = src[3];
                        }
                    }
                }
            } else {
                const int kk = f >> 5, m4 = f & 31;
                const int gk = k0 + kk;
                const int gm = row0 + m4 * 4;
                if (gk < kend && gm < M) {
                    const float* src = A + (long long)gk * d.lda + gm;
                    if (p.avec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (gm + 1 < M) v.y = src[1];
                        if (gm + 2 < M) v.z = src[2];
                        if (gm + 3 < M) v.w = src[3];
                    }
                }
            }
            ra[i] = v;
        }
    };

    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BKC) {
                const int r = f >> 2, kq = f & 3;
                const int gn = col0 + r;
                const int gk = k0 + kq * 4;
                if (gn < N && gk < kend) {
                    const float* src = B + (long long)gn * d.ldb + gk;
                    if (p.bvec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (gk + 1 < kend) v.y = src[1];
                        if (gk + 2 < kend) v.z = src[2];
                        if (gk + 3 < kend) v.w = src[3];
                    }
                }
            } else {
                const int kk = f >> 5, n4 = f & 31;
                const int gk = k0 + kk;
                const int gn = col0 + n4 * 4;
                if (gk < kend && gn < N) {
                    long long srow = gk;
                    int col = gn;
                    bool ok = true;
                    if (d.convB_T > 0) {
                        const int sh = b_tap[i] - d.convB_pad;
                        const int tt = (gk % d.convB_T) + sh;
                        ok = (tt >= 0) && (tt < d.convB_T);
                        srow = (long long)gk + sh;
                        col = b_ci[i];
                    }
                    if (ok) {
                        const float* src = B + srow * d.ldb + col;
                        if (p.bvec) {
                            v = *reinterpret_cast<const float4*>(src);
                        } else {
                            v.x = src[0];
                            if (gn + 1 < N) v.y = src[1];
                            if (gn + 2 < N) v.z = src[2];
                            if (gn + 3 < N) v.w = src[3];
                        }
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            if (AK) {
                const int r = f >> 2, kq = f & 3;
                As[buf][kq * 4 + 0][r] = ra[i].x;
                As[buf][kq * 4 + 1][r] = ra[i].y;
                As[buf][kq * 4 + 2][r] = ra[i].z;
                As[buf][kq * 4 + 3][r] = ra[i].w;
            } else {
                const int kk = f >> 5, m4 = f & 31;
                *reinterpret_cast<float4*>(&As[buf][kk][m4 * 4]) = ra[i];
            }
            if (BKC) {
                const int r = f >> 2, kq = f & 3;
                Bs[buf][kq * 4 + 0][r] = rb[i].x;
                Bs[buf][kq * 4 + 1][r] = rb[i].y;
                Bs[buf][kq * 4 + 2][r] = rb[i].z;
                Bs[buf][kq * 4 + 3][r] = rb[i].w;
            } else {
                const int kk = f >> 5, n4 = f & 31;
                *reinterpret_cast<float4*>(&Bs[buf][kk][n4 * 4]) = rb[i];
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_a(kbeg);
        load_b(kbeg);
        store_lds(0);
    }
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_a(kbeg + (kt + 1) * GBK);
            load_b(kbeg + (kt + 1) * GBK);
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            const int krow = kk * 2 + lhi;
            const float a0 = As[cur][krow][wm * 64 + l31];
            const float a1 = As[cur][krow][wm * 64 + 32 + l31];
            const float b0 = Bs[cur][krow][wn * 64 + l31];
            const float b1 = Bs[cur][krow][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue --------------------------------------------------------------------
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int gn = col0 + wn * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int gm = row0 + wm * 64 + tm * 32 + row;
                if (gm < M && gn < N) {
                    float val = acc[tm][tn][r];
                    float* cp = C + (long long)gm * d.ldc + gn;
                    if (d.bias) val += d.bias[gn];
                    if (d.accumulate) val += *cp;
                    if (d.act == 1) val = fmaxf(val, 0.f);
                    if (d.keep) val = d.keep[(long long)gm * d.ldkeep + gn] ? val * d.keep_scale : 0.f;
                    *cp = val;
                }
            }
        }
    }
}

template <bool AK, bool BKC, bool X3, int TB>
__global__ __launch_bounds__(2 * TB) void gemm_bf16x3_kernel(GemmParams p);

// Output tile edge the library will use for an (M, N) product at this precision / operand layout when `nz` = batch *
// splitk slices of it are launched: callers that split K size the split from it.  256 only for the bf16 kernels, on outputs
// at least two 256-tiles wide both ways, and only when the launch still has >= 192 workgroups (a 256-tile workgroup
// owns a whole CU).  T2AMD_GEMM_TILE=128 forces the small tile (A/B measurements).
extern "C" int t2amd_gemm_tile_size(int M, int N, int precision, int nz, int a_kcontig, int b_kcontig) {
    static const bool force128 = [] { const char* e = getenv("T2AMD_GEMM_TILE"); return e && atoi(e) == 128; }();
    if (force128 || precision < 1 || M < 512 || N < 512) return 128;
    // split-bf16 (precision 1) keeps hi and lo images: 256-tiles fit the 160 KB of LDS only when both operands are
    // M/N-contiguous (unpadded pair-interleaved images, 128 KB) -- the weight-gradient layout
    if (precision == 1 && (a_kcontig || b_kcontig)) return 128;
    const long long wgs = (long long)t2_cdiv(M, 256) * t2_cdiv(N, 256) * (nz > 0 ? nz : 1);
    return wgs >= 192 ? 256 : 128;
}

extern "C" int t2amd_gemm_f32(const t2amd_gemm_desc* dp, void* stream) {
    T2_REQUIRE(dp != nullptr, "gemm: null descriptor");
    GemmParams p;
    p.d = *dp;
    t2amd_gemm_desc& d = p.d;
    T2_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    T2_REQUIRE(d.M > 0 && d.N > 0 && d.K >= 0, "gemm: bad dims");
    if (d.batch < 1) d.batch = 1;
    if (d.splitk < 1) d.splitk = 1;
    if (d.convA_T > 0) {
        T2_REQUIRE(d.a_kcontig == 1, "gemm: convA needs K-contiguous A");
        T2_REQUIRE(d.convA_C % GBK == 0 && d.K % d.convA_C == 0, "gemm: convA_C must be a multiple of 16 dividing K");
        T2_REQUIRE(d.convA_sign == 1 || d.convA_sign == -1, "gemm: convA_sign must be +-1");
        T2_REQUIRE(d.M % d.convA_T == 0, "gemm: convA rows must be whole utterances");
    }
    if (d.convB_T > 0) {
        T2_REQUIRE(d.b_kcontig == 0, "gemm: convB needs N-contiguous B");
        T2_REQUIRE(d.convB_C % 4 == 0 && d.N % d.convB_C == 0, "gemm: convB_C must be a multiple of 4 dividing N");
        T2_REQUIRE(d.K % d.convB_T == 0, "gemm: convB rows must be whole utterances");
    }
    if (d.splitk > 1) {
        T2_REQUIRE(!d.bias && !d.keep && d.act == 0 && !d.accumulate, "gemm: split-K needs a plain epilogue");
    }
    T2_REQUIRE(d.act == 0 || d.act == 1, "gemm: act must be 0 or 1");
    // vector-load eligibility
    if (d.a_kcontig) {
        const int kdim = d.convA_T > 0 ? d.convA_C : d.K;
        p.avec = (kdim % 4 == 0) && (d.lda % 4 == 0) && t2_aligned16(d.A) && (d.strideA % 4 == 0);
    } else {
        p.avec = (d.M % 4 == 0) && (d.lda % 4 == 0) && t2_aligned16(d.A) && (d.strideA % 4 == 0);
    }
    if (d.b_kcontig) {
        p.bvec = (d.K % 4 == 0) && (d.ldb % 4 == 0) && t2_aligned16(d.B) && (d.strideB % 4 == 0);
    } else {
        const int ndim = d.convB_T > 0 ? d.convB_C : d.N;
        p.bvec = (ndim % 4 == 0) && (d.ldb % 4 == 0) && t2_aligned16(d.B) && (d.strideB % 4 == 0);
    }
    T2_REQUIRE(d.precision >= 0 && d.precision <= 2, "gemm: precision must be 0 (exact f32), 1 (split-bf16 x3) or 2 (bf16)");
    // the bf16 kernels step K by 32: implicit-conv channel counts must be multiples of 32
    const bool fast = d.precision >= 1 && (d.convA_T == 0 || d.convA_C % 32 == 0);
    const int bk = fast ? 32 : GBK;
    const int nkt = t2_cdiv(d.K, bk);
    p.ktiles_per_split = t2_cdiv(nkt > 0 ? nkt : 1, d.splitk);
    dim3 grid(t2_cdiv(d.N, GBN), t2_cdiv(d.M, GBM), d.batch * d.splitk);
    T2_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
    hipStream_t s = (hipStream_t)stream;
    // (round 6) The 256-tile split-bf16 branch must stand IN FRONT of the 128-tile one: since the round it was written in it stood
    // behind it -- behind an unconditional return -- and never ran, while the engine's split-K policy (engine._choose_splitk via
    // t2amd_gemm_tile_size) sized the weight-gradient launches for 256-tiles.  T2AMD_GEMM_TILE=128 keeps the old behaviour (A/B).
    if (fast && d.precision == 1 && t2amd_gemm_tile_size(d.M, d.N, 1, d.batch * d.splitk, d.a_kcontig, d.b_kcontig) == 256) {
        dim3 g2(t2_cdiv(d.N, 256), t2_cdiv(d.M, 256), d.batch * d.splitk);
        static bool attr_set = false;
        if (!attr_set && !t2amd_validate_only_flag_()) {      // 128 KB of static LDS is above the 64 KB default limit
            (void)hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<false, false, true, 256>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 0);
            attr_set = true;
        }
        T2_LAUNCH((gemm_bf16x3_kernel<false, false, true, 256>), g2, dim3(512), 0, s, p);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (fast && d.precision == 1) {
        if (d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, true, true, 128>), grid, dim3(256), 0, s, p);
        else if (d.a_kcontig && !d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, false, true, 128>), grid, dim3(256), 0, s, p);
        else if (!d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<false, true, true, 128>), grid, dim3(256), 0, s, p);
        else
            T2_LAUNCH((gemm_bf16x3_kernel<false, false, true, 128>), grid, dim3(256), 0, s, p);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (fast && d.precision == 2 && t2amd_gemm_tile_size(d.M, d.N, 2, d.batch * d.splitk, d.a_kcontig, d.b_kcontig) == 256) {
        dim3 g2(t2_cdiv(d.N, 256), t2_cdiv(d.M, 256), d.batch * d.splitk);
        if (d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, true, false, 256>), g2, dim3(512), 0, s, p);
        else if (d.a_kcontig && !d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, false, false, 256>), g2, dim3(512), 0, s, p);
        else if (!d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<false, true, false, 256>), g2, dim3(512), 0, s, p);
        else
            T2_LAUNCH((gemm_bf16x3_kernel<false, false, false, 256>), g2, dim3(512), 0, s, p);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (fast) {
        if (d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, true, false, 128>), grid, dim3(256), 0, s, p);
        else if (d.a_kcontig && !d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<true, false, false, 128>), grid, dim3(256), 0, s, p);
        else if (!d.a_kcontig && d.b_kcontig)
            T2_LAUNCH((gemm_bf16x3_kernel<false, true, false, 128>), grid, dim3(256), 0, s, p);
        else
            T2_LAUNCH((gemm_bf16x3_kernel<false, false, false, 128>), grid, dim3(256), 0, s, p);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (d.a_kcontig && d.b_kcontig)
        T2_LAUNCH((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, p);
    else if (d.a_kcontig && !d.b_kcontig)
        T2_LAUNCH((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, p);
    else if (!d.a_kcontig && d.b_kcontig)
        T2_LAUNCH((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, p);
    else
        T2_LAUNCH((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// out[i] (+)= sum_s partials[s][i], optional (co, tap, ci) -> (co, ci, tap) permutation.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, long long stride,
                                     float* __restrict__ out, long long n, int accumulate,
                                     int taps, int ci_dim) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        // slabs four at a time, loads first (same add order as the plain loop, which waits for every slab in turn)
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            const float v0 = part[(long long)k * stride + i], v1 = part[(long long)(k + 1) * stride + i];
            const float v2 = part[(long long)(k + 2) * stride + i], v3 = part[(long long)(k + 3) * stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nsplit; ++k) s += part[(long long)k * stride + i];
        long long o = i;
        if (taps > 0) {
            const long long per_co = (long long)taps * ci_dim;
            const long long co = i / per_co;
            const long long rem = i - co * per_co;
            const long long tap = rem / ci_dim;
            const long long ci = rem - tap * ci_dim;
            o = co * per_co + ci * taps + tap;
        }
        if (accumulate) s += out[o];
        out[o] = s;
    }
}

extern "C" int t2amd_splitk_reduce_f32(const float* partials, int nsplit, long long stride, float* out,
                                       long long n, int accumulate, int perm_taps, int perm_ci,
                                       void* stream) {
    T2_REQUIRE(partials && out && nsplit >= 1 && n > 0, "splitk_reduce: bad args");
    if (perm_taps > 0) T2_REQUIRE(perm_ci > 0 && n % ((long long)perm_taps * perm_ci) == 0, "splitk_reduce: bad permutation dims");
    int blocks = t2_cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partials,
                       nsplit, stride, out, n, accumulate, perm_taps, perm_ci);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// out[r][c] (row stride ldo) (+)= sum_s partials[s][r cols + c]: the split-K partials of a product that fills a column
// block of a wider matrix (one input block of [dW_ih | dW_hh]).
__global__ void splitk_reduce2d_kernel(const float* __restrict__ part, int nsplit, long long stride, float* __restrict__ out,
                                       long long n, int cols, long long ldo, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            const float v0 = part[(long long)k * stride + i], v1 = part[(long long)(k + 1) * stride + i];
            const float v2 = part[(long long)(k + 2) * stride + i], v3 = part[(long long)(k + 3) * stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nsplit; ++k) s += part[(long long)k * stride + i];
        const long long r = i / cols;
        const long long o = r * ldo + (i - r * cols);
        if (accumulate) s += out[o];
        out[o] = s;
    }
}

extern "C" int t2amd_splitk_reduce2d_f32(const float* partials, int nsplit, long long stride, float* out, int rows, int cols,
                                         long long ldo, int accumulate, void* stream) {
    T2_REQUIRE(partials && out && nsplit >= 1 && rows > 0 && cols > 0 && ldo >= cols, "splitk_reduce2d: bad args");
    const long long n = (long long)rows * cols;
    int blocks = t2_cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(splitk_reduce2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partials, nsplit, stride, out, n, cols, ldo,
              accumulate);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// =========================================================================================
// Split-bf16 GEMM ("bf16x3"): every f32 operand element x is split on its way into LDS into
// hi = bf16(x) and lo = bf16(x - hi) (round-to-nearest-even, v_cvt_pk_bf16_f32), and the product
// is formed as Ah.Bh + Ah.Bl + Al.Bh on v_mfma_f32_32x32x16_bf16 with f32 accumulation: three
// matrix instructions at 16x the f32-MFMA rate = 5.3x the exact-f32 kernel's MFMA ceiling, at a
// relative error of ~2^-17 per product (f32: 2^-24, plain bf16: 2^-9).  The engine uses it for
// GRADIENT GEMMs only (tolerance 1e-3 of the gradient's max); every forward GEMM stays on the
// exact-f32 kernel above so that outputs remain bit-faithful to an fmaf chain.
//
// Tile 128 x 128 x 32, 4 waves (2 x 2), each wave 2 x 2 MFMA tiles of 32 x 32.  Both operands sit
// K-contiguous in LDS ([row][32 k] bf16, row stride 40 = 80 B: conflict-free ds_read_b128 fragment
// reads); M/N-contiguous operands are transposed in registers (4 k x 4 m blocks) before the store.
// =========================================================================================
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define HBM_ 128
#define HBN_ 128
#define HBK_ 32
#define HLD_ 40     // LDS row stride in bf16 elements

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// four consecutive-k f32 values -> packed hi (2 dwords) and lo (2 dwords)
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
    hi.x = cvt_pk_bf16(x0, x1);
    hi.y = cvt_pk_bf16(x2, x3);
    const float h0 = __uint_as_float(hi.x << 16), h1 = __uint_as_float(hi.x & 0xffff0000u);
    const float h2 = __uint_as_float(hi.y << 16), h3 = __uint_as_float(hi.y & 0xffff0000u);
    lo.x = cvt_pk_bf16(x0 - h0, x1 - h1);
    lo.y = cvt_pk_bf16(x2 - h2, x3 - h3);
}

// X3 = false is the plain bf16 product (precision 2): only the hi halves are formed, stored and multiplied.
// TB = 128: 256 threads, waves 2 x 2, 64 x 64 per wave.  TB = 256: 512 threads, waves 2 (m) x 4 (n), 128 x 64 per
// wave -- every operand byte fetched from L2 feeds twice the MFMA work (the 128-tile kernel saturates at ~300 TF on
// the 55k-deep wgrads whatever its occupancy: it is bound by operand traffic, not by latency).
template <bool AK, bool BKC, bool X3, int TB>
__global__ __launch_bounds__(2 * TB) void gemm_bf16x3_kernel(GemmParams p) {
    constexpr int NTHR = 2 * TB;
    constexpr int WN = TB / 64;            // waves along n (2 or 4); 2 along m
    constexpr int TM = TB / 64;            // 32-row MFMA tiles per wave along m (2 or 4); 2 along n
    // [buf][hi/lo][row][HLD_]; the plain-bf16 variant has no lo image: 40 KB instead of 80 KB -> twice the
    // workgroups per CU
    // K-contiguous operands use padded rows (HLD_ shorts); the pair-interleaved image of an M-contiguous operand is
    // exactly [16][128] dwords and needs no padding: 32 KB for a wgrad (both M-contiguous), five workgroups per CU.
    constexpr int NH = X3 ? 2 : 1;
    constexpr int ALD = AK ? HLD_ : 32, BLD = BKC ? HLD_ : 32;
    constexpr int AIMG = TB * ALD, BIMG = TB * BLD;       // shorts per image
    __shared__ __attribute__((aligned(16))) unsigned short As[2][NH * AIMG];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][NH * BIMG];

    const t2amd_gemm_desc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const TileId tile = xcd_tile_id();
    const int z = tile.bz;
    const int bidx = z / d.splitk;
    const int split = z - bidx * d.splitk;
    const float* __restrict__ A = d.A + (long long)bidx * d.strideA;
    const float* __restrict__ B = d.B + (long long)bidx * d.strideB;
    float* __restrict__ C = d.C + (long long)bidx * d.strideC + (long long)split * d.strideSplitC;

    const int row0 = tile.by * TB;
    const int col0 = tile.bx * TB;
    const int M = d.M, N = d.N;
    const int kbeg = split * p.ktiles_per_split * HBK_;
    int kend = kbeg + p.ktiles_per_split * HBK_;
    if (kend > d.K) kend = d.K;
    const int nk = (kend > kbeg) ? (kend - kbeg + HBK_ - 1) / HBK_ : 0;

    // K-contiguous operand: 128 rows x 32 k = 1024 float4, f = tid + 256*i -> (row = f>>3, kq = f&7)
    // M-contiguous operand: 32 k x 128 m: thread -> (m4 = tid&31, kq = tid>>5), 4 float4 = k 4kq..4kq+3
    int a_t[4] = {0, 0, 0, 0};
    if (AK && d.convA_T > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_t[i] = (row0 + ((tid + NTHR * i) >> 3)) % d.convA_T;
    }
    int b_tap = 0, b_ci = 0;
    if (!BKC && d.convB_T > 0) {
        const int gn = col0 + (tid & (TB / 4 - 1)) * 4;
        b_tap = gn / d.convB_C;
        b_ci = gn - b_tap * d.convB_C;
    }

    float4 ra[4], rb[4];

    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AK) {
                const int f = tid + NTHR * i;
                const int r = f >> 3, kq = f & 7;
                const int gm = row0 + r;
                const int gk = k0 + kq * 4;
                if (gm < M && gk < kend) {
                    long long srow = gm;
                    int col = gk;
                    bool ok = true;
                    if (d.convA_T > 0) {
                        const int tap = k0 / d.convA_C;
                        col = (k0 - tap * d.convA_C) + kq * 4;
                        const int sh = (tap - d.convA_pad) * d.convA_sign;
                        const int tt = a_t[i] + sh;
                        ok = (tt >= 0) && (tt < d.convA_T);
                        srow = (long long)gm + sh;
                    }
                    if (ok) {
                        const float* src = A + srow * d.lda + col;
                        if (p.avec) {
                            v = *reinterpret_cast<const float4*>(src);
                        } else {
                            v.x = src[0];
                            if (gk + 1 < kend) v.y = src[1];
                            if (gk + 2 < kend) v.z = src[2];
                            if (gk + 3 < kend) v.w = src[3];
                        }
                    }
                }
            } else {
                const int m4 = tid & (TB / 4 - 1), kq = tid / (TB / 4);
                const int gk = k0 + kq * 4 + i;
                const int gm = row0 + m4 * 4;
                if (gk < kend && gm < M) {
                    const float* src = A + (long long)gk * d.lda + gm;
                    if (p.avec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (gm + 1 < M) v.y = src[1];
                        if (gm + 2 < M) v.z = src[2];
                        if (gm + 3 < M) v.w = src[3];
                    }
                }
            }
            ra[i] = v;
        }
    };

    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BKC) {
                const int f = tid + NTHR * i;
                const int r = f >> 3, kq = f & 7;
                const int gn = col0 + r;
                const int gk = k0 + kq * 4;
                if (gn < N && gk < kend) {
                    const float* src = B + (long long)gn * d.ldb + gk;
                    if (p.bvec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (gk + 1 < kend) v.y = src[1];
                        if (gk + 2 < kend) v.z = src[2];
                        if (gk + 3 < kend) v.w = src[3];
                    }
                }
            } else {
                const int n4 = tid & (TB / 4 - 1), kq = tid / (TB / 4);
                const int gk = k0 + kq * 4 + i;
                const int gn = col0 + n4 * 4;
                if (gk < kend && gn < N) {
                    long long srow = gk;
                    int col = gn;
                    bool ok = true;
                    if (d.convB_T > 0) {
                        const int sh = b_tap - d.convB_pad;
                        const int tt = (gk % d.convB_T) + sh;
                        ok = (tt >= 0) && (tt < d.convB_T);
                        srow = (long long)gk + sh;
                        col = b_ci;
                    }
                    if (ok) {
                        const float* src = B + srow * d.ldb + col;
                        if (p.bvec) {
                            v = *reinterpret_cast<const float4*>(src);
                        } else {
                            v.x = src[0];
                            if (gn + 1 < N) v.y = src[1];
                            if (gn + 2 < N) v.z = src[2];
                            if (gn + 3 < N) v.w = src[3];
                        }
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto store_one = [&](unsigned short* S, int img, bool kc, const float4 (&r)[4]) {
        // S = As[buf] or Bs[buf] : hi image, then (X3 only) the lo image `img` shorts further
        if (kc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + NTHR * i;
                const int row = f >> 3, kq = f & 7;
                uint2 hi, lo;
                split4(r[i].x, r[i].y, r[i].z, r[i].w, hi, lo);
                *reinterpret_cast<uint2*>(&S[row * HLD_ + kq * 4]) = hi;
                if (X3) *reinterpret_cast<uint2*>(&S[img + row * HLD_ + kq * 4]) = lo;
            }
        } else {
            // M-contiguous operand: "pair-interleaved" image P[k/2][m] (one dword = the bf16 pair (k, k+1) of
            // row m), aliased onto the same storage.  This thread holds k = 4kq..4kq+3 for m = 4m4..4m4+3:
            // two 16-byte stores per half, consecutive lanes -> consecutive 16 B: conflict-free.
            const int m4 = tid & (TB / 4 - 1), kq = tid / (TB / 4);
            unsigned* Ph = reinterpret_cast<unsigned*>(S);
            unsigned* Pl = reinterpret_cast<unsigned*>(S + (X3 ? img : 0));
            uint4 h0, h1, l0, l1;
            {
                const float a0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};   // k = 4kq
                const float a1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};   // k = 4kq + 1
                const float a2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
                const float a3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
                unsigned hh0[4], hh1[4], ll0[4], ll1[4];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    hh0[mm] = cvt_pk_bf16(a0[mm], a1[mm]);
                    hh1[mm] = cvt_pk_bf16(a2[mm], a3[mm]);
                    ll0[mm] = cvt_pk_bf16(a0[mm] - __uint_as_float(hh0[mm] << 16),
                                          a1[mm] - __uint_as_float(hh0[mm] & 0xffff0000u));
                    ll1[mm] = cvt_pk_bf16(a2[mm] - __uint_as_float(hh1[mm] << 16),
                                          a3[mm] - __uint_as_float(hh1[mm] & 0xffff0000u));
                }
                h0 = make_uint4(hh0[0], hh0[1], hh0[2], hh0[3]);
                h1 = make_uint4(hh1[0], hh1[1], hh1[2], hh1[3]);
                l0 = make_uint4(ll0[0], ll0[1], ll0[2], ll0[3]);
                l1 = make_uint4(ll1[0], ll1[1], ll1[2], ll1[3]);
            }
            *reinterpret_cast<uint4*>(&Ph[(2 * kq + 0) * TB + m4 * 4]) = h0;
            *reinterpret_cast<uint4*>(&Ph[(2 * kq + 1) * TB + m4 * 4]) = h1;
            if (X3) {
                *reinterpret_cast<uint4*>(&Pl[(2 * kq + 0) * TB + m4 * 4]) = l0;
                *reinterpret_cast<uint4*>(&Pl[(2 * kq + 1) * TB + m4 * 4]) = l1;
            }
        }
    };

    // fragment (8 consecutive k of one row) from either LDS image
    auto frag = [&](const unsigned short* S, bool kc, int row, int ks, int lhi_) -> bf16x8 {
        if (kc) {
            return *reinterpret_cast<const bf16x8*>(&S[row * HLD_ + ks * 16 + lhi_ * 8]);
        } else {
            const unsigned* P = reinterpret_cast<const unsigned*>(S);
            const int p0 = ks * 8 + lhi_ * 4;
            union { unsigned u[4]; bf16x8 v; } t;
            t.u[0] = P[(p0 + 0) * TB + row];
            t.u[1] = P[(p0 + 1) * TB + row];
            t.u[2] = P[(p0 + 2) * TB + row];
            t.u[3] = P[(p0 + 3) * TB + row];
            return t.v;
        }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_a(kbeg);
        load_b(kbeg);
        store_one(As[0], AIMG, AK, ra);
        store_one(Bs[0], BIMG, BKC, rb);
    }
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_a(kbeg + (kt + 1) * HBK_);
            load_b(kbeg + (kt + 1) * HBK_);
        }
#pragma unroll
        for (int ks = 0; ks < HBK_ / 16; ++ks) {
            bf16x8 ah[TM], al[TM], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                ah[t] = frag(As[cur], AK, wm * (32 * TM) + t * 32 + l31, ks, lhi);
                if (X3) al[t] = frag(As[cur] + AIMG, AK, wm * (32 * TM) + t * 32 + l31, ks, lhi);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = frag(Bs[cur], BKC, wn * 64 + t * 32 + l31, ks, lhi);
                if (X3) bl[t] = frag(Bs[cur] + BIMG, BKC, wn * 64 + t * 32 + l31, ks, lhi);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (X3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            store_one(As[cur ^ 1], AIMG, AK, ra);
            store_one(Bs[cur ^ 1], BIMG, BKC, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA is dtype independent) -------------------
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int gn = col0 + wn * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int gm = row0 + wm * (32 * TM) + tm * 32 + row;
                if (gm < M && gn < N) {
                    float val = acc[tm][tn][r];
                    float* cp = C + (long long)gm * d.ldc + gn;
                    if (d.bias) val += d.bias[gn];
                    if (d.accumulate) val += *cp;
                    if (d.act == 1) val = fmaxf(val, 0.f);
                    if (d.keep) val = d.keep[(long long)gm * d.ldkeep + gn] ? val * d.keep_scale : 0.f;
                    *cp = val;
                }
            }
        }
    }
}
