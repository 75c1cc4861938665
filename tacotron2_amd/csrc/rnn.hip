// Recurrent-cell kernels for gfx950: batch x K x 4H "skinny" GEMM on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32) with the LSTM cell fused into the epilogue.
//
// One workgroup (4 waves) owns 64 batch rows x 16 gate columns = 4 hidden units x {i,f,g,o},
// over the whole K (no split: the cell needs the complete pre-activation), so 4H/16
// workgroups stream disjoint 16-row slices of the packed weight [4H][K] exactly once per
// step from HBM/MALL; the activations (64 x K) are re-read by every workgroup from L2.
//
// Operand staging is LDS-DMA (global_load_lds_dwordx4): no VGPRs, no ds_write pass, and the DMA of
// tiles kt+1, kt+2 stays in flight across the one barrier per k-tile (raw s_barrier + counted
// vmcnt, never __syncthreads()).  A DMA instruction writes 64 lanes x 16 B contiguously, so the
// LDS image is plain row-major [row][64 floats]; bank conflicts on the b128 fragment reads are
// removed by permuting the SOURCE: LDS chunk p of row r holds global chunk p ^ (r & 15), and the
// reader asks for chunk (4s+lg) ^ (row & 15).  Wave w DMAs exactly the 16 activation rows it
// multiplies itself, plus 4 of the 16 weight rows everybody reads.
//
// Replaces torch.nn.LSTMCell + F.dropout (reference model.py:352-356, 366-371) and the
// per-timestep work of nn.LSTM (model.py:181-188).
#include "common.h"
#include "cell_bwd.h"
#include <hip/hip_ext.h>

#include "skinny_wide.h"

#define SK_DEPTH 4    // register ring: tiles kt+1 .. kt+3 are in flight from HBM/L2 while tile kt is multiplied

// TAG only gives each role its own kernel symbol, so that rocprofv3 --kernel-trace --stats reports
// the decoder's attention-LSTM (1) / decoder-LSTM (2) launches separately from the rest (0).
// BF = true: the activation segments and W hold bf16 (widths, ld, Ktot in ELEMENTS); a 256-byte tile row is
// then 128 k, one 16-byte chunk is one v_mfma_f32_16x16x32_bf16 fragment, and everything byte-shaped (DMA,
// LDS image, swizzle, fragment reads) is unchanged.  Accumulation, cell state and all outputs stay f32.
template <bool LSTM, int TAG, bool BF>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyDual dp) {
    constexpr int ES = BF ? 2 : 4;          // bytes per operand element
    constexpr int BK = 256 / ES;            // k per tile (tile rows are 256 bytes)
    constexpr int EPC = 16 / ES;            // elements per 16-byte chunk
    // ONE shared array (a second __shared__ object makes hipcc drain the DMA queue before every ds_read)
    __shared__ __attribute__((aligned(16))) float smem[SK_NBUF * (SK_XT + SK_WT) + SK_ROWS * 17];
    float* const Xs = smem;                               // [NBUF][64][64]
    float* const Ws = smem + SK_NBUF * SK_XT;             // [NBUF][16][64]
    float* const Os = smem + SK_NBUF * (SK_XT + SK_WT);   // [64][17]

    const bool second = (int)blockIdx.x >= dp.nblk0;
    const SkinnyParams p = skinny_select(dp, second);
    const int lb = (int)blockIdx.x - (second ? dp.nblk0 : 0);
    const int bx = lb % p.gx;
    const int by = (lb / p.gx) % p.gy;
    const int bz = lb / (p.gx * p.gy);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int rowbase = by * SK_ROWS;
    const int B = p.B;

    // Virtual k-tiles: absent (all-zero) segments contribute nothing and are skipped outright.
    const int n0 = p.x[0].p ? p.x[0].width / BK : 0;
    const int n1 = (p.nseg > 1 && p.x[1].p) ? p.x[1].width / BK : 0;
    const int n2 = (p.nseg > 2 && p.x[2].p) ? p.x[2].width / BK : 0;
    const int wo1 = p.x[0].width, wo2 = p.x[0].width + (p.nseg > 1 ? p.x[1].width : 0);
    const int nvt = n0 + n1 + n2;
    int kt_beg = 0, kt_end = nvt;
    if (!LSTM) {
        kt_beg = bz * p.ktiles_per_split;
        kt_end = kt_beg + p.ktiles_per_split;
        if (kt_end > nvt) kt_end = nvt;
    }

    // LSTM epilogue operands: fetched up front so that their latency hides behind the main loop
    const int erow = tid >> 2, ejj = tid & 3;
    const int egr = rowbase + erow;
    const int ej = bx * 4 + ejj;
    // (raw values only: a comparison here would make the compiler wait for the load before the first DMA)
    float e_gin[4] = {0.f, 0.f, 0.f, 0.f}, e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_cp = 0.f;
    int e_keep_raw = 1;
    int e_len = 0x7fffffff;
    if (LSTM && egr < B) {
        const int H = p.H;
        const int* lens_ = p.lens;
        const float* gin_ = p.gin;
        const float* bias_ = p.bias;
        const float* cprev_ = p.c_prev;
        const uint8_t* keep_ = p.keep;
        const long long ld_gin_ = p.ld_gin, ld_cprev_ = p.ld_cprev, ld_keep_ = p.ld_keep;
        if (lens_) e_len = lens_[egr];
        if (gin_) {
            const float* g = gin_ + (long long)egr * ld_gin_;
#pragma unroll
            for (int q = 0; q < 4; ++q) e_gin[q] = g[q * H + ej];
        }
        if (bias_) {
#pragma unroll
            for (int q = 0; q < 4; ++q) e_bias[q] = bias_[q * H + ej];
        }
        if (cprev_) e_cp = cprev_[(long long)egr * ld_cprev_ + ej];
        if (keep_) e_keep_raw = keep_[(long long)egr * ld_keep_ + ej];
    }

    // ---- DMA source offsets (loop invariant) ------------------------------------------------------
    // X: instruction i of this wave fills LDS rows 16*wave + 4*i + lg, LDS chunk l15 <- global chunk
    // l15 ^ (row & 15).  Rows past B are clamped (their results are never stored).
    long long xo0[4], xo1[4], xo2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * wave + 4 * i + lg;
        int gr = rowbase + r;
        if (gr > B - 1) gr = B - 1;
        const int c4 = EPC * (l15 ^ (r & 15));
        xo0[i] = ((long long)gr * p.x[0].ld + c4) * ES;      // byte offsets
        xo1[i] = ((long long)gr * p.x[1].ld + c4) * ES;
        xo2[i] = ((long long)gr * p.x[2].ld + c4) * ES;
    }
    // W: this wave fills weight-tile rows 4*wave + lg.  Rows past N (plain kernel, ragged last block)
    // are clamped: their columns are never stored.
    long long wo;
    {
        const int c = 4 * wave + lg;
        long long wrow;
        if (LSTM) {
            wrow = (long long)(c >> 2) * p.H + bx * 4 + (c & 3);
        } else {
            wrow = (long long)bx * 16 + c;
            if (wrow > p.N - 1) wrow = p.N - 1;
        }
        wo = (wrow * p.Ktot + EPC * (l15 ^ (c & 15))) * ES;   // bytes
    }
    const char* const Wp = reinterpret_cast<const char*>(p.W) + wo;
    const char* const xp0 = reinterpret_cast<const char*>(p.x[0].p);
    const char* const xp1 = reinterpret_cast<const char*>(p.x[1].p);
    const char* const xp2 = reinterpret_cast<const char*>(p.x[2].p);
    const int kt_last = kt_end - 1;

    // Running DMA source pointers: the tile sequence only moves forward (and sticks at the last tile),
    // so each issue is five DMA instructions plus five pointer bumps; the segment switch is a rare,
    // wave-uniform branch.  iss_seg/iss_rem: segment of the next tile to issue / tiles left in it.
    const char* xq0; const char* xq1; const char* xq2; const char* xq3; const char* wq;   // byte pointers
    int iss_kt = kt_beg, iss_seg = 0, iss_rem = 0;
    // position the pointers on tile LOCAL of segment SEG (wave-uniform arguments; static array indices only)
#define SK_SEEK(SEG, LOCAL)                                                                            \
    {                                                                                                  \
        const int seg_ = (SEG), loc_ = (LOCAL);                                                        \
        if (seg_ == 0) {                                                                               \
            const char* sp_ = xp0 + loc_ * 256;                                                        \
            xq0 = sp_ + xo0[0]; xq1 = sp_ + xo0[1]; xq2 = sp_ + xo0[2]; xq3 = sp_ + xo0[3];            \
            wq = Wp + loc_ * 256; iss_rem = n0 - loc_;                                                 \
        } else if (seg_ == 1) {                                                                        \
            const char* sp_ = xp1 + loc_ * 256;                                                        \
            xq0 = sp_ + xo1[0]; xq1 = sp_ + xo1[1]; xq2 = sp_ + xo1[2]; xq3 = sp_ + xo1[3];            \
            wq = Wp + wo1 * ES + loc_ * 256; iss_rem = n1 - loc_;                                      \
        } else {                                                                                       \
            const char* sp_ = xp2 + loc_ * 256;                                                        \
            xq0 = sp_ + xo2[0]; xq1 = sp_ + xo2[1]; xq2 = sp_ + xo2[2]; xq3 = sp_ + xo2[3];            \
            wq = Wp + wo2 * ES + loc_ * 256; iss_rem = n2 - loc_;                                      \
        }                                                                                              \
        iss_seg = seg_;                                                                                \
    }
    xq0 = xq1 = xq2 = xq3 = wq = reinterpret_cast<const char*>(p.W);
    if (kt_end > kt_beg) {
        if (kt_beg < n0) SK_SEEK(0, kt_beg)
        else if (kt_beg < n0 + n1) SK_SEEK(1, kt_beg - n0)
        else SK_SEEK(2, kt_beg - n0 - n1)
    }

#define SK_ISSUE(BUF)                                                                                  \
    {                                                                                                  \
        float* xd_ = Xs + (BUF) * SK_XT + wave * (16 * SK_BK);                                         \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq0), (t2_lptr)(xd_), 16, 0, 0);                    \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq1), (t2_lptr)(xd_ + 256), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq2), (t2_lptr)(xd_ + 512), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq3), (t2_lptr)(xd_ + 768), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((t2_gptr)(wq), (t2_lptr)(Ws + (BUF) * SK_WT + wave * 256), 16, 0, 0); \
        if (iss_kt < kt_last) {                                                                        \
            ++iss_kt;                                                                                  \
            if (--iss_rem > 0) {                                                                       \
                xq0 += 256; xq1 += 256; xq2 += 256; xq3 += 256; wq += 256;                             \
            } else if (iss_seg == 0 && n1 > 0) {                                                       \
                SK_SEEK(1, 0)                                                                          \
            } else {                                                                                   \
                SK_SEEK(2, 0)                                                                          \
            }                                                                                          \
        }                                                                                              \
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // Fragment reads: row l15 (of this wave's 16 rows / of the 16 weight rows), chunk (4s+lg)^l15.
    // They are issued from inline asm: hipcc orders every ds_read it can see behind ALL pending LDS-DMA
    // (s_waitcnt vmcnt(0)), which would drain the ring each tile; the DMA/read ordering is ours
    // (counted vmcnt + barrier above each tile), the LDS wait is inside the statement.
    unsigned aa[4], ba[4];
    {
        const unsigned xbase = (unsigned)reinterpret_cast<size_t>((t2_lptr)(Xs + wave * (16 * SK_BK)));
        const unsigned wbase = (unsigned)reinterpret_cast<size_t>((t2_lptr)(Ws));
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const unsigned fo = 4u * (unsigned)(l15 * SK_BK + 4 * ((4 * s_ + lg) ^ l15));
            aa[s_] = xbase + fo;
            ba[s_] = wbase + fo;
        }
    }

    // Two fragment register sets: while the MFMAs consume set X (tile kt), the ds_reads of tile kt+1
    // fill set Y; the statement that waits for them closes the same step, so nothing the compiler
    // might do at the loop back-edge can touch a register whose LDS data is still in flight.
    f32x4 fa0, fa1, fa2, fa3, ga0, ga1, ga2, ga3;     // set A: activations / weights
    f32x4 fb0, fb1, fb2, fb3, gb0, gb1, gb2, gb3;     // set B

#define SK_READ(BUF, X0, X1, X2, X3, W0, W1, W2, W3)                                                   \
    asm volatile(                                                                                      \
        "ds_read_b128 %0, %8 offset:%16\n\t"                                                           \
        "ds_read_b128 %4, %12 offset:%17\n\t"                                                          \
        "ds_read_b128 %1, %9 offset:%16\n\t"                                                           \
        "ds_read_b128 %5, %13 offset:%17\n\t"                                                          \
        "ds_read_b128 %2, %10 offset:%16\n\t"                                                          \
        "ds_read_b128 %6, %14 offset:%17\n\t"                                                          \
        "ds_read_b128 %3, %11 offset:%16\n\t"                                                          \
        "ds_read_b128 %7, %15 offset:%17"                                                              \
        : "=&v"(X0), "=&v"(X1), "=&v"(X2), "=&v"(X3), "=&v"(W0), "=&v"(W1), "=&v"(W2), "=&v"(W3)       \
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba[0]), "v"(ba[1]), "v"(ba[2]), "v"(ba[3]), \
          "i"((BUF) * SK_XT * 4), "i"((BUF) * SK_WT * 4)                                               \
        : "memory");
#define SK_WAITR(X0, X1, X2, X3, W0, W1, W2, W3)                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                \
                 : "+v"(X0), "+v"(X1), "+v"(X2), "+v"(X3), "+v"(W0), "+v"(W1), "+v"(W2), "+v"(W3)      \
                 :                                                                                     \
                 : "memory");
#define SK_FMA4(X, W, ACC)                                                                             \
    if constexpr (BF) {                                                                                \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, (X)),              \
                                                      __builtin_bit_cast(sk_bf16x8, (W)), ACC, 0, 0, 0); \
    } else {                                                                                           \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((X)[0], (W)[0], acc0, 0, 0, 0);                    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((X)[1], (W)[1], acc1, 0, 0, 0);                    \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((X)[2], (W)[2], acc0, 0, 0, 0);                    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((X)[3], (W)[3], acc1, 0, 0, 0);                    \
    }
#define SK_FMA(X0, X1, X2, X3, W0, W1, W2, W3) SK_FMA4(X0, W0, acc0) SK_FMA4(X1, W1, acc1) SK_FMA4(X2, W2, acc0) SK_FMA4(X3, W3, acc1)
#define SK_SETA fa0, fa1, fa2, fa3, ga0, ga1, ga2, ga3
#define SK_SETB fb0, fb1, fb2, fb3, gb0, gb1, gb2, gb3
#define SK_X(M, ...) M(__VA_ARGS__)
    // One k-tile (tile KT sits in LDS buffer BUF and, already, in register set CUR).  Every wave has
    // issued exactly 5 DMA instructions per tile, always (tile indices are clamped, never branched on),
    // so "tile KT+1 landed" is vmcnt(5): only tile KT+2's five may be pending.  The barrier then (a) makes
    // the other waves' weight rows of tile KT+1 visible and (b) proves everybody has the fragments of
    // tile KT in registers, so its buffer can take the DMA of tile KT+3.
#define SK_STEP(BUF, KT, CUR, NXT)                                           \
    {                                                                        \
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                     \
        __builtin_amdgcn_s_barrier();                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SK_X(SK_READ, ((BUF) + 1) % SK_NBUF, NXT)                            \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SK_ISSUE(BUF)                                                        \
        SK_X(SK_FMA, CUR)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SK_X(SK_WAITR, NXT)                                                  \
    }

    if (kt_end > kt_beg) {
        SK_ISSUE(0)
        SK_ISSUE(1)
        SK_ISSUE(2)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SK_X(SK_READ, 0, SK_SETA)
        SK_X(SK_WAITR, SK_SETA)
        int kt = kt_beg;
        for (; kt + 6 <= kt_end; kt += 6) {
            SK_STEP(0, kt, SK_SETA, SK_SETB)
            SK_STEP(1, kt + 1, SK_SETB, SK_SETA)
            SK_STEP(2, kt + 2, SK_SETA, SK_SETB)
            SK_STEP(0, kt + 3, SK_SETB, SK_SETA)
            SK_STEP(1, kt + 4, SK_SETA, SK_SETB)
            SK_STEP(2, kt + 5, SK_SETB, SK_SETA)
        }
        if (kt < kt_end) {
            SK_STEP(0, kt, SK_SETA, SK_SETB)
            if (kt + 1 < kt_end) {
                SK_STEP(1, kt + 1, SK_SETB, SK_SETA)
                if (kt + 2 < kt_end) {
                    SK_STEP(2, kt + 2, SK_SETA, SK_SETB)
                    if (kt + 3 < kt_end) {
                        SK_STEP(0, kt + 3, SK_SETB, SK_SETA)
                        if (kt + 4 < kt_end) SK_STEP(1, kt + 4, SK_SETA, SK_SETB)
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped duplicate tiles
    }
#undef SK_ISSUE
#undef SK_SEEK
#undef SK_READ
#undef SK_WAITR
#undef SK_FMA4
#undef SK_FMA
#undef SK_STEP
#undef SK_X

    // D layout (16x16): col = lane&15, row = (lane>>4)*4 + reg
    if (!LSTM) {
        float* __restrict__ Y = p.Y + (long long)bz * p.split_stride;
        const int gn = bx * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = rowbase + wave * 16 + lg * 4 + r;
            if (gr < B && gn < p.N) {
                float v = acc0[r] + acc1[r];
                if (p.bias) v += p.bias[gn];
                if (p.act == 1) v = fmaxf(v, 0.f);
                if (p.keep) v = p.keep[(long long)gr * p.ld_keep + gn] ? v * p.keep_scale : 0.f;
                Y[(long long)gr * p.ldy + gn] = v;
                if (p.h16_out) p.h16_out[(long long)gr * p.ld_h16 + gn] = t2_f32_to_bf16(v);
                if (p.stop_active && gn == p.stop_col) skinny_stop_test(p, gr, v);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(wave * 16 + lg * 4 + r) * 17 + l15] = acc0[r] + acc1[r];
    __syncthreads();

    // LSTM cell: thread -> (row = tid>>2, unit jj = tid&3); gate g lives in column g*4+jj
    if (egr >= B) return;
    const int H = p.H;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, hn = 0.f;
    asm volatile("" : "+v"(e_keep_raw), "+v"(e_len));   // keeps the comparisons (and the wait for the loads) down here
    const bool e_valid = p.t < e_len, e_keep = e_keep_raw != 0;
    if (e_valid) {
        const float pi = Os[erow * 17 + ejj] + e_gin[0] + e_bias[0];
        const float pf = Os[erow * 17 + 4 + ejj] + e_gin[1] + e_bias[1];
        const float pg = Os[erow * 17 + 8 + ejj] + e_gin[2] + e_bias[2];
        const float po = Os[erow * 17 + 12 + ejj] + e_gin[3] + e_bias[3];
        gi = t2_sigmoid(pi);
        gf = t2_sigmoid(pf);
        gg = tanhf(pg);
        go = t2_sigmoid(po);
        cn = gf * e_cp + gi * gg;
        hn = go * tanhf(cn);
        if (p.keep) hn = e_keep ? hn * p.keep_scale : 0.f;
    }
    if (p.gates_out) {
        float* go_ = p.gates_out + (long long)egr * p.ld_gates;
        go_[ej] = gi;
        go_[H + ej] = gf;
        go_[2 * H + ej] = gg;
        go_[3 * H + ej] = go;
    }
    p.c_out[(long long)egr * p.ld_c + ej] = cn;
    p.h_out[(long long)egr * p.ld_h + ej] = hn;
    if (p.h16_out) p.h16_out[(long long)egr * p.ld_h16 + ej] = t2_f32_to_bf16(hn);
}

// the wide tile as a kernel of its own: one launch per step (body: skinny_wide.h)
// MODE: SW_BF16 / SW_F32 / SW_X3 (skinny_wide.h)
template <bool LSTM, int TAG, int MODE = SW_BF16>
__global__ __launch_bounds__(512) void skinny_wide_kernel(SkinnyDual dp) {
    __shared__ __attribute__((aligned(16))) char smem[SW_NBUF * (SW_XB + SW_WB)];
    const bool second = (int)blockIdx.x >= dp.nblk0;
    const SkinnyParams p = skinny_select(dp, second);
    const int lb = (int)blockIdx.x - (second ? dp.nblk0 : 0);
    NoGate ng;
    skinny_wide_body<LSTM, false, NoGate, false, MODE>(p, lb, smem, dp.ts, ng);
}

// ---------------------------------------------------------------------------------------
// 64 x 64 bf16 variant of the wide kernel, LSTM form only: 64 batch rows x 16 hidden units (x 4 gates) per workgroup.
//
// At decode batch sizes of 256 and more the 64 x 32 kernel above is two or more rounds of workgroups (B/64 row tiles x
// H/8 column tiles), and every round pays the fixed cost of a workgroup again (first DMA, first tile, partial-sum
// exchange, cell: about 5 us around a 3.6 us k loop); per output it also moves a 16 KB activation tile for every 8 KB of
// weights through the L2 -> LDS path.  Here a tile step feeds 64 gate columns: 16 KB of activations + 16 KB of weights
// (-33 % bytes per output), four v_mfma_f32_32x32x16_bf16 per wave (same (row half) x (k quarter) wave layout, the two
// activation fragments are multiplied by both column halves), and B = 256, H = 1024 is one round of 256 workgroups.
// 128 KB of LDS (4-tile ring), every wave issues exactly 4 DMA instructions per tile.  Same swizzle, counted vmcnt
// and inline-asm fragment reads as above.
// ---------------------------------------------------------------------------------------
#define S6_WB (64 * 256)    // bytes per weight tile
template <int TAG>
__global__ __launch_bounds__(512) void skinny_wide64_kernel(SkinnyDual dp) {
    constexpr int BK = 128;
    __shared__ __attribute__((aligned(16))) char smem6[SW_NBUF * (SW_XB + S6_WB)];
    char* const Xs = smem6;                      // [NBUF][64][256 B]
    char* const Ws = smem6 + SW_NBUF * SW_XB;    // [NBUF][64][256 B]
    float* const Ps = reinterpret_cast<float*>(smem6);   // epilogue: [4][64][65] partial sums (aliases the ring)

    const bool second = (int)blockIdx.x >= dp.nblk0;
    const SkinnyParams p = skinny_select(dp, second);
    const int lb = (int)blockIdx.x - (second ? dp.nblk0 : 0);
    const int bx = lb % p.gx;
    const int by = lb / p.gx;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wi = wave & 1, wk = wave >> 1;
    const int rowbase = by * SK_ROWS;
    const int B = p.B;

    const int n0 = p.x[0].p ? p.x[0].width / BK : 0;
    const int n1 = (p.nseg > 1 && p.x[1].p) ? p.x[1].width / BK : 0;
    const int n2 = (p.nseg > 2 && p.x[2].p) ? p.x[2].width / BK : 0;
    const int wo1 = p.x[0].width, wo2 = p.x[0].width + (p.nseg > 1 ? p.x[1].width : 0);
    const int nvt = n0 + n1 + n2;
    const int kt_beg = 0, kt_end = nvt;

    // cell operands, fetched up front: thread -> (row = tid>>3, units ejj and ejj + 8 of the workgroup's 16)
    const int erow = tid >> 3, ejj = tid & 7;
    const int egr = rowbase + erow;
    float e_gin[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, e_bias[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float e_cp[2] = {0.f, 0.f};
    int e_keep_raw[2] = {1, 1};
    int e_len = 0x7fffffff;
    if (egr < B) {
        const int H = p.H;
        const int* lens_ = p.lens;
        const float* gin_ = p.gin;
        const float* bias_ = p.bias;
        const float* cprev_ = p.c_prev;
        const uint8_t* keep_ = p.keep;
        const long long ld_gin_ = p.ld_gin, ld_cprev_ = p.ld_cprev, ld_keep_ = p.ld_keep;
        if (lens_) e_len = lens_[egr];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ej = bx * 16 + ejj + 8 * u;
            if (gin_) {
                const float* g = gin_ + (long long)egr * ld_gin_;
#pragma unroll
                for (int q = 0; q < 4; ++q) e_gin[u][q] = g[q * H + ej];
            }
            if (bias_) {
#pragma unroll
                for (int q = 0; q < 4; ++q) e_bias[u][q] = bias_[q * H + ej];
            }
            if (cprev_) e_cp[u] = cprev_[(long long)egr * ld_cprev_ + ej];
            if (keep_) e_keep_raw[u] = keep_[(long long)egr * ld_keep_ + ej];
        }
    }

    // DMA sources.  X: instruction q of this wave fills tile rows 8*wave + 4q + lg, LDS chunk l15 <- global chunk
    // l15 ^ (row & 15).  W: instruction q fills weight-tile rows (= gate columns) 32q + 4*wave + lg; tile column c is
    // gate c >> 4 of unit bx*16 + (c & 15).
    long long xo0[2], xo1[2], xo2[2], wo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * wave + 4 * q + lg;
        int gr = rowbase + r;
        if (gr > B - 1) gr = B - 1;
        const int c8 = 8 * (l15 ^ (r & 15));
        xo0[q] = ((long long)gr * p.x[0].ld + c8) * 2;
        xo1[q] = ((long long)gr * p.x[1].ld + c8) * 2;
        xo2[q] = ((long long)gr * p.x[2].ld + c8) * 2;
        const int c = 32 * q + 4 * wave + lg;
        const long long wrow = (long long)(c >> 4) * p.H + bx * 16 + (c & 15);
        wo[q] = (wrow * p.Ktot + 8 * (l15 ^ (c & 15))) * 2;
    }
    const char* const Wp0 = reinterpret_cast<const char*>(p.W) + wo[0];
    const long long wdelta = wo[1] - wo[0];
    const char* const xp0 = reinterpret_cast<const char*>(p.x[0].p);
    const char* const xp1 = reinterpret_cast<const char*>(p.x[1].p);
    const char* const xp2 = reinterpret_cast<const char*>(p.x[2].p);
    const int kt_last = kt_end - 1;

    const char* xq0; const char* xq1; const char* wq;
    int iss_kt = kt_beg, iss_seg = 0, iss_rem = 0;
#define S6_SEEK(SEG, LOCAL)                                                                            \
    {                                                                                                  \
        const int seg_ = (SEG), loc_ = (LOCAL);                                                        \
        if (seg_ == 0) {                                                                               \
            const char* sp_ = xp0 + loc_ * 256;                                                        \
            xq0 = sp_ + xo0[0]; xq1 = sp_ + xo0[1];                                                    \
            wq = Wp0 + loc_ * 256; iss_rem = n0 - loc_;                                                \
        } else if (seg_ == 1) {                                                                        \
            const char* sp_ = xp1 + loc_ * 256;                                                        \
            xq0 = sp_ + xo1[0]; xq1 = sp_ + xo1[1];                                                    \
            wq = Wp0 + wo1 * 2 + loc_ * 256; iss_rem = n1 - loc_;                                      \
        } else {                                                                                       \
            const char* sp_ = xp2 + loc_ * 256;                                                        \
            xq0 = sp_ + xo2[0]; xq1 = sp_ + xo2[1];                                                    \
            wq = Wp0 + wo2 * 2 + loc_ * 256; iss_rem = n2 - loc_;                                      \
        }                                                                                              \
        iss_seg = seg_;                                                                                \
    }
    xq0 = xq1 = wq = reinterpret_cast<const char*>(p.W);
    if (kt_end > kt_beg) {
        if (n0 > 0) S6_SEEK(0, 0)
        else if (n1 > 0) S6_SEEK(1, 0)
        else S6_SEEK(2, 0)
    }

#define S6_ISSUE(BUF)                                                                                  \
    {                                                                                                  \
        char* xd_ = Xs + (BUF) * SW_XB + wave * (8 * 256);                                             \
        char* wd_ = Ws + (BUF) * S6_WB + wave * 1024;                                                  \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq0), (t2_lptr)(xd_), 16, 0, 0);                    \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq1), (t2_lptr)(xd_ + 1024), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((t2_gptr)(wq), (t2_lptr)(wd_), 16, 0, 0);                     \
        __builtin_amdgcn_global_load_lds((t2_gptr)(wq + wdelta), (t2_lptr)(wd_ + 32 * 256), 16, 0, 0); \
        if (iss_kt < kt_last) {                                                                        \
            ++iss_kt;                                                                                  \
            if (--iss_rem > 0) {                                                                       \
                xq0 += 256; xq1 += 256; wq += 256;                                                     \
            } else if (iss_seg == 0 && n1 > 0) {                                                       \
                S6_SEEK(1, 0)                                                                          \
            } else {                                                                                   \
                S6_SEEK(2, 0)                                                                          \
            }                                                                                          \
        }                                                                                              \
    }

    f32x16 acc0, acc1, acc2, acc3;      // (k chunk pair m = 0, 1) x (column half 0, 1): acc0/acc1 half 0, acc2/acc3 half 1
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }

    // fragment addresses: MFMA m of this wave uses k chunk 4*wk + 2m + lhi of activation row 32*wi + l31 and of weight
    // rows l31 (column half 0) and 32 + l31 (half 1: same chunk swizzle, (32 + l31) & 15 == l31 & 15)
    unsigned ax[2], aw[2];
    {
        const unsigned xbase = (unsigned)reinterpret_cast<size_t>((t2_lptr)(Xs));
        const unsigned wbase = (unsigned)reinterpret_cast<size_t>((t2_lptr)(Ws));
        const int rx = 32 * wi + l31;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = 4 * wk + 2 * m + lhi;
            ax[m] = xbase + (unsigned)(rx * 256 + ((c ^ (rx & 15)) << 4));
            aw[m] = wbase + (unsigned)(l31 * 256 + ((c ^ (l31 & 15)) << 4));
        }
    }
    f32x4 xa0, xa1, wa0, wa1, wa2, wa3, xb0, xb1, wb0, wb1, wb2, wb3;

#define S6_READ(BUF, X0, X1, W0, W1, W2, W3)                                                           \
    asm volatile(                                                                                      \
        "ds_read_b128 %0, %6 offset:%10\n\t"                                                           \
        "ds_read_b128 %2, %8 offset:%11\n\t"                                                           \
        "ds_read_b128 %4, %8 offset:%12\n\t"                                                           \
        "ds_read_b128 %1, %7 offset:%10\n\t"                                                           \
        "ds_read_b128 %3, %9 offset:%11\n\t"                                                           \
        "ds_read_b128 %5, %9 offset:%12"                                                               \
        : "=&v"(X0), "=&v"(X1), "=&v"(W0), "=&v"(W1), "=&v"(W2), "=&v"(W3)                             \
        : "v"(ax[0]), "v"(ax[1]), "v"(aw[0]), "v"(aw[1]), "i"((BUF) * SW_XB), "i"((BUF) * S6_WB),      \
          "i"((BUF) * S6_WB + 32 * 256)                                                                \
        : "memory");
#define S6_WAITR(X0, X1, W0, W1, W2, W3)                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X0), "+v"(X1), "+v"(W0), "+v"(W1), "+v"(W2), "+v"(W3) : : "memory");
#define S6_M(ACC, X, W)                                                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X)), __builtin_bit_cast(sk_bf16x8, (W)), ACC, 0, 0, 0);
#define S6_FMA(X0, X1, W0, W1, W2, W3)                                                                 \
    S6_M(acc0, X0, W0) S6_M(acc2, X0, W2) S6_M(acc1, X1, W1) S6_M(acc3, X1, W3)
#define S6_SETA xa0, xa1, wa0, wa1, wa2, wa3
#define S6_SETB xb0, xb1, wb0, wb1, wb2, wb3
#define S6_X(M, ...) M(__VA_ARGS__)
    // every wave issues exactly 4 DMA instructions per tile: "tile KT+1 landed" is vmcnt(8) (tiles KT+2, KT+3 may be
    // pending); the barrier publishes it and frees tile KT's buffer for the DMA of tile KT+4
#define S6_STEP(BUF, CUR, NXT)                                               \
    {                                                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                     \
        __builtin_amdgcn_s_barrier();                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        S6_X(S6_READ, ((BUF) + 1) % SW_NBUF, NXT)                            \
        __builtin_amdgcn_sched_barrier(0);                                   \
        S6_ISSUE(BUF)                                                        \
        S6_X(S6_FMA, CUR)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
        S6_X(S6_WAITR, NXT)                                                  \
    }

    if (kt_end > kt_beg) {
        S6_ISSUE(0)
        S6_ISSUE(1)
        S6_ISSUE(2)
        S6_ISSUE(3)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        S6_X(S6_READ, 0, S6_SETA)
        S6_X(S6_WAITR, S6_SETA)
        int kt = kt_beg;
        for (; kt + 4 <= kt_end; kt += 4) {
            S6_STEP(0, S6_SETA, S6_SETB)
            S6_STEP(1, S6_SETB, S6_SETA)
            S6_STEP(2, S6_SETA, S6_SETB)
            S6_STEP(3, S6_SETB, S6_SETA)
        }
        if (kt < kt_end) {
            S6_STEP(0, S6_SETA, S6_SETB)
            if (kt + 1 < kt_end) {
                S6_STEP(1, S6_SETB, S6_SETA)
                if (kt + 2 < kt_end) S6_STEP(2, S6_SETA, S6_SETB)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped duplicate tiles
    }
#undef S6_ISSUE
#undef S6_SEEK
#undef S6_READ
#undef S6_WAITR
#undef S6_FMA
#undef S6_M
#undef S6_STEP
#undef S6_X

    // k-quarter partial sums -> LDS (the ring is dead once every wave's DMA has drained)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Ps[(wk * 64 + row) * 65 + l31] = acc0[r] + acc1[r];
        Ps[(wk * 64 + row) * 65 + 32 + l31] = acc2[r] + acc3[r];
    }
    __syncthreads();

    if (egr >= B) return;
    const int H = p.H;
    asm volatile("" : "+v"(e_keep_raw[0]), "+v"(e_keep_raw[1]), "+v"(e_len));   // keeps the comparisons down here
    const bool e_valid = p.t < e_len;
    float* go_ = p.gates_out + (long long)egr * p.ld_gates;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ej = bx * 16 + ejj + 8 * u;
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, hn = 0.f;
        if (e_valid) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 16 + ejj + 8 * u;
                pre[q] = ((Ps[erow * 65 + c] + Ps[(64 + erow) * 65 + c]) + (Ps[(128 + erow) * 65 + c] + Ps[(192 + erow) * 65 + c]))
                         + e_gin[u][q] + e_bias[u][q];
            }
            gi = t2_sigmoid_fast(pre[0]);
            gf = t2_sigmoid_fast(pre[1]);
            gg = t2_tanh(pre[2]);
            go = t2_sigmoid_fast(pre[3]);
            cn = gf * e_cp[u] + gi * gg;
            hn = go * t2_tanh(cn);
            if (p.keep) hn = e_keep_raw[u] != 0 ? hn * p.keep_scale : 0.f;
        }
        if (p.gates_out) {
            go_[ej] = gi;
            go_[H + ej] = gf;
            go_[2 * H + ej] = gg;
            go_[3 * H + ej] = go;
        }
        p.c_out[(long long)egr * p.ld_c + ej] = cn;
        p.h_out[(long long)egr * p.ld_h + ej] = hn;
        if (p.h16_out) p.h16_out[(long long)egr * p.ld_h16 + ej] = t2_f32_to_bf16(hn);
    }
}

// T2AMD_SKINNY_NARROW=1 keeps the 64x16 kernel for bf16 operands too (A/B measurements)
static bool skinny_wide_enabled() {
    static const bool on = [] { const char* e = getenv("T2AMD_SKINNY_NARROW"); return !(e && e[0] == '1'); }();
    return on;
}

// k per 256-byte tile row by operand mode (t2amd_lstm_step.bf16): bf16 rows hold 128 k, f32 rows and split-bf16 images (3) 64
static inline int sk_bk(int mode) { return (mode == 1 || mode == 2) ? 128 : 64; }
static int check_segs(const t2amd_seg* x, int nseg, int Ktot, int bk) {
    if (nseg < 1 || nseg > 3) T2_FAIL("skinny: nseg must be 1..3");
    int sum = 0;
    for (int i = 0; i < nseg; ++i) {
        if (x[i].width <= 0 || x[i].width % bk != 0) T2_FAIL("skinny: segment widths must be positive multiples of 64 (f32, split-bf16 x3) / 128 (bf16)");
        if (x[i].p && (!t2_aligned16(x[i].p) || x[i].ld % 8 != 0)) T2_FAIL("skinny: segment must be 16-byte aligned with ld % 8 == 0");
        sum += x[i].width;
    }
    if (sum != Ktot) T2_FAIL("skinny: segment widths do not add up to Ktot");
    return T2AMD_OK;
}

static int fill_lstm(const t2amd_lstm_step* a, SkinnyParams& p) {
    T2_REQUIRE(a != nullptr, "lstm_step: null args");
    T2_REQUIRE(a->bf16 >= 0 && a->bf16 <= 3, "lstm_step: bf16 must be 0 (f32), 1 (bf16), 2 (bf16 weights only, small batch) or 3 (split-bf16 x3)");
    T2_PROPAGATE(check_segs(a->x, a->nseg, a->Ktot, sk_bk(a->bf16)));
    T2_REQUIRE(a->W && t2_aligned16(a->W), "lstm_step: W must be 16-byte aligned");
    T2_REQUIRE(a->H > 0 && a->H % 4 == 0 && a->B > 0, "lstm_step: H must be a multiple of 4");
    T2_REQUIRE(a->c_out && a->h_out, "lstm_step: null outputs");       // gates_out may be NULL: the gate activations are not kept
    p = SkinnyParams{};
    for (int i = 0; i < 3; ++i) p.x[i] = a->x[i];
    p.nseg = a->nseg;
    p.W = a->W; p.Ktot = a->Ktot; p.B = a->B; p.H = a->H; p.N = 4 * a->H;
    p.gin = a->gin; p.ld_gin = a->ld_gin; p.bias = a->bias;
    p.c_prev = a->c_prev; p.ld_cprev = a->ld_cprev;
    p.gates_out = a->gates_out; p.ld_gates = a->ld_gates;
    p.c_out = a->c_out; p.ld_c = a->ld_c; p.h_out = a->h_out; p.ld_h = a->ld_h;
    p.h16_out = (unsigned short*)a->h16_out; p.ld_h16 = a->ld_h16;
    p.keep = a->keep; p.ld_keep = a->ld_keep; p.keep_scale = a->keep_scale;
    p.lens = a->lens; p.t = a->t;
    p.gx = a->H / 4; p.gy = t2_cdiv(a->B, SK_ROWS); p.gz = 1;
    p.wcol[0] = p.wcol[1] = p.wcol[2] = -1;
    return T2AMD_OK;
}

// One launch for one or two independent LSTM steps (b may be NULL).  The kernel symbol / profiling
// role is a's tag.
// swap01: bit 0 / bit 1 = visit segment 1 of problem a / b BEFORE its segment 0 (explicit weight columns; the wide bf16 tile only --
// the other kernels keep the stored order).  The training loop asks for it on the attention LSTM (loops.hip), so that the launch chain
// sums in the same order as the persistent loop, which starts that LSTM on h_att while ctx is still being produced.
static int lstm_step_fwd2_impl(const t2amd_lstm_step* a, const t2amd_lstm_step* b, int swap01, void* stream) {
    SkinnyDual d;
    d.ts = t2amd_debug_ts_();
    T2_PROPAGATE(fill_lstm(a, d.p[0]));
    d.nblk0 = d.p[0].gx * d.p[0].gy;
    int total = d.nblk0;
    if (b) {
        T2_PROPAGATE(fill_lstm(b, d.p[1]));
        total += d.p[1].gx * d.p[1].gy;
    } else {
        d.p[1] = d.p[0];
    }
    hipStream_t s = (hipStream_t)stream;
    // bench.py's roofline leg: this role's launches carry an event pair stamped by the dispatch itself
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    const bool prof = !t2amd_validate_only_flag_() && t2amd_profile_pair_(a->tag, &pe0, &pe1);
#define LSTM_LAUNCH(K, G, BLK)                                                         \
    do {                                                                               \
        if (prof) hipExtLaunchKernelGGL(K, G, BLK, 0, s, pe0, pe1, 0, d);              \
        else T2_LAUNCH(K, G, BLK, 0, s, d);                                            \
    } while (0)
    T2_REQUIRE(!b || a->bf16 == b->bf16, "lstm_step: both problems of a launch must share the operand type");
    const bool x3 = a->bf16 == 3;
    T2_REQUIRE(!x3 || (skinny_wide_enabled() && a->H % 8 == 0 && (!b || b->H % 8 == 0)), "lstm_step: split-bf16 x3 operands run on the wide tile only (H % 8 == 0)");
    // several rounds of 64 x 32 workgroups (decode batches of 256 and more): one round of 64 x 64 ones instead
    static const int wide64 = [] { const char* e = getenv("T2AMD_LSTM_WIDE64"); return e ? atoi(e) : -1; }();   // A/B runs only
    if (a->bf16 && !x3 && skinny_wide_enabled() && !b && a->H % 16 == 0 &&
        (wide64 < 0 ? (long long)d.p[0].gy * (a->H / 8) >= 512 : wide64 != 0)) {
        d.p[0].gx = a->H / 16;
        d.nblk0 = total = d.p[0].gx * d.p[0].gy;
        d.p[1] = d.p[0];
        if (a->tag == 1) LSTM_LAUNCH((skinny_wide64_kernel<1>), dim3(total), dim3(512));
        else if (a->tag == 2) LSTM_LAUNCH((skinny_wide64_kernel<2>), dim3(total), dim3(512));
        else LSTM_LAUNCH((skinny_wide64_kernel<0>), dim3(total), dim3(512));
    } else if ((a->bf16 || (swap01 & 4)) && skinny_wide_enabled() && a->H % 8 == 0 && (!b || b->H % 8 == 0)) {
        // (swap01 bit 2, f32 operands: the wide tile on the exact-f32 MFMA -- what the fp32 parity mode's TRAINING loop asks for,
        // so that its launch chain and its persistent launch are the same arithmetic: csrc/skinny_wide.h, F32)
        d.p[0].gx = a->H / 8;
        d.nblk0 = d.p[0].gx * d.p[0].gy;
        total = d.nblk0;
        for (int k = 0; k < (b ? 2 : 1); ++k) {
            SkinnyParams& q = d.p[k];
            if (((swap01 >> k) & 1) && q.nseg >= 2) {
                const int w0 = q.x[0].width, w1 = q.x[1].width;
                const t2amd_seg s0 = q.x[0];
                q.x[0] = q.x[1]; q.x[1] = s0;
                q.wcol[0] = w0; q.wcol[1] = 0; q.wcol[2] = w0 + w1;
            }
        }
        if (b) { d.p[1].gx = b->H / 8; total += d.p[1].gx * d.p[1].gy; } else { d.p[1] = d.p[0]; }
        if (x3) {
            if (a->tag == 1) LSTM_LAUNCH((skinny_wide_kernel<true, 1, SW_X3>), dim3(total), dim3(512));
            else if (a->tag == 2) LSTM_LAUNCH((skinny_wide_kernel<true, 2, SW_X3>), dim3(total), dim3(512));
            else if (a->tag == 3) LSTM_LAUNCH((skinny_wide_kernel<true, 3, SW_X3>), dim3(total), dim3(512));
            else LSTM_LAUNCH((skinny_wide_kernel<true, 0, SW_X3>), dim3(total), dim3(512));
        } else
        if (!a->bf16) {
            if (a->tag == 1) LSTM_LAUNCH((skinny_wide_kernel<true, 1, SW_F32>), dim3(total), dim3(512));
            else if (a->tag == 2) LSTM_LAUNCH((skinny_wide_kernel<true, 2, SW_F32>), dim3(total), dim3(512));
            else if (a->tag == 3) LSTM_LAUNCH((skinny_wide_kernel<true, 3, SW_F32>), dim3(total), dim3(512));
            else LSTM_LAUNCH((skinny_wide_kernel<true, 0, SW_F32>), dim3(total), dim3(512));
        } else
        if (a->tag == 1) LSTM_LAUNCH((skinny_wide_kernel<true, 1>), dim3(total), dim3(512));
        else if (a->tag == 2) LSTM_LAUNCH((skinny_wide_kernel<true, 2>), dim3(total), dim3(512));
        else if (a->tag == 3) LSTM_LAUNCH((skinny_wide_kernel<true, 3>), dim3(total), dim3(512));
        else LSTM_LAUNCH((skinny_wide_kernel<true, 0>), dim3(total), dim3(512));
    } else if (a->bf16) {
        if (a->tag == 1) LSTM_LAUNCH((skinny_gemm_kernel<true, 1, true>), dim3(total), dim3(256));
        else if (a->tag == 2) LSTM_LAUNCH((skinny_gemm_kernel<true, 2, true>), dim3(total), dim3(256));
        else if (a->tag == 3) LSTM_LAUNCH((skinny_gemm_kernel<true, 3, true>), dim3(total), dim3(256));
        else LSTM_LAUNCH((skinny_gemm_kernel<true, 0, true>), dim3(total), dim3(256));
    } else {
        if (a->tag == 1) LSTM_LAUNCH((skinny_gemm_kernel<true, 1, false>), dim3(total), dim3(256));
        else if (a->tag == 2) LSTM_LAUNCH((skinny_gemm_kernel<true, 2, false>), dim3(total), dim3(256));
        else if (a->tag == 3) LSTM_LAUNCH((skinny_gemm_kernel<true, 3, false>), dim3(total), dim3(256));
        else LSTM_LAUNCH((skinny_gemm_kernel<true, 0, false>), dim3(total), dim3(256));
    }
#undef LSTM_LAUNCH
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_step_fwd2_f32(const t2amd_lstm_step* a, const t2amd_lstm_step* b, void* stream) {
    return lstm_step_fwd2_impl(a, b, 0, stream);
}
// internal (loops.hip): see lstm_step_fwd2_impl
extern "C" int t2amd_lstm_step_fwd2_order_(const t2amd_lstm_step* a, const t2amd_lstm_step* b, int swap01, void* stream) {
    return lstm_step_fwd2_impl(a, b, swap01, stream);
}

extern "C" int t2amd_lstm_step_fwd_f32(const t2amd_lstm_step* a, void* stream) {
    return t2amd_lstm_step_fwd2_f32(a, nullptr, stream);
}

static int fill_plain(const t2amd_skinny_gemm* a, SkinnyParams& p) {
    T2_REQUIRE(a != nullptr, "skinny_gemm: null args");
    T2_REQUIRE(a->bf16 == 0 || a->bf16 == 1 || a->bf16 == 3, "skinny_gemm: bf16 must be 0 (f32), 1 (bf16) or 3 (split-bf16 x3)");
    T2_PROPAGATE(check_segs(a->x, a->nseg, a->Ktot, sk_bk(a->bf16)));
    T2_REQUIRE(a->W && t2_aligned16(a->W) && a->Y, "skinny_gemm: bad pointers");
    T2_REQUIRE(a->N > 0 && a->B > 0 && a->nsplit >= 1, "skinny_gemm: bad dims");
    p = SkinnyParams{};
    for (int i = 0; i < 3; ++i) p.x[i] = a->x[i];
    p.nseg = a->nseg;
    p.W = a->W; p.Ktot = a->Ktot; p.B = a->B; p.N = a->N; p.H = 0;
    p.Y = a->Y; p.ldy = a->ldy; p.nsplit = a->nsplit; p.split_stride = a->split_stride;
    T2_REQUIRE((!a->bias && !a->act && !a->keep) || a->nsplit == 1, "skinny_gemm: the bias/act/keep epilogue needs nsplit == 1");
    T2_REQUIRE(a->act == 0 || a->act == 1, "skinny_gemm: act must be 0 or 1");
    p.bias = a->bias; p.act = a->act; p.keep = a->keep; p.ld_keep = a->ld_keep; p.keep_scale = a->keep_scale;
    T2_REQUIRE(!a->Y16 || a->nsplit == 1, "skinny_gemm: the bf16 copy of Y needs nsplit == 1");
    p.h16_out = (unsigned short*)a->Y16; p.ld_h16 = a->ldy16;       // plain epilogue: bf16 copy of Y
    T2_REQUIRE(!a->stop_active || (a->nsplit == 1 && a->stop_lengths && a->stop_done && a->stop_col >= 0 && a->stop_col < a->N),
               "skinny_gemm: the stop test needs nsplit == 1, its three arrays and a column of Y");
    p.stop_active = a->stop_active; p.stop_lengths = a->stop_lengths; p.stop_done = a->stop_done;
    p.stop_col = a->stop_col; p.stop_max_steps = a->stop_max_steps; p.stop_thr = a->stop_threshold; p.t = a->stop_t;
    const int ktiles = a->Ktot / sk_bk(a->bf16);
    p.ktiles_per_split = t2_cdiv(ktiles, a->nsplit);
    p.gx = t2_cdiv(a->N, 16); p.gy = t2_cdiv(a->B, SK_ROWS); p.gz = a->nsplit;
    p.wcol[0] = p.wcol[1] = p.wcol[2] = -1;
    return T2AMD_OK;
}

static int skinny_gemm2_impl(const t2amd_skinny_gemm* a, const t2amd_skinny_gemm* b, int order, void* stream);
extern "C" int t2amd_skinny_gemm2_f32(const t2amd_skinny_gemm* a, const t2amd_skinny_gemm* b, void* stream) {
    return skinny_gemm2_impl(a, b, 0, stream);
}
// internal (loops.hip): bit 2 of `order` = f32 operands on the WIDE 64 x 32 tile (csrc/skinny_wide.h, F32: exact-f32 MFMA) -- the BPTT
// data gradients of the fp32 parity mode's training loop (round 1 measured that tiling 2 % ahead of the 64 x 16 kernel on the fp32
// step, all of it in this product; the tile body exists since round 5 for that mode's persistent forward loop)
extern "C" int t2amd_skinny_gemm2_order_(const t2amd_skinny_gemm* a, const t2amd_skinny_gemm* b, int order, void* stream) {
    return skinny_gemm2_impl(a, b, order, stream);
}
static int skinny_gemm2_impl(const t2amd_skinny_gemm* a, const t2amd_skinny_gemm* b, int order, void* stream) {
    SkinnyDual d;
    d.ts = t2amd_debug_ts_();
    T2_PROPAGATE(fill_plain(a, d.p[0]));
    d.nblk0 = d.p[0].gx * d.p[0].gy * d.p[0].gz;
    int total = d.nblk0;
    if (b) {
        T2_PROPAGATE(fill_plain(b, d.p[1]));
        total += d.p[1].gx * d.p[1].gy * d.p[1].gz;
    } else {
        d.p[1] = d.p[0];
    }
    hipStream_t s = (hipStream_t)stream;
    T2_REQUIRE(!b || a->bf16 == b->bf16, "skinny_gemm: both problems of a launch must share the operand type");
    const bool x3 = a->bf16 == 3;
    T2_REQUIRE(!x3 || skinny_wide_enabled(), "skinny_gemm: split-bf16 x3 operands run on the wide tile only");
    if ((a->bf16 || (order & 4)) && skinny_wide_enabled()) {
        d.p[0].gx = t2_cdiv(a->N, 32);
        d.nblk0 = d.p[0].gx * d.p[0].gy * d.p[0].gz;
        total = d.nblk0;
        if (b) { d.p[1].gx = t2_cdiv(b->N, 32); total += d.p[1].gx * d.p[1].gy * d.p[1].gz; } else { d.p[1] = d.p[0]; }
        if (x3) {
            if (a->tag == 1) T2_LAUNCH((skinny_wide_kernel<false, 1, SW_X3>), dim3(total), dim3(512), 0, s, d);
            else if (a->tag == 2) T2_LAUNCH((skinny_wide_kernel<false, 2, SW_X3>), dim3(total), dim3(512), 0, s, d);
            else if (a->tag == 3) T2_LAUNCH_ROLE(6, (skinny_wide_kernel<false, 3, SW_X3>), dim3(total), dim3(512), 0, s, d);   // BPTT dgrad pair
            else T2_LAUNCH((skinny_wide_kernel<false, 0, SW_X3>), dim3(total), dim3(512), 0, s, d);
        } else
        if (!a->bf16) {
            if (a->tag == 1) T2_LAUNCH((skinny_wide_kernel<false, 1, SW_F32>), dim3(total), dim3(512), 0, s, d);
            else if (a->tag == 2) T2_LAUNCH((skinny_wide_kernel<false, 2, SW_F32>), dim3(total), dim3(512), 0, s, d);
            else if (a->tag == 3) T2_LAUNCH_ROLE(6, (skinny_wide_kernel<false, 3, SW_F32>), dim3(total), dim3(512), 0, s, d);   // BPTT dgrad pair
            else T2_LAUNCH((skinny_wide_kernel<false, 0, SW_F32>), dim3(total), dim3(512), 0, s, d);
        } else
        if (a->tag == 1) T2_LAUNCH((skinny_wide_kernel<false, 1>), dim3(total), dim3(512), 0, s, d);
        else if (a->tag == 2) T2_LAUNCH((skinny_wide_kernel<false, 2>), dim3(total), dim3(512), 0, s, d);
        else if (a->tag == 3) T2_LAUNCH_ROLE(6, (skinny_wide_kernel<false, 3>), dim3(total), dim3(512), 0, s, d);   // BPTT dgrad pair
        else T2_LAUNCH((skinny_wide_kernel<false, 0>), dim3(total), dim3(512), 0, s, d);
    } else if (a->bf16) {
        if (a->tag == 1) T2_LAUNCH((skinny_gemm_kernel<false, 1, true>), dim3(total), dim3(256), 0, s, d);
        else if (a->tag == 2) T2_LAUNCH((skinny_gemm_kernel<false, 2, true>), dim3(total), dim3(256), 0, s, d);
        else if (a->tag == 3) T2_LAUNCH_ROLE(6, (skinny_gemm_kernel<false, 3, true>), dim3(total), dim3(256), 0, s, d);
        else T2_LAUNCH((skinny_gemm_kernel<false, 0, true>), dim3(total), dim3(256), 0, s, d);
    } else {
        if (a->tag == 1) T2_LAUNCH((skinny_gemm_kernel<false, 1, false>), dim3(total), dim3(256), 0, s, d);
        else if (a->tag == 2) T2_LAUNCH((skinny_gemm_kernel<false, 2, false>), dim3(total), dim3(256), 0, s, d);
        else if (a->tag == 3) T2_LAUNCH_ROLE(6, (skinny_gemm_kernel<false, 3, false>), dim3(total), dim3(256), 0, s, d);
        else T2_LAUNCH((skinny_gemm_kernel<false, 0, false>), dim3(total), dim3(256), 0, s, d);
    }
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_skinny_gemm_f32(const t2amd_skinny_gemm* a, void* stream) {
    return t2amd_skinny_gemm2_f32(a, nullptr, stream);
}

extern "C" int t2amd_skinny_wide_enabled_(void) { return skinny_wide_enabled() ? 1 : 0; }

// ---------------------------------------------------------------------------------------
// LSTM cell backward (pointwise part): given dL/dh' (dropped-out hidden) and the carried
// dL/dc, produce the gate pre-activation gradients and the new dL/dc carry.
// ---------------------------------------------------------------------------------------
struct LstmBwdParams { t2amd_lstm_bwd a[2]; int nblk0; unsigned long long* ts; };

// one thread per (row, 4 consecutive units): 16-byte loads/stores; blocks [0, nblk0) serve a[0], the rest a[1]
#ifdef T2AMD_PHASE_STAMPS
#define PW_TS(slot)                                                                        \
    do {                                                                                   \
        if ((slot) == 0) ts_on = t2_ts_begin(p.ts, 96);                                    \
        else t2_ts_mark(ts_on, p.ts, 96 + (slot));                                         \
    } while (0)
#else
#define PW_TS(slot) do { (void)ts_on; } while (0)
#endif
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(LstmBwdParams p) {
    bool ts_on = false;
    PW_TS(0);
    const bool second = (int)blockIdx.x >= p.nblk0;
    const t2amd_lstm_bwd& a = second ? p.a[1] : p.a[0];
    const int lb = (int)blockIdx.x - (second ? p.nblk0 : 0);
    const int H = a.H, H4 = H >> 2;
    const long long n = (long long)a.B * H4;
    const long long idx = (long long)lb * 256 + threadIdx.x;
    if (idx >= n) return;
    const int b = (int)(idx / H4);
    const int j = (int)(idx - (long long)b * H4) * 4;
    float* dcp = a.dc + (long long)b * a.ld_dc + j;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = true;
    if (a.lens) valid = a.t < a.lens[b];
    if (!valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.dgates) *reinterpret_cast<float4*>(a.dgates + (long long)b * a.ld_dgates + j + q * H) = z4;
            if (a.dgates16) {
                const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
                cell_d16_store<false>(a, cell_d16_at(a, b, q * H + j), zero4);
            }
        }
        *reinterpret_cast<float4*>(dcp) = z4;
        return;
    }
    // issue every independent load before the first use
    const CellOperands r = cell_bwd_issue(a, b, j);
    const Slab4 s0 = addend_issue4(a.dh[0], b, j), s1 = addend_issue4(a.dh[1], b, j), s2 = addend_issue4(a.dh[2], b, j);
    const float4 d0 = addend_finish4(s0, a.dh[0], b, j), d1 = addend_finish4(s1, a.dh[1], b, j), d2 = addend_finish4(s2, a.dh[2], b, j);
    PW_TS(1);
    cell_bwd_finish(a, r, d0, d1, d2, b, j);
    PW_TS(2);
}

int t2amd_check_lstm_bwd_(const t2amd_lstm_bwd* a) {
    T2_REQUIRE(a && a->gates && a->c && a->dc && (a->dgates || a->dgates16), "lstm_bwd: null args (dgates may be NULL only beside dgates16)");
    T2_REQUIRE(a->B > 0 && a->H > 0 && a->H % 4 == 0, "lstm_bwd: H must be a positive multiple of 4");
    // 16-byte accesses: every row base and stride must keep 4-float alignment
    T2_REQUIRE(t2_aligned16(a->gates) && t2_aligned16(a->c) && t2_aligned16(a->dc) && (!a->dgates || (t2_aligned16(a->dgates) && a->ld_dgates % 4 == 0)) &&
                   a->ld_gates % 4 == 0 && a->ld_c % 4 == 0 && a->ld_dc % 4 == 0 &&
                   (!a->c_prev || (t2_aligned16(a->c_prev) && a->ld_cprev % 4 == 0)) &&
                   (!a->keep || ((reinterpret_cast<uintptr_t>(a->keep) & 3u) == 0 && a->ld_keep % 4 == 0)),
               "lstm_bwd: operands must be 16-byte aligned with strides % 4 == 0");
    for (int i = 0; i < 3; ++i)
        if (a->dh[i].p)
            T2_REQUIRE(t2_aligned16(a->dh[i].p) && a->dh[i].ld % 4 == 0 && a->dh[i].split_stride % 4 == 0,
                       "lstm_bwd: dh addends must be 16-byte aligned with strides % 4 == 0");
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_pointwise_bwd2_f32(const t2amd_lstm_bwd* a, const t2amd_lstm_bwd* b, void* stream) {
    T2_PROPAGATE(t2amd_check_lstm_bwd_(a));
    LstmBwdParams p;
    p.ts = t2amd_debug_ts_();
    p.a[0] = *a;
    p.nblk0 = t2_cdiv((long long)a->B * a->H / 4, 256);
    int total = p.nblk0;
    if (b) {
        T2_PROPAGATE(t2amd_check_lstm_bwd_(b));
        p.a[1] = *b;
        total += t2_cdiv((long long)b->B * b->H / 4, 256);
    } else {
        p.a[1] = *a;
    }
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 3; ++i)
            if (p.a[k].dh[i].p && p.a[k].dh[i].nsplit < 1) p.a[k].dh[i].nsplit = 1;
    T2_LAUNCH(lstm_pointwise_bwd_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_pointwise_bwd_f32(const t2amd_lstm_bwd* a, void* stream) {
    return t2amd_lstm_pointwise_bwd2_f32(a, nullptr, stream);
}
