// The 64 x 32 bf16 weight-streaming tile of the LSTM steps and BPTT dgrads (DESIGN 4.1 "wide variant") as a device function,
// with the problem description it works from.  Included by rnn.hip (one launch per step: skinny_wide_kernel) and by
// attention.hip (the persistent training-forward kernel: one launch per utterance batch, the tile once per time step).
#pragma once
#include "common.h"

#define SK_BK 64      // k per LDS tile
#define SK_ROWS 64    // batch rows per workgroup
#define SK_NBUF 3     // LDS ring: tile kt is multiplied while kt+1, kt+2 are in flight
#define SK_XT (SK_ROWS * SK_BK)   // floats per X tile
#define SK_WT (16 * SK_BK)        // floats per W tile

typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* t2_gptr;
typedef __attribute__((address_space(3))) void* t2_lptr;

struct SkinnyParams {
    t2amd_seg x[3];
    int nseg;
    const float* W;
    int Ktot, B;
    int H;            // LSTM: hidden size; plain: unused
    int N;            // plain: output columns
    // LSTM epilogue
    const float* gin; long long ld_gin;
    const float* bias;
    const float* c_prev; long long ld_cprev;
    float* gates_out; long long ld_gates;
    float* c_out; long long ld_c;
    float* h_out; long long ld_h;
    unsigned short* h16_out; long long ld_h16;   // optional bf16 copy of h (the next step's MFMA operand)
    const uint8_t* keep; long long ld_keep; float keep_scale;
    const int* lens; int t;
    // plain epilogue
    float* Y; long long ldy; int nsplit; long long split_stride; int ktiles_per_split;
    int act;          // plain epilogue: 1 = relu (bias / keep / keep_scale above are shared with the LSTM epilogue)
    uint8_t* stop_active; int* stop_lengths; int* stop_done;     // plain epilogue: decode stop test (t2amd_skinny_gemm)
    int stop_col, stop_max_steps; float stop_thr;                // (the step is `t` above)
    int gx, gy, gz;   // logical grid of this problem
    // K column of W at which segment i starts, or wcol[0] < 0 = "cumulative" (segment i starts where segment i-1 ended).  Explicit
    // columns let the segments be VISITED in another order than W stores them (wide bf16 tile only): the attention LSTM of the
    // training loop walks [h_att | ctx] over Wa_rec = [ctx columns | h_att columns], so that in the persistent loop the k-tiles of
    // h_att -- complete one phase earlier than ctx -- can run ahead of the wait for ctx (gate_seg below).
    int wcol[3];
    // persistent loops only: segments with index >= gate_seg are not touched (no DMA issued) before the caller's gate has opened;
    // 0 = no gate inside the tile; negative = the gate stands in front of the tile's FIRST DMA (every activation segment is what
    // the previous phase of the same launch wrote: the BPTT dgrad pair of the persistent backward loop).
    int gate_seg;
};

// Two independent problems in one launch (blocks [0, nblk0) -> p[0], the rest -> p[1]): the decoder
// LSTM of step t-1 rides along with the attention LSTM of step t (and likewise their BPTT dgrads),
// which puts two workgroups on every CU so that one's MFMAs cover the other's loads and barriers.
struct SkinnyDual { SkinnyParams p[2]; int nblk0; unsigned long long* ts; };

// The workgroup's problem, selected FIELD BY FIELD from the two kernel-argument copies.  Taking a reference to
// dp.p[second] makes every later field access a scalar load from a computed address, which the compiler issues one
// at a time and waits for (1.7 us of serial s_load round trips before the first DMA); selecting per field keeps
// every load at a constant kernarg offset, so they are batched into a few wide s_loads issued together.
__device__ __forceinline__ SkinnyParams skinny_select(const SkinnyDual& dp, bool second) {
    const SkinnyParams& a = dp.p[0];
    const SkinnyParams& b = dp.p[1];
    SkinnyParams p;
#define SK_SEL(F) p.F = second ? b.F : a.F
#pragma unroll
    for (int i = 0; i < 3; ++i) { SK_SEL(x[i].p); SK_SEL(x[i].ld); SK_SEL(x[i].width); }
    SK_SEL(nseg); SK_SEL(W); SK_SEL(Ktot); SK_SEL(B); SK_SEL(H); SK_SEL(N);
    SK_SEL(gin); SK_SEL(ld_gin); SK_SEL(bias); SK_SEL(c_prev); SK_SEL(ld_cprev);
    SK_SEL(gates_out); SK_SEL(ld_gates); SK_SEL(c_out); SK_SEL(ld_c); SK_SEL(h_out); SK_SEL(ld_h);
    SK_SEL(h16_out); SK_SEL(ld_h16); SK_SEL(keep); SK_SEL(ld_keep); SK_SEL(keep_scale); SK_SEL(lens); SK_SEL(t);
    SK_SEL(Y); SK_SEL(ldy); SK_SEL(nsplit); SK_SEL(split_stride); SK_SEL(ktiles_per_split); SK_SEL(act);
    SK_SEL(stop_active); SK_SEL(stop_lengths); SK_SEL(stop_done); SK_SEL(stop_col); SK_SEL(stop_max_steps); SK_SEL(stop_thr);
    SK_SEL(gx); SK_SEL(gy); SK_SEL(gz);
    SK_SEL(wcol[0]); SK_SEL(wcol[1]); SK_SEL(wcol[2]); SK_SEL(gate_seg);
#undef SK_SEL
    return p;
}

// Stop test of free-running decoding on a finished gate logit (reference model.py:439-444: strict >, the stopping frame is
// part of the output).  Same arithmetic as infer_finish_step_kernel (loops.hip), which it replaces at B > 8.
__device__ __forceinline__ void skinny_stop_test(const SkinnyParams& p, int row, float logit) {
    if (!p.stop_active[row]) return;
    const float sg = 1.0f / (1.0f + expf(-logit));
    if (sg > p.stop_thr || p.t + 1 >= p.stop_max_steps) {
        p.stop_lengths[row] = p.t + 1;
        p.stop_active[row] = 0;
        atomicAdd(p.stop_done, 1);
    }
}

// ---------------------------------------------------------------------------------------
// Wide bf16 variant: 64 batch rows x 32 output columns per workgroup, 8 waves.
//
// The 64x16 kernel above re-reads the activation tile once per 16 gate columns (L2 -> LDS through the CU's
// 64 B/clk vector-memory path) and every wave re-reads the whole weight tile from LDS; in bf16 mode, where HBM
// bytes are halved, those two become the bound.  Here a tile row is still 256 bytes (128 k), but the tile feeds
// 32 columns, and the eight waves split it as (row half i) x (k quarter kk): wave (i, kk) multiplies rows
// 32i..32i+31 by all 32 columns over k chunks 4kk..4kk+3 with two v_mfma_f32_32x32x16_bf16 -- every activation
// byte is read from LDS exactly once, every weight byte twice.  The four k-quarter partial sums are added
// through LDS once, after the loop.  96 KB of LDS (4-tile ring), one workgroup per CU: the fused LSTM pair and the
// split-K BPTT dgrad pair are 256 workgroups each.
// ---------------------------------------------------------------------------------------
#define SW_NBUF 4
#define SW_XB (64 * 256)    // bytes per activation tile
#define SW_WB (32 * 256)    // bytes per weight tile

#ifdef T2AMD_PHASE_STAMPS
#define SW_TS(slot)                                                                        \
    do {                                                                                   \
        if ((slot) == 0) ts_on = t2_ts_begin(ts_buf, LSTM ? 64 : 80);                      \
        else t2_ts_mark(ts_on, ts_buf, (LSTM ? 64 : 80) + (slot));                         \
    } while (0)
#else
#define SW_TS(slot) do { (void)ts_on; (void)ts_buf; } while (0)
#endif
// The body is a device function: skinny_wide_kernel (rnn.hip) runs it once per launch; the persistent training-forward kernel
// (attention.hip, dec_train_fwd_persistent_kernel) runs it once per decoder time step inside ONE launch.  PERSIST changes what
// another workgroup of the SAME launch reads or has written: the activation tiles are fetched with sc1 (L1-bypassing) LDS-DMA
// and h / its bf16 copy leave as write-through (sc1) stores, to be drained by the caller before it publishes its flag
// (Guideline 16 R1).  Same arithmetic in the same order either way: bit-identical results.
// `smem`: SW_NBUF * (SW_XB + SW_WB) bytes of LDS, 16-byte aligned; `lb`: the workgroup's index within problem `p`.
// Gate = what the persistent loop hands in so that the tile can start on the operand segments that are already complete and wait
// for the rest INSIDE the k loop:  early_issue() at entry (one poll, issued before any DMA: loads return in order, so it has
// landed by the time the first tile has), early_check() right behind the first tile's wait (no stall: everything older than that
// tile has returned), wait() once, at the workgroup-uniform point where the first DMA of a gated segment is about to be issued
// (a blocking poll by wave 0 only if the early one did not see every flag, then a barrier).  The slowest workgroup of the
// previous phase -- the one on the critical path -- finds all flags set in its early poll and never stalls; earlier finishers pay a
// drained DMA queue they have the slack for.  NoGate: the per-step kernels.
__device__ __forceinline__ int t2_tid_opaque() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
struct NoGate {
    static constexpr bool on = false;
    __device__ __forceinline__ void early_issue(int, int) {}
    __device__ __forceinline__ void early_check(int, int) {}
    __device__ __forceinline__ void wait(int, int) {}
};
// XPLAIN (PERSIST only): the activation tiles are fetched with ORDINARY DMA although another workgroup of the same launch wrote
// them.  Legal only where no stale copy can exist in this XCD's L2: the operand lives at an address that NOTHING read before it
// was written (a per-step slab), and every 128-byte line of it was written WHOLE by one wave (write-through: the line a writer's
// L2 may keep is complete and current).  The bf16 gate gradients of the BPTT loop are such an operand (cell_bwd_finish: a wave
// writes 512 contiguous bytes per gate row); h_att / h_dec of the forward loop are not (eight workgroups share a line).  What it
// buys: the 80 (48) column tiles that read the same 256 KB of gate gradients hit their XCD's L2 instead of going to the fabric
// 256 times per step -- 64 MB per step, which is what bounded the phase.
// `pref` (persistent FORWARD loop only, round 5): ring slots 0..3 already hold k-tiles 0..3 of segment 0 -- skinny_wide_prefetch4
// below issued them one phase ago, while the workgroup sat in its attention step, and the caller has drained and published them
// (s_waitcnt vmcnt(0) + barrier at the end of that phase).  The tile then starts multiplying at once: what used to stand between
// its entry and the first MFMA -- address set-up, four tiles of DMA issue and the first tile's L2 round trip, ~2.5 us of a ~9 us
// phase -- ran under the latency-bound attention step.  Same tiles in the same slots, same k order: bit-identical results.
// Requires n0 >= 5 (the caller checks skinny_wide_prefetch_ok).
// F32 (round 5: the fp32 parity mode's tile, so that its time loop can be the same persistent launch): the operands are f32 and
// the products run on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).  Everything byte-shaped is unchanged -- a
// tile row is still 256 bytes (64 k instead of 128), a 16-byte chunk is four k values instead of eight, the DMA, the swizzle and
// the fragment reads are the same instructions -- and a chunk pair feeds four MFMAs of k = 2 instead of one of k = 16: element e of
// the lane's X and W chunks is one k index in both, which is all the instruction asks for.
// MODE (round 6; was `bool F32`): SW_BF16 = 0, SW_F32 = 1 as above, SW_X3 = 2 -- SPLIT-bf16 operands (t2amd_split_bf16x3_f32 images: 4
// bytes per k, every 16 k stored as 16 hi then 16 lo bf16 -- the engine's 'bf16x3' mode).  In bytes such a row IS an f32 row: 64 k per
// 256-byte tile row, addresses, DMA, swizzle and fragment reads are the f32 form's.  What a wave reads as its two fragments (chunks 4 wk
// + {0, 1} and 4 wk + {2, 3} of a tile row = one 64-byte group) are then the hi halves and the lo halves of the SAME 16 k, so the
// step's two bf16 MFMAs X0.W0 + X1.W1 become three: hi.hi into acc0, lo.hi + hi.lo into acc1 (the small terms keep an accumulator
// of their own; the epilogue adds the two as it always did).  f32-class products (~2^-17 relative each, f32 accumulation) at the f32
// byte stream and 3 x 32 cycles per SIMD instead of 8 x 64: the k loop is bound by the LDS-DMA rate again, not by the matrix pipe.
#define SW_BF16 0
#define SW_F32 1
#define SW_X3 2
template <bool LSTM, bool PERSIST, class Gate, bool XPLAIN = false, int MODE = SW_BF16>
__device__ __forceinline__ void skinny_wide_body(const SkinnyParams& p, const int lb, char* const smem, unsigned long long* const ts_buf,
                                                 Gate& gate, const bool pref = false) {
    constexpr bool F32 = MODE == SW_F32;
    constexpr bool X3 = MODE == SW_X3;
    constexpr int BK = MODE == SW_BF16 ? 128 : 64;       // k per 256-byte tile row
    constexpr int ES = MODE == SW_BF16 ? 2 : 4;          // bytes per k of an operand row (X3: a hi and a lo bf16)
    constexpr int CE = 16 / ES;              // k per 16-byte chunk (address arithmetic; X3: a chunk HOLDS 8 hi or 8 lo values)
    constexpr int XAUX = (PERSIST && !XPLAIN) ? 16 : 0;       // aux bit 4 = sc1 on the activation DMA
    bool ts_on = false;
    SW_TS(0);
    char* const Xs = smem;                       // [NBUF][64][256 B]
    char* const Ws = smem + SW_NBUF * SW_XB;     // [NBUF][32][256 B]
    float* const Ps = reinterpret_cast<float*>(smem);   // epilogue: [4][64][33] partial sums (aliases the ring)

    const int bx = lb % p.gx;
    const int by = (lb / p.gx) % p.gy;
    const int bz = lb / (p.gx * p.gy);

    // (persistent backward loop: an opaque thread index per call -- otherwise everything derived from it is hoisted out of the
    // loop over the time steps and spilled, ~100 scratch reloads per step)
    const int tid = (PERSIST && !LSTM) ? t2_tid_opaque() : (int)threadIdx.x;
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
    const bool wexp = p.wcol[0] >= 0;
    const int wo0 = wexp ? p.wcol[0] : 0;
    const int wo1 = wexp ? p.wcol[1] : p.x[0].width, wo2 = wexp ? p.wcol[2] : p.x[0].width + (p.nseg > 1 ? p.x[1].width : 0);
    const int nvt = n0 + n1 + n2;
    int kt_beg = 0, kt_end = nvt;
    if (!LSTM) {
        kt_beg = bz * p.ktiles_per_split;
        kt_end = kt_beg + p.ktiles_per_split;
        if (kt_end > nvt) kt_end = nvt;
    }

    // LSTM epilogue operands, fetched up front: thread -> (row = tid>>3, unit = tid&7)
    const int erow = tid >> 3, ejj = tid & 7;
    const int egr = rowbase + erow;
    const int ej = bx * 8 + ejj;
    // (raw values only: a comparison here would make the compiler wait for the load before the first DMA)
    float e_gin[4] = {0.f, 0.f, 0.f, 0.f}, e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_cp = 0.f;
    int e_keep_raw = 1;
    int e_len = 0x7fffffff;
    // Issued BEHIND the first tile's DMA (round 3; T2AMD_SW_EPI_FIRST=1 at build time restores "before"): the eleven loads
    // and their 64-bit address arithmetic used to stand between kernel entry and the first weight byte (1.5 us to the first
    // DMA against 0.7 us in the dgrad form, which has no such operands).  The counted vmcnt waits stay correct wherever the
    // compiler finally places these loads: returns are in order, so a wait can only cover more than it needs, never less.
#define SW_LOAD_EPI()                                                                                  \
    if (LSTM && egr < B) {                                                                             \
        const int H = p.H;                                                                             \
        const int* lens_ = p.lens;                                                                     \
        const float* gin_ = p.gin;                                                                     \
        const float* bias_ = p.bias;                                                                   \
        const float* cprev_ = p.c_prev;                                                                \
        const uint8_t* keep_ = p.keep;                                                                 \
        const long long ld_gin_ = p.ld_gin, ld_cprev_ = p.ld_cprev, ld_keep_ = p.ld_keep;              \
        if (lens_) e_len = lens_[egr];                                                                 \
        if (gin_) {                                                                                    \
            const float* g = gin_ + (long long)egr * ld_gin_;                                          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) e_gin[q] = g[q * H + ej];                    \
        }                                                                                              \
        if (bias_) {                                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) e_bias[q] = bias_[q * H + ej];               \
        }                                                                                              \
        if (cprev_) e_cp = cprev_[(long long)egr * ld_cprev_ + ej];                                    \
        if (keep_) e_keep_raw = keep_[(long long)egr * ld_keep_ + ej];                                 \
    }
#ifdef T2AMD_SW_EPI_FIRST
    SW_LOAD_EPI()
#endif

    // DMA sources.  X: instruction q of this wave fills tile rows 8*wave + 4q + lg, LDS chunk l15 <- global
    // chunk l15 ^ (row & 15).  W: this wave fills weight-tile rows 4*wave + lg.
    long long xo0[2], xo1[2], xo2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * wave + 4 * q + lg;
        int gr = rowbase + r;
        if (gr > B - 1) gr = B - 1;
        const int c8 = CE * (l15 ^ (r & 15));
        xo0[q] = ((long long)gr * p.x[0].ld + c8) * ES;
        xo1[q] = ((long long)gr * p.x[1].ld + c8) * ES;
        xo2[q] = ((long long)gr * p.x[2].ld + c8) * ES;
    }
    long long wo;
    {
        const int c = 4 * wave + lg;                  // tile column 0..31
        long long wrow;
        if (LSTM) {
            wrow = (long long)(c >> 3) * p.H + bx * 8 + (c & 7);     // column = gate*8 + unit
        } else {
            wrow = (long long)bx * 32 + c;
            if (wrow > p.N - 1) wrow = p.N - 1;
        }
        wo = (wrow * p.Ktot + CE * (l15 ^ (c & 15))) * ES;
    }
    const char* const Wp = reinterpret_cast<const char*>(p.W) + wo;
    const char* const xp0 = reinterpret_cast<const char*>(p.x[0].p);
    const char* const xp1 = reinterpret_cast<const char*>(p.x[1].p);
    const char* const xp2 = reinterpret_cast<const char*>(p.x[2].p);
    const int kt_last = kt_end - 1;

    const char* xq0; const char* xq1; const char* wq;
    int iss_kt = kt_beg, iss_seg = 0, iss_rem = 0;
#define SW_SEEK(SEG, LOCAL)                                                                            \
    {                                                                                                  \
        const int seg_ = (SEG), loc_ = (LOCAL);                                                        \
        if (seg_ == 0) {                                                                               \
            const char* sp_ = xp0 + loc_ * 256;                                                        \
            xq0 = sp_ + xo0[0]; xq1 = sp_ + xo0[1];                                                    \
            wq = Wp + wo0 * ES + loc_ * 256; iss_rem = n0 - loc_;                                       \
        } else if (seg_ == 1) {                                                                        \
            const char* sp_ = xp1 + loc_ * 256;                                                        \
            xq0 = sp_ + xo1[0]; xq1 = sp_ + xo1[1];                                                    \
            wq = Wp + wo1 * ES + loc_ * 256; iss_rem = n1 - loc_;                                       \
        } else {                                                                                       \
            const char* sp_ = xp2 + loc_ * 256;                                                        \
            xq0 = sp_ + xo2[0]; xq1 = sp_ + xo2[1];                                                    \
            wq = Wp + wo2 * ES + loc_ * 256; iss_rem = n2 - loc_;                                       \
        }                                                                                              \
        iss_seg = seg_;                                                                                \
    }
    xq0 = xq1 = wq = reinterpret_cast<const char*>(p.W);
    if (kt_end > kt_beg) {
        if (pref) { SW_SEEK(0, 4) iss_kt = kt_beg + 4; }          // (n0 >= 5: tile 4 is still in segment 0)
        else if (kt_beg < n0) SW_SEEK(0, kt_beg)
        else if (kt_beg < n0 + n1) SW_SEEK(1, kt_beg - n0)
        else SW_SEEK(2, kt_beg - n0 - n1)
    }

    bool gate_armed = Gate::on && p.gate_seg != 0;
    if constexpr (Gate::on) gate.early_issue(wave, lane);
#define SW_ISSUE(BUF)                                                                                  \
    {                                                                                                  \
        if constexpr (Gate::on) {                                                                      \
            if (gate_armed && iss_seg >= p.gate_seg) { gate.wait(wave, lane); gate_armed = false; }    \
        }                                                                                              \
        char* xd_ = Xs + (BUF) * SW_XB + wave * (8 * 256);                                             \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq0), (t2_lptr)(xd_), 16, 0, XAUX);                 \
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq1), (t2_lptr)(xd_ + 1024), 16, 0, XAUX);          \
        __builtin_amdgcn_global_load_lds((t2_gptr)(wq), (t2_lptr)(Ws + (BUF) * SW_WB + wave * 1024), 16, 0, 0); \
        if (iss_kt < kt_last) {                                                                        \
            ++iss_kt;                                                                                  \
            if (--iss_rem > 0) {                                                                       \
                xq0 += 256; xq1 += 256; wq += 256;                                                     \
            } else if (iss_seg == 0 && n1 > 0) {                                                       \
                SW_SEEK(1, 0)                                                                          \
            } else {                                                                                   \
                SW_SEEK(2, 0)                                                                          \
            }                                                                                          \
        }                                                                                              \
    }

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // fragment addresses: MFMA m of this wave uses k chunk 4*wk + 2m + lhi of row (32*wi + l31) / column l31
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
    f32x4 xa0, xa1, wa0, wa1, xb0, xb1, wb0, wb1;

#define SW_READ(BUF, X0, X1, W0, W1)                                                                   \
    asm volatile(                                                                                      \
        "ds_read_b128 %0, %4 offset:%8\n\t"                                                            \
        "ds_read_b128 %2, %6 offset:%9\n\t"                                                            \
        "ds_read_b128 %1, %5 offset:%8\n\t"                                                            \
        "ds_read_b128 %3, %7 offset:%9"                                                                \
        : "=&v"(X0), "=&v"(X1), "=&v"(W0), "=&v"(W1)                                                   \
        : "v"(ax[0]), "v"(ax[1]), "v"(aw[0]), "v"(aw[1]), "i"((BUF) * SW_XB), "i"((BUF) * SW_WB)       \
        : "memory");
#define SW_WAITR(X0, X1, W0, W1)                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X0), "+v"(X1), "+v"(W0), "+v"(W1) : : "memory");
#define SW_FMA(X0, X1, W0, W1)                                                                         \
    if constexpr (F32) {                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((X0)[e_], (W0)[e_], acc0, 0, 0, 0);            \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((X1)[e_], (W1)[e_], acc1, 0, 0, 0);            \
        }                                                                                              \
    } else if constexpr (X3) {                                                                         \
        /* (the two small-term products are kept apart by the big one: no back-to-back dependent pair) */ \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X1)),            \
                                                       __builtin_bit_cast(sk_bf16x8, (W0)), acc1, 0, 0, 0); \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X0)),            \
                                                       __builtin_bit_cast(sk_bf16x8, (W0)), acc0, 0, 0, 0); \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X0)),            \
                                                       __builtin_bit_cast(sk_bf16x8, (W1)), acc1, 0, 0, 0); \
    } else {                                                                                           \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X0)),            \
                                                       __builtin_bit_cast(sk_bf16x8, (W0)), acc0, 0, 0, 0); \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, (X1)),            \
                                                       __builtin_bit_cast(sk_bf16x8, (W1)), acc1, 0, 0, 0); \
    }
#define SW_SETA xa0, xa1, wa0, wa1
#define SW_SETB xb0, xb1, wb0, wb1
#define SW_X(M, ...) M(__VA_ARGS__)
    // Same protocol as the 64x16 kernel: every wave issues exactly 3 DMA instructions per tile, so "tile KT+1
    // landed" is vmcnt(6) (tiles KT+2, KT+3 may be pending); the barrier publishes it and frees tile KT's buffer for
    // the DMA of tile KT+4.
    // WAIT false: the tile this step publishes was prefetched and drained a phase ago (the first three steps of a `pref` tile)
#define SW_STEP_W(BUF, CUR, NXT, WAIT)                                       \
    {                                                                        \
        if (WAIT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SW_X(SW_READ, ((BUF) + 1) % SW_NBUF, NXT)                            \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SW_ISSUE(BUF)                                                        \
        SW_X(SW_FMA, CUR)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
        SW_X(SW_WAITR, NXT)                                                  \
    }

    if (kt_end > kt_beg) {
        SW_TS(1);
        if (!pref) {
            SW_ISSUE(0)
#ifndef T2AMD_SW_EPI_FIRST
            __builtin_amdgcn_sched_barrier(0);
            SW_LOAD_EPI()
            __builtin_amdgcn_sched_barrier(0);
#endif
            SW_ISSUE(1)
            SW_ISSUE(2)
            SW_ISSUE(3)
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            if constexpr (Gate::on) gate.early_check(wave, lane);       // (older than tile 0's DMA: it has returned)
        } else {
#ifndef T2AMD_SW_EPI_FIRST
            __builtin_amdgcn_sched_barrier(0);
            SW_LOAD_EPI()
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SW_TS(2);
        SW_X(SW_READ, 0, SW_SETA)
        SW_X(SW_WAITR, SW_SETA)
        int kt = kt_beg;
        bool first = pref;                       // (workgroup-uniform)
        for (; kt + 4 <= kt_end; kt += 4) {
            SW_STEP_W(0, SW_SETA, SW_SETB, !first)
            SW_STEP_W(1, SW_SETB, SW_SETA, !first)
            SW_STEP_W(2, SW_SETA, SW_SETB, !first)
            SW_STEP_W(3, SW_SETB, SW_SETA, true)
            if constexpr (Gate::on) {
                // the entry poll is older than tile 4's DMA, whose wait stands in the step above: it has returned
                if (first) gate.early_check(wave, lane);
            }
            first = false;
        }
        if (kt < kt_end) {
            SW_STEP_W(0, SW_SETA, SW_SETB, true)
            if (kt + 1 < kt_end) {
                SW_STEP_W(1, SW_SETB, SW_SETA, true)
                if (kt + 2 < kt_end) SW_STEP_W(2, SW_SETA, SW_SETB, true)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped duplicate tiles
    }
#ifndef T2AMD_SW_EPI_FIRST
    else { SW_LOAD_EPI() }
#endif
#undef SW_LOAD_EPI
#undef SW_ISSUE
#undef SW_SEEK
#undef SW_READ
#undef SW_WAITR
#undef SW_FMA
#undef SW_STEP_W
#undef SW_X

    // k-quarter partial sums -> LDS (the ring is dead once every wave's DMA has drained)
    SW_TS(3);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Ps[(wk * 64 + row) * 33 + l31] = acc0[r] + acc1[r];
    }
    __syncthreads();
    SW_TS(4);

    if (!LSTM) {
        // thread -> (row = tid>>3, 4 consecutive columns)
        const int gr = rowbase + erow;
        if (gr >= B) return;
        float* __restrict__ Y = p.Y + (long long)bz * p.split_stride + (long long)gr * p.ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ejj * 4 + j;
            const int gn = bx * 32 + c;
            const float v = (Ps[erow * 33 + c] + Ps[(64 + erow) * 33 + c]) + (Ps[(128 + erow) * 33 + c] + Ps[(192 + erow) * 33 + c]);
            if (gn < p.N) {
                float o = v;
                if (p.bias) o += p.bias[gn];
                if (p.act == 1) o = fmaxf(o, 0.f);
                if (p.keep) o = p.keep[(long long)gr * p.ld_keep + gn] ? o * p.keep_scale : 0.f;
                if constexpr (PERSIST) __hip_atomic_store(&Y[gn], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the next phase of this launch
                else Y[gn] = o;
                if (p.h16_out) {
                    if constexpr (X3) {
                        unsigned short hi_, lo_;
                        t2_split_bf16(o, hi_, lo_);
                        unsigned short* const q_ = p.h16_out + (long long)gr * p.ld_h16 * 2 + t2_x3_pos(gn);
                        q_[0] = hi_; q_[16] = lo_;
                    } else {
                        p.h16_out[(long long)gr * p.ld_h16 + gn] = t2_f32_to_bf16(o);
                    }
                }
                if (p.stop_active && gn == p.stop_col) skinny_stop_test(p, gr, o);
            }
        }
        SW_TS(5);
        return;
    }
    if (egr >= B) return;
    const int H = p.H;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, hn = 0.f;
    asm volatile("" : "+v"(e_keep_raw), "+v"(e_len));   // keeps the comparisons (and the wait for the loads) down here
    const bool e_valid = p.t < e_len, e_keep = e_keep_raw != 0;
    if (e_valid) {
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 8 + ejj;
            pre[q] = ((Ps[erow * 33 + c] + Ps[(64 + erow) * 33 + c]) + (Ps[(128 + erow) * 33 + c] + Ps[(192 + erow) * 33 + c]))
                     + e_gin[q] + e_bias[q];
        }
        gi = t2_sigmoid_fast(pre[0]);
        gf = t2_sigmoid_fast(pre[1]);
        gg = t2_tanh(pre[2]);
        go = t2_sigmoid_fast(pre[3]);
        cn = gf * e_cp + gi * gg;
        hn = go * t2_tanh(cn);
        if (p.keep) hn = e_keep ? hn * p.keep_scale : 0.f;
    }
    if (p.gates_out) {           // NULL (round 6): free-running decode -- only a backward reads the gate activations
        float* go_ = p.gates_out + (long long)egr * p.ld_gates;
        go_[ej] = gi;
        go_[H + ej] = gf;
        go_[2 * H + ej] = gg;
        go_[3 * H + ej] = go;
    }
    p.c_out[(long long)egr * p.ld_c + ej] = cn;
    if constexpr (PERSIST) {
        // read by other workgroups of this launch (attention: h; the next step's LSTM tiles: the bf16 copy): write-through.
        // (round 5, measured and NOT adopted -- -DT2AMD_SW_VECTOR_WT builds it) As 16-BYTE stores: the eight units of a row sit in
        // eight consecutive lanes, so lane 0 of the group collects them (seven lane shifts) and writes the bf16 copy as ONE store and,
        // with lane 4, h as two -- 192 instead of 1024 fabric writes per tile and step in front of the drain the flag waits for
        // (MI355X_MICROARCH.md: per byte a scalar dword write-through store costs ~6 x, a short ~12.5 x a dwordx4).  Same bits, but
        // the forward loop measured 22.65 vs 22.0 ms with it (three alternating pairs of builds): the seven ds_bpermute per lane and
        // their waits sit on the tile's critical tail, and the drain was not waiting for the NUMBER of writes.
#ifndef T2AMD_SW_VECTOR_WT
        __hip_atomic_store(&p.h_out[(long long)egr * p.ld_h + ej], hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (X3) {
            if (p.h16_out) {
                unsigned short hi_, lo_;
                t2_split_bf16(hn, hi_, lo_);
                unsigned short* const q_ = p.h16_out + (long long)egr * p.ld_h16 * 2 + t2_x3_pos(ej);
                __hip_atomic_store(q_, hi_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q_ + 16, lo_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else
        if (p.h16_out) __hip_atomic_store(&p.h16_out[(long long)egr * p.ld_h16 + ej], t2_f32_to_bf16(hn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        float hv[8];
        hv[0] = hn;
#pragma unroll
        for (int k = 1; k < 8; ++k) hv[k] = __shfl_down(hn, k, 64);      // (lanes of other rows / past the wave: never used)
        float* const hrow = p.h_out + (long long)egr * p.ld_h + ej;       // ejj 0 -> units 0..3, ejj 4 -> units 4..7
        if ((ejj & 3) == 0) {
            const f32x4 v = {hv[0], hv[1], hv[2], hv[3]};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(hrow), "v"(v) : "memory");
        }
        if (p.h16_out && ejj == 0) {
            // (the chain's own conversion, bit for bit)
            const sk_u32x4 pk = {(unsigned)t2_f32_to_bf16(hv[0]) | ((unsigned)t2_f32_to_bf16(hv[1]) << 16),
                                 (unsigned)t2_f32_to_bf16(hv[2]) | ((unsigned)t2_f32_to_bf16(hv[3]) << 16),
                                 (unsigned)t2_f32_to_bf16(hv[4]) | ((unsigned)t2_f32_to_bf16(hv[5]) << 16),
                                 (unsigned)t2_f32_to_bf16(hv[6]) | ((unsigned)t2_f32_to_bf16(hv[7]) << 16)};
            unsigned short* const h16row = p.h16_out + (long long)egr * p.ld_h16 + ej;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(h16row), "v"(pk) : "memory");
        }
#endif
    } else {
        p.h_out[(long long)egr * p.ld_h + ej] = hn;
        if constexpr (X3) {
            if (p.h16_out) {
                unsigned short hi_, lo_;
                t2_split_bf16(hn, hi_, lo_);
                unsigned short* const q_ = p.h16_out + (long long)egr * p.ld_h16 * 2 + t2_x3_pos(ej);
                q_[0] = hi_; q_[16] = lo_;
            }
        } else
        if (p.h16_out) p.h16_out[(long long)egr * p.ld_h16 + ej] = t2_f32_to_bf16(hn);
    }
    SW_TS(5);
}

// ---------------------------------------------------------------------------------------
// Persistent forward loop (round 5): k-tiles 0..3 of segment 0 of the NEXT step's LSTM tile, issued from inside the attention
// phase -- the workgroup's tile of step t+1 multiplies [h_att(t) | ...] first in both roles, and h_att(t) is complete the moment the
// attention phase has seen the LSTM flags of step t.  Every wave issues the very 12 DMA instructions skinny_wide_body would issue at
// its entry (SW_ISSUE(0..3)): same sources, same LDS image.  `x0`: bf16 rows of segment 0 (read with sc1: another workgroup of
// this launch wrote them), `ld0` in elements; `wcol0`: W's K column of segment 0; `bx`: the tile's index; B rows (<= 64).
// ---------------------------------------------------------------------------------------
// (`bk`: k per tile row -- 128 for bf16 operands, 64 for f32 and for split-bf16 images, whose rows are f32 rows in bytes: F32 = true)
__device__ __forceinline__ bool skinny_wide_prefetch_ok(const int width0, const int ntiles, const int bk = 128) { return width0 >= 5 * bk && ntiles >= 8 && (ntiles & 3) == 0; }
template <bool F32 = false>
__device__ __forceinline__ void skinny_wide_prefetch4(const void* const x0, const long long ld0, const void* const W,
                                                      const int Ktot, const int H, const int wcol0, const int B, const int bx,
                                                      char* const smem) {
    constexpr int ES = F32 ? 4 : 2, CE = 16 / ES;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    char* const Xs = smem;
    char* const Ws = smem + SW_NBUF * SW_XB;
    const char* xq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * wave + 4 * q + lg;
        int gr = r;
        if (gr > B - 1) gr = B - 1;
        const int c8 = CE * (l15 ^ (r & 15));
        xq[q] = reinterpret_cast<const char*>(x0) + ((long long)gr * ld0 + c8) * ES;
    }
    const int c = 4 * wave + lg;
    const long long wrow = (long long)(c >> 3) * H + bx * 8 + (c & 7);
    const char* wq = reinterpret_cast<const char*>(W) + (wrow * Ktot + CE * (l15 ^ (c & 15))) * ES + (long long)wcol0 * ES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        char* const xd = Xs + s * SW_XB + wave * (8 * 256);
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq[0] + s * 256), (t2_lptr)(xd), 16, 0, 16);            // aux bit 4 = sc1
        __builtin_amdgcn_global_load_lds((t2_gptr)(xq[1] + s * 256), (t2_lptr)(xd + 1024), 16, 0, 16);
        __builtin_amdgcn_global_load_lds((t2_gptr)(wq + s * 256), (t2_lptr)(Ws + s * SW_WB + wave * 1024), 16, 0, 0);
    }
}
