// Persistent, weight-stationary decode loop for ONE utterance (BASELINE config 4: Tacotron2.inference, batch 1,
// greedy decode to the gate stop).  Replaces the whole of Decoder.inference's loop (reference model.py:435-449
// around Decoder.decode :340-379, Prenet :97-100, Attention :43-86, LocationLayer :22-26) with ONE launch.
//
// Why: at B = 1 a decode step is a chain of matrix-VECTOR products -- 36.4 MB of bf16 weights per step, every byte
// used once -- and seven dependent launches (40 us per step measured, of which the weight stream is ~6 us).  The
// step weights fit the chip's LDS: H/4 = 256 workgroups (one per CU, all co-resident) each keep the 4 x {i,f,g,o}
// rows of BOTH LSTMs they own (16 x 1792 + 16 x 2560 bf16 = 136 KB) in LDS for the whole utterance, their slice of
// the small matrices in registers, and their slice of the recurrent state in registers.  Nothing is streamed from
// HBM inside the loop; what remains per step is six all-to-all hand-offs of small vectors:
//     p2 -> [LSTM_a] -> h_a -> [energies: 8 dim teams x H/32 position groups] -> partial energies -> [softmax + context slice, every CU]
//        -> ctx -> [LSTM_d] -> h_d -> [frame/gate rows + prenet layer 1 (folded through the frame projection)]
//        -> p1 (+ stop flag) -> [prenet layer 2] -> p2 ...
// Every hand-off is the guide's R2 form (cdna_hip_programming.md Guideline 16): 8-byte {tag = step + 1, f32 value}
// granules written by ONE relaxed agent-scope store (sc1) and polled with relaxed agent-scope loads -- the data is
// the flag, no fences.  EVERY workgroup sweeps EVERY mailbox once per step, in the same order; since each mailbox's
// next producers depend (through the chain above) on values that every workgroup can only have produced after that
// sweep, a mailbox is never overwritten before all its readers are done: no double buffering, no barrier.
// All spins are bounded by the 100 MHz wall clock; a timeout (e.g. fewer than H/4 CUs free) sets `status`, every
// workgroup leaves, and the host falls back to the launch chain (loops.hip) -- loudly.
//
// Arithmetic: bf16 LSTM weight rows against f32 inputs, f32 accumulation, f32 state, cell and outputs -- the same
// operand precision as the launch chain's bf16 mode for B <= 8 (t2amd_lstm_step.bf16 == 2); prenet, projection,
// query and location weights stay f32.  Prenet layer 1 is folded through the frame projection,
// p1 = relu(W1 (Wp hc + bp)) = relu((W1 Wp) hc + W1 bp), which removes one hand-off (frame -> p1) from the chain:
// the rows of [W1 Wp ; Wp ; Wg] are one distributed matrix-vector product.
#include "common.h"
#include <stdlib.h>

#define PB_NT 256
#define PB_POLLERS 256                   // every thread polls
#define PB_COMM 0                        // first thread of the wave that publishes (cells, context, projection rows): wave 0
                                         // (wave 3 measured 17.9 vs 16.4 us per step)
#define PB_TEAMS 8                       // attention teams: 16 of the 128 attention dims each
#define PB_TDIM (T2AMD_ATT_DIM / PB_TEAMS)
#define PB_PPW 8                         // positions per workgroup in the energy phase: Ti <= 8 * (workgroups / 8)
#define PB_MAXFR 6                       // rows of the folded projection per workgroup
#define PB_MAXKPT 6                      // (H + E) / 256 elements of such a row per thread
#define PB_MAXP2R 2                      // prenet layer-2 rows per workgroup
#define PB_MAXEPW 4                      // context channels per workgroup
#define PB_MAXQ 64                       // H / 16 W_q elements per energy thread
#define PB_HALO 15
#define PB_TIMEOUT_TICKS 3000000ll       // 30 ms of the 100 MHz wall clock per wait (default; PersistParams.timeout_ticks)

typedef unsigned long long pb_u64;

struct PersistParams {
    t2amd_dec_persist a;
    int nwg, tip;
    // s_sleep units (64 clocks) before the first poll of each mailbox (p2, h_a, energies, ctx, h_d, p1) and between the
    // polls of the singly-polled ones: the 256 CUs would otherwise hammer a mailbox's few cache lines while the stores
    // into them are still on their way (measured: 16 units before the p2 sweep alone took 17.3 -> 15.2 us off a step)
    int delay[6], poll_sleep;
    long long timeout_ticks;      // bounded spins: ticks of the 100 MHz wall clock per wait
};

// element k of a vector staged for the LSTM dot products: units of 8 consecutive k are split into two float4 halves
// (first half of the buffer: elements 0-3 of every unit, second half: elements 4-7), so that a lane's two 16-byte
// reads per unit are both lane-consecutive (conflict-free ds_read_b128)
__device__ __forceinline__ int pb_xoff(int k, int len) { return ((k & 4) ? (len >> 1) : 0) + ((k >> 3) << 2) + (k & 3); }

__device__ __forceinline__ void pb_publish(pb_u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((pb_u64)tag << 32) | (pb_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bounded spins: true when this wave must stop waiting -- its own 30 ms ran out (it then records the timeout) or any
// workgroup has already given up.  The decision is wave-uniform.
__device__ __forceinline__ bool pb_give_up(long long t0, int* status, long long limit) {
    const bool late = (long long)wall_clock64() - t0 > limit;
    const bool other = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (!__any(late || other)) return false;
    if (late && !other && (threadIdx.x & 63) == 0) atomicCAS(status, 0, T2AMD_PERSIST_TIMEOUT);
    return true;
}

// Every thread polls its own granules (tid, tid + 256, ...: NPT of them, all loads of a poll in flight together) until
// all of a wave's carry `tag`; values go to LDS.  Returns true on failure (timeout or another workgroup already failed).
// Wave-uniform control flow.
// (Measured and dropped: a communicator wave that issues every global store while only the other three waves poll --
// the idea being that a poll queued behind a write-through store is retired only after the store's acknowledgement --
// was slower, 18.0 vs 16.4 us per step: six granules per polling thread instead of four cost more than the stores did.)
template <bool SPLIT, int NPT, bool DBL>
__device__ __forceinline__ bool pb_sweep(const pb_u64* __restrict__ g, int n, unsigned tag, float* __restrict__ dst, int len,
                                         int* status, int tid, int poll_sleep, long long timeout) {
    // DBL: two polls in flight -- while poll A's loads are checked, poll B's are already on their way, so a publication is
    // seen half a round trip earlier.  Measured: pays for the 256-granule edges (p1: 2.4 -> 1.0 us), costs on the 1024-
    // granule ones (h_a, h_d, ctx: the doubled poll traffic of 256 CUs slows every round trip), so those poll singly.
    // (granules past n: the index is clamped -- an unconditional load of a valid word, no branch -- and ignored)
    pb_u64 xa[NPT], xb[NPT];
    const pb_u64* gp[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) gp[j] = g + ((tid + PB_POLLERS * j < n) ? tid + PB_POLLERS * j : n - 1);
    const long long t0 = wall_clock64();
    unsigned spins = 0;
#define PB_POLL(X)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NPT; ++j) X[j] = __hip_atomic_load(gp[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define PB_READY(X, OK)                                                                                         \
    OK = true;                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NPT; ++j) OK = OK && ((unsigned)(X[j] >> 32) == tag);
    PB_POLL(xa)
    for (;;) {
        bool ok;
        if constexpr (DBL) {
            PB_POLL(xb)
            PB_READY(xa, ok)
            if (__all(ok)) break;
            PB_POLL(xa)
            PB_READY(xb, ok)
            if (__all(ok)) {
#pragma unroll
                for (int j = 0; j < NPT; ++j) xa[j] = xb[j];
                break;
            }
        } else {
            PB_READY(xa, ok)
            if (__all(ok)) break;
            for (int d_ = 0; d_ < poll_sleep; ++d_) __builtin_amdgcn_s_sleep(1);
            PB_POLL(xa)
        }
        if ((++spins & 31u) == 0 && pb_give_up(t0, status, timeout)) return true;
    }
#undef PB_POLL
#undef PB_READY
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + PB_POLLERS * j;
        if (i < n) dst[SPLIT ? pb_xoff(i, len) : i] = __uint_as_float((unsigned)xa[j]);
    }
    return false;
}

// acc[u] += W_s[row u of this wave][seg_off + k] * x[k], k < seg_len (multiple of 8): bf16 rows in LDS, f32 x in LDS
__device__ __forceinline__ void pb_dot_seg(const unsigned short* __restrict__ Wrows /* this wave's 4 rows */, int K, int seg_off,
                                           int seg_len, const float* __restrict__ xs, float (&acc)[4], int lane) {
    const int n8 = seg_len >> 3, half = seg_len >> 1;
    for (int kk = lane; kk < n8; kk += 64) {
        const float4 xa = *reinterpret_cast<const float4*>(xs + kk * 4);
        const float4 xb = *reinterpret_cast<const float4*>(xs + half + kk * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint4 w = *reinterpret_cast<const uint4*>(Wrows + (size_t)u * K + seg_off + kk * 8);
            float s = acc[u];
            s = fmaf(__uint_as_float(w.x << 16), xa.x, s); s = fmaf(__uint_as_float(w.x & 0xffff0000u), xa.y, s);
            s = fmaf(__uint_as_float(w.y << 16), xa.z, s); s = fmaf(__uint_as_float(w.y & 0xffff0000u), xa.w, s);
            s = fmaf(__uint_as_float(w.z << 16), xb.x, s); s = fmaf(__uint_as_float(w.z & 0xffff0000u), xb.y, s);
            s = fmaf(__uint_as_float(w.w << 16), xb.z, s); s = fmaf(__uint_as_float(w.w & 0xffff0000u), xb.w, s);
            acc[u] = s;
        }
    }
}

// ---- exact-f32 weights (t2amd_dec_persist.weights_f32) --------------------------------------------------------------
// A row's LDS / register image is laid out segment by segment in the SAME split order as the staged input vectors
// (pb_xoff), so a dot product is an element-wise walk over two identically permuted arrays: lane l takes the float4s
// l, l + 64, ... of the segment -- lane-consecutive 16-byte LDS reads on both sides.
// position `pos` of a split segment of length `len` -> the k it holds (pos is a multiple of 4: four consecutive k)
__device__ __forceinline__ int pb_xinv4(int pos, int len) {
    const int half = len >> 1;
    return pos < half ? (pos >> 2) * 8 : ((pos - half) >> 2) * 8 + 4;
}
// acc[u] += image row u [seg_off .. seg_off + seg_len) . xs: f32 rows in LDS
__device__ __forceinline__ void pb_dot_seg_f32(const float* __restrict__ Wrows, int K, int seg_off, int seg_len,
                                               const float* __restrict__ xs, float (&acc)[4], int lane) {
    const int n4 = seg_len >> 2;
    for (int i4 = lane; i4 < n4; i4 += 64) {
        const float4 x = *reinterpret_cast<const float4*>(xs + i4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w = *reinterpret_cast<const float4*>(Wrows + (size_t)u * K + seg_off + i4 * 4);
            float s = acc[u];
            s = fmaf(w.x, x.x, s); s = fmaf(w.y, x.y, s); s = fmaf(w.z, x.z, s); s = fmaf(w.w, x.w, s);
            acc[u] = s;
        }
    }
}
#define PB_RJ 4                          // float4s per lane, row and 1024-wide segment held in registers (H, E <= 1024)
// the same against register-resident rows: wr[u][j] = float4 (lane + 64 j) of row u's segment
__device__ __forceinline__ void pb_dot_reg(const float4 (&wr)[4][PB_RJ], int nj, const float* __restrict__ xs,
                                           float (&acc)[4], int lane) {
#pragma unroll
    for (int j = 0; j < PB_RJ; ++j) {
        if (j < nj) {
            const float4 x = *reinterpret_cast<const float4*>(xs + (lane + 64 * j) * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s = acc[u];
                s = fmaf(wr[u][j].x, x.x, s); s = fmaf(wr[u][j].y, x.y, s); s = fmaf(wr[u][j].z, x.z, s); s = fmaf(wr[u][j].w, x.w, s);
                acc[u] = s;
            }
        }
    }
}

// the ctx segment: units 0..2 from LDS rows [3][E] (split image), unit 3 from registers
__device__ __forceinline__ void pb_dot_ctx(const float4 (&wr)[1][PB_RJ], const float* __restrict__ Wl, int E, int nj,
                                           const float* __restrict__ xs, float (&acc)[4], int lane) {
#pragma unroll
    for (int j = 0; j < PB_RJ; ++j) {
        if (j < nj) {
            const int o = (lane + 64 * j) * 4;
            const float4 x = *reinterpret_cast<const float4*>(xs + o);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float4 w = *reinterpret_cast<const float4*>(Wl + (size_t)u * E + o);
                float s = acc[u];
                s = fmaf(w.x, x.x, s); s = fmaf(w.y, x.y, s); s = fmaf(w.z, x.z, s); s = fmaf(w.w, x.w, s);
                acc[u] = s;
            }
            float s = acc[3];
            s = fmaf(wr[0][j].x, x.x, s); s = fmaf(wr[0][j].y, x.y, s); s = fmaf(wr[0][j].z, x.z, s); s = fmaf(wr[0][j].w, x.w, s);
            acc[3] = s;
        }
    }
}

template <bool F32W>
__global__ __launch_bounds__(PB_NT, 1) void decode_persistent_b1_kernel(PersistParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const t2amd_dec_persist& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x, NWG = p.nwg;
    const int H = a.H, E = a.E, P = a.P, C = a.C, Ti = a.Ti, TIP = p.tip;
    const int Ka = P + E + H, Kd = H + E + H, KF = H + E;
    const int NF = P + C + 1;                       // rows of the folded projection: p1 rows, frame rows, gate row
    const int EPW = E / NWG;
    const int team = k % PB_TEAMS, NPG = NWG / PB_TEAMS, pgx = k / PB_TEAMS;    // attention dim team, position group
    if (p.timeout_ticks == 1 && k == 0) {
        // test hook (T2AMD_PB_TIMEOUT_TICKS=1): workgroup 0 behaves like a workgroup that never became resident -- it
        // records the timeout and leaves; every other workgroup finds `status` set in its first long wait and leaves too
        if (tid == 0) atomicCAS(a.status, 0, T2AMD_PERSIST_TIMEOUT);
        return;
    }

    // ---- LDS carve (all offsets multiples of 16 bytes) -------------------------------------------------------
    unsigned short* Wa_s = reinterpret_cast<unsigned short*>(smem_raw);            // [16][Ka] bf16 (F32W: [16][Ka] f32, Wd in registers)
    unsigned short* Wd_s = Wa_s + (size_t)16 * Ka;                                 // [16][Kd] bf16
    float* const Waf_s = reinterpret_cast<float*>(smem_raw);
    float* const Wdc_s = Waf_s + (size_t)16 * Ka;                                  // F32W: [4 gates][3 units][E] ctx segment of the decoder rows
    float* xp2_s = F32W ? Wdc_s + (size_t)12 * E : reinterpret_cast<float*>(Wd_s + (size_t)16 * Kd);   // [P]   split layout
    float* xctx_s = xp2_s + P;                                                     // [E]   split layout
    float* xha_s = xctx_s + E;                                                     // [H]   split layout
    float* xhd_s = xha_s + H;                                                      // [H]   split layout
    float* xp1_s = xhd_s + H;                                                      // [P + 4] linear (+ stop flag)
    float* w_s = xp1_s + P + 4;                                                    // [TiP4] attention weights of the step
    const int TiP4 = (Ti + 3) & ~3;
    float* win_s = w_s + TiP4;                                                     // [2][TIP] halo windows
    float* os_s = win_s + 2 * TIP;                                                 // [16][4] gate pre-activation partials (row, 16-lane group)
    float* red_s = os_s + 64;                                                      // [256] block-reduction scratch
    float* qpart_s = red_s + 256;                                                  // [16][16]
    float* q_s = qpart_s + 256;                                                    // [16]

    pb_u64* const G_p2 = a.mailbox;
    pb_u64* const G_ha = G_p2 + P;
    pb_u64* const G_pe = G_ha + H;                  // [PB_TEAMS][TiP4]
    pb_u64* const G_ctx = G_pe + (size_t)PB_TEAMS * TiP4;
    pb_u64* const G_hd = G_ctx + E;
    pb_u64* const G_p1 = G_hd + H;                  // [P] + stop granule at [P]

    // ---- one-time: this workgroup's LSTM rows -> LDS (row g*4+u of the image = row g*H + 4k + u of the matrix) ----
    // F32W: the decoder-LSTM rows of this wave (gate `wave`, units 0..3) live in registers: segment s of row u as float4s
    // lane + 64 j of its split image
    // (the ctx segment of units 0..2 goes to LDS behind the attention rows -- 3/4 of 16 x E x 4 bytes = 24 KB is what the
    // LDS still has -- so that the register file is not overcommitted: wd1 holds unit 3 only)
    float4 wd0[4][PB_RJ], wd1[1][PB_RJ], wd2[4][PB_RJ];        // segments h_a [H], ctx [E] (unit 3), h_d [H]
    const int njH = H >> 8, njE = E >> 8;                        // float4s per lane (H, E multiples of 256)
    if constexpr (F32W) {
        const float* __restrict__ Wa = reinterpret_cast<const float*>(a.Wa16);
        const float* __restrict__ Wd = reinterpret_cast<const float*>(a.Wd16);
        // attention-LSTM rows -> LDS image: row r = gate*4 + unit, segments [P | E | H] each in split order
        const int q4 = Ka >> 2;
        for (int i = tid; i < 16 * q4; i += PB_NT) {
            const int r = i / q4, pos = (i - r * q4) * 4;
            const long long grow = (long long)(r >> 2) * H + 4 * k + (r & 3);
            int so, sl;
            if (pos < P) { so = 0; sl = P; } else if (pos < P + E) { so = P; sl = E; } else { so = P + E; sl = H; }
            const int kk = so + pb_xinv4(pos - so, sl);
            *reinterpret_cast<float4*>(Waf_s + (size_t)r * Ka + pos) = *reinterpret_cast<const float4*>(Wa + grow * Ka + kk);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* row = Wd + ((long long)wave * H + 4 * k + u) * Kd;
#pragma unroll
            for (int j = 0; j < PB_RJ; ++j) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                wd0[u][j] = j < njH ? *reinterpret_cast<const float4*>(row + pb_xinv4((lane + 64 * j) * 4, H)) : z;
                if (u == 3) wd1[0][j] = j < njE ? *reinterpret_cast<const float4*>(row + H + pb_xinv4((lane + 64 * j) * 4, E)) : z;
                else if (j < njE) *reinterpret_cast<float4*>(Wdc_s + ((size_t)(wave * 3 + u) * E) + (lane + 64 * j) * 4) =
                    *reinterpret_cast<const float4*>(row + H + pb_xinv4((lane + 64 * j) * 4, E));
                wd2[u][j] = j < njH ? *reinterpret_cast<const float4*>(row + H + E + pb_xinv4((lane + 64 * j) * 4, H)) : z;
            }
        }
        for (int i = tid; i < P; i += PB_NT) xp2_s[i] = 0.f;
        for (int i = tid; i < 2 * TIP; i += PB_NT) win_s[i] = 0.f;
    } else {
        const int ua = Ka >> 3, ud = Kd >> 3;       // 16-byte units per row
        const uint4* __restrict__ Wa = reinterpret_cast<const uint4*>(a.Wa16);
        const uint4* __restrict__ Wd = reinterpret_cast<const uint4*>(a.Wd16);
        for (int i = tid; i < 16 * ua; i += PB_NT) {
            const int r = i / ua, c = i - r * ua;
            const long long grow = (long long)(r >> 2) * H + 4 * k + (r & 3);
            reinterpret_cast<uint4*>(Wa_s)[i] = Wa[grow * ua + c];
        }
        for (int i = tid; i < 16 * ud; i += PB_NT) {
            const int r = i / ud, c = i - r * ud;
            const long long grow = (long long)(r >> 2) * H + 4 * k + (r & 3);
            reinterpret_cast<uint4*>(Wd_s)[i] = Wd[grow * ud + c];
        }
        for (int i = tid; i < P; i += PB_NT) xp2_s[i] = 0.f;                       // p2(0) = prenet(go frame) = 0 exactly
        for (int i = tid; i < 2 * TIP; i += PB_NT) win_s[i] = 0.f;                 // w(-1) = 0, cum(-1) = 0
    }

    // ---- one-time: register-resident slices ------------------------------------------------------------------
    // folded projection rows r = k + j*NWG < NF: thread holds elements e = tid + 256*i of each
    // (F32W: the register file also holds the decoder-LSTM rows, so the per-workgroup row budgets are those of H >= 704:
    // at most 2 projection rows, 1 prenet row, 2 context channels -- checked by t2amd_decoder_persist_supported)
    constexpr int MAXFR = F32W ? 2 : PB_MAXFR, MAXP2R = F32W ? 1 : PB_MAXP2R, MAXEPW = F32W ? 2 : PB_MAXEPW;
    float wf[MAXFR][PB_MAXKPT];
    float bf_[MAXFR];
#pragma unroll
    for (int j = 0; j < MAXFR; ++j) {
        const int r = k + j * NWG;
        bf_[j] = (r < NF) ? a.bias_f[r] : 0.f;
#pragma unroll
        for (int i = 0; i < PB_MAXKPT; ++i) {
            const int e = tid + PB_NT * i;
            wf[j][i] = (r < NF && e < KF) ? a.Wf[(long long)r * KF + e] : 0.f;
        }
    }
    // prenet layer-2 rows r = k + j*NWG < P: element tid (P <= 256)
    float w2[MAXP2R];
#pragma unroll
    for (int j = 0; j < MAXP2R; ++j) {
        const int r = k + j * NWG;
        w2[j] = (r < P && tid < P) ? a.W2[(long long)r * P + tid] : 0.f;
    }
    // context: thread i < Ti keeps memory[i][EPW channels of this workgroup]
    float memr[MAXEPW];
#pragma unroll
    for (int c = 0; c < MAXEPW; ++c) memr[c] = (tid < Ti && c < EPW) ? a.memory[(long long)tid * E + k * EPW + c] : 0.f;
    // attention slice of this workgroup: dims team*16 .. +15, positions pgx + NPG*pl (pl < 8).  Thread (td = tid & 15,
    // slot = tid >> 4) keeps H/16 elements of W_q row team*16 + td (part `slot` of the row) for the q phase, and for the
    // energy phase -- where slot = 2*pl + c -- the 31 taps of window channel c of U row td, v[td] and pm[position][td]
    float wq[PB_MAXQ], ureg[T2AMD_LOC_KERNEL], pm1 = 0.f, vd;
    const int td = tid & 15, slot = tid >> 4, HQ = H >> 4;
    const int epl = slot >> 1, ech = slot & 1, epos = pgx + NPG * epl;
    {
        const int drow = team * PB_TDIM + td;
#pragma unroll
        for (int j = 0; j < PB_MAXQ; ++j) wq[j] = (j < HQ) ? a.Wq[(long long)drow * H + slot * HQ + j] : 0.f;
#pragma unroll
        for (int j = 0; j < T2AMD_LOC_KERNEL; ++j) ureg[j] = a.U[(long long)drow * T2AMD_LOC_TAPS + ech * T2AMD_LOC_KERNEL + j];
        if (epos < Ti) pm1 = a.pm[(long long)epos * T2AMD_ATT_DIM + drow];
        vd = a.v[drow];
    }
    // cell state of unit 4k + tid (threads 0..3), biases of its four gates
    // (publishing lanes: ct = tid - PB_COMM owns unit 4k + ct, projection row k + ct*NWG, prenet row k + ct*NWG ...)
    const int ct = tid - PB_COMM;
    float c_a = 0.f, c_d = 0.f, ba[4] = {0.f, 0.f, 0.f, 0.f}, bd[4] = {0.f, 0.f, 0.f, 0.f};
    if (ct >= 0 && ct < 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ba[g] = a.bias_a[g * H + 4 * k + ct];
            bd[g] = a.bias_d[g * H + 4 * k + ct];
        }
    }
    float accA[4] = {0.f, 0.f, 0.f, 0.f}, accD[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* WaW = Wa_s + (size_t)wave * 4 * Ka;      // this wave's gate: rows wave*4 .. wave*4+3
    const unsigned short* WdW = Wd_s + (size_t)wave * 4 * Kd;
    const float* WaF = Waf_s + (size_t)wave * 4 * Ka;
    // the two dot products, by operand form
#define PB_DOT_A(SEG_OFF, SEG_LEN, XS)                                                        \
    do {                                                                                      \
        if constexpr (F32W) pb_dot_seg_f32(WaF, Ka, SEG_OFF, SEG_LEN, XS, accA, lane);        \
        else pb_dot_seg(WaW, Ka, SEG_OFF, SEG_LEN, XS, accA, lane);                           \
    } while (0)
#define PB_DOT_D(SEGI, SEG_OFF, SEG_LEN, XS)                                                  \
    do {                                                                                      \
        if constexpr (F32W) {                                                                 \
            if (SEGI == 0) pb_dot_reg(wd0, njH, XS, accD, lane);                              \
            else if (SEGI == 1) pb_dot_ctx(wd1, Wdc_s + (size_t)wave * 3 * E, E, njE, XS, accD, lane); \
            else pb_dot_reg(wd2, njH, XS, accD, lane);                                        \
        } else pb_dot_seg(WdW, Kd, SEG_OFF, SEG_LEN, XS, accD, lane);                         \
    } while (0)
    const float thr = a.gate_threshold;
    float* const trace = a.trace;
    const int TRW = H + E + H + P + P;
    __syncthreads();

    // optional phase clock (a.timing != NULL): thread 0 of the first (a team) and of the last workgroup accumulate the
    // 100 MHz wall clock between phase boundaries: slot 16*w + phase, w = 0 first / 1 last workgroup
    unsigned long long* const tim = (a.timing && tid == 0 && (k == 0 || k == NWG - 1)) ? a.timing + (k == 0 ? 0 : 16) : nullptr;
    unsigned long long tprev = tim ? wall_clock64() : 0ull, tacc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) tacc[i] = 0ull;
#define PB_T(slot)                                                  \
    do {                                                            \
        if (tim) {                                                  \
            const unsigned long long now_ = wall_clock64();         \
            tacc[slot] += now_ - tprev;     /* registers: a global read-modify-write per stamp would cost ~1 us each */ \
            tprev = now_;                                           \
        }                                                           \
    } while (0)

#define PB_DELAY(i) do { for (int d_ = 0; d_ < p.delay[i]; ++d_) __builtin_amdgcn_s_sleep(1); } while (0)
    int t = 0;
    for (; t < a.max_steps; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        bool fail = false;
        unsigned char keep1 = 1, keep2 = 1;
        // (1) p2(t): the prenet output for this step (published with tag t by step t-1; zeros at t = 0)
        if (t > 0) PB_DELAY(0);
        if (t > 0) fail = pb_sweep<true, 1, true>(G_p2, P, (unsigned)t, xp2_s, P, a.status, tid, p.poll_sleep, p.timeout_ticks);
        if (__syncthreads_or(fail)) break;
        PB_T(0);

        // (2) attention LSTM: accA already holds the ctx(t-1) and h_a(t-1) parts
        PB_DOT_A(0, P, xp2_s);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = row16_sum(accA[u]);                       // 16-lane partials; the cell thread adds the four
            if ((lane & 15) == 0) os_s[(wave * 4 + u) * 4 + (lane >> 4)] = s;
            accA[u] = 0.f;
        }
        __syncthreads();
        if (ct >= 0 && ct < 4) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q4 = *reinterpret_cast<const float4*>(&os_s[(g * 4 + ct) * 4]);
                pre[g] = ((q4.x + q4.y) + (q4.z + q4.w)) + ba[g];
            }
            const float gi = t2_sigmoid_fast(pre[0]), gf = t2_sigmoid_fast(pre[1]), gg = t2_tanh(pre[2]), go = t2_sigmoid_fast(pre[3]);
            c_a = gf * c_a + gi * gg;
            const float h = go * t2_tanh(c_a);
            pb_publish(G_ha + 4 * k + ct, tag, h);
            if (trace) trace[(long long)t * TRW + 4 * k + ct] = h;
        }

        PB_T(1);
        // (3) h_a(t) from every workgroup
        PB_DELAY(1);
        fail = pb_sweep<true, 4, false>(G_ha, H, tag, xha_s, H, a.status, tid, p.poll_sleep, p.timeout_ticks);
        if (__syncthreads_or(fail)) break;
        PB_T(2);
        // keep-masks of the prenet outputs this workgroup will publish for step t + 1: fetched here (first use ~8 us away;
        // issued in front of a sweep they would hold that wave's polls back: returns are in order)
        if (t + 1 < a.max_steps) {
            if (ct >= 0 && ct < MAXFR && k + ct * NWG < P) keep1 = a.keep_prenet[((long long)(t + 1) * 2 + 0) * P + k + ct * NWG];
            if (ct >= 0 && ct < MAXP2R && k + ct * NWG < P) keep2 = a.keep_prenet[((long long)(t + 1) * 2 + 1) * P + k + ct * NWG];
        }

        // (4) q for this workgroup's 16 attention dims, then the partial energy over those dims of its (up to) 8 positions
        {
            float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
            const float* xq = xha_s;
#pragma unroll
            for (int j = 0; j < PB_MAXQ; j += 8) {
                if (j < HQ) {
                    // elements slot*HQ + j .. + 7 are one 8-unit of the split layout: two float4 reads (HQ is a multiple of 8)
                    const int kk = (slot * HQ + j) >> 3;
                    const float4 xa = *reinterpret_cast<const float4*>(xq + kk * 4);
                    const float4 xb = *reinterpret_cast<const float4*>(xq + (H >> 1) + kk * 4);
                    q0 = fmaf(wq[j], xa.x, q0); q1 = fmaf(wq[j + 1], xa.y, q1); q2 = fmaf(wq[j + 2], xa.z, q2); q3 = fmaf(wq[j + 3], xa.w, q3);
                    q0 = fmaf(wq[j + 4], xb.x, q0); q1 = fmaf(wq[j + 5], xb.y, q1); q2 = fmaf(wq[j + 6], xb.z, q2); q3 = fmaf(wq[j + 7], xb.w, q3);
                }
            }
            qpart_s[slot * 16 + td] = (q0 + q1) + (q2 + q3);
            // this thread's window operands: 31 consecutive halo-window elements of channel ech, all reads in flight
            const int ic = epos < Ti ? epos : Ti - 1;
            float wv[T2AMD_LOC_KERNEL];
#pragma unroll
            for (int j = 0; j < T2AMD_LOC_KERNEL; ++j) wv[j] = win_s[ech * TIP + ic + j];
            __syncthreads();
            if (tid < 16) {
                float q = 0.f;
#pragma unroll
                for (int pp = 0; pp < 16; ++pp) q += qpart_s[pp * 16 + tid];
                q_s[tid] = q;
            }
            float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll
            for (int j = 0; j + 3 < T2AMD_LOC_KERNEL; j += 4) {
                l0 = fmaf(ureg[j], wv[j], l0); l1 = fmaf(ureg[j + 1], wv[j + 1], l1);
                l2 = fmaf(ureg[j + 2], wv[j + 2], l2); l3 = fmaf(ureg[j + 3], wv[j + 3], l3);
            }
            l0 = fmaf(ureg[28], wv[28], l0); l1 = fmaf(ureg[29], wv[29], l1); l2 = fmaf(ureg[30], wv[30], l2);
            float loc = (l0 + l1) + (l2 + l3);
            loc += __shfl_xor(loc, 16, 64);                          // the other window channel (slot ^ 1)
            __syncthreads();
            float e = vd * t2_tanh(q_s[td] + loc + pm1);
            e = row16_sum(e);                                        // over the 16 dims (lanes td = 0..15)
            if (td == 0 && ech == 0 && epos < Ti) pb_publish(G_pe + (size_t)team * TiP4 + epos, tag, e);
        }

        // (5) the h_a(t) parts of the decoder LSTM of this step and of the attention LSTM of the next one (the partial
        // energies are on their way meanwhile)
        PB_DOT_D(0, 0, H, xha_s);
        PB_DOT_A(P + E, H, xha_s);
        PB_T(3);

        // (6) energies = fixed-order sum of the 8 team partials; softmax; this workgroup's context channels
        {
            float e = -INFINITY;
            PB_DELAY(2);
            {
                const bool mine = tid < Ti;
                const pb_u64* gq = G_pe + (mine ? tid : Ti - 1);
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                pb_u64 xa[PB_TEAMS];
#define PB_POLL8(X) _Pragma("unroll") for (int q = 0; q < PB_TEAMS; ++q) X[q] = __hip_atomic_load(gq + (size_t)q * TiP4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define PB_READY8(X, OK) OK = true; _Pragma("unroll") for (int q = 0; q < PB_TEAMS; ++q) OK = OK && ((unsigned)(X[q] >> 32) == tag);
                PB_POLL8(xa)
                for (;;) {
                    bool ok;
                    PB_READY8(xa, ok)
                    if (__all(ok)) break;
                    if ((++spins & 31u) == 0 && pb_give_up(t0, a.status, p.timeout_ticks)) { fail = true; break; }
                    for (int d_ = 0; d_ < p.poll_sleep; ++d_) __builtin_amdgcn_s_sleep(1);
                    PB_POLL8(xa)
                }
#undef PB_POLL8
#undef PB_READY8
                if (mine && !fail) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < PB_TEAMS; ++q) s += __uint_as_float((unsigned)xa[q]);       // fixed order
                    e = s;
                }
            }
            float m = row16_max(e);
            if ((lane & 15) == 0) red_s[tid >> 4] = m;
            if (__syncthreads_or(fail)) break;
            PB_T(4);
            {
                const float4 m0 = *reinterpret_cast<const float4*>(&red_s[0]), m1 = *reinterpret_cast<const float4*>(&red_s[4]);
                const float4 m2 = *reinterpret_cast<const float4*>(&red_s[8]), m3 = *reinterpret_cast<const float4*>(&red_s[12]);
                m = fmaxf(fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w))),
                          fmaxf(fmaxf(fmaxf(m2.x, m2.y), fmaxf(m2.z, m2.w)), fmaxf(fmaxf(m3.x, m3.y), fmaxf(m3.z, m3.w))));
            }
            const float ex = (tid < Ti) ? expf(e - m) : 0.f;
            const float ls = row16_sum(ex);
            if ((lane & 15) == 0) red_s[16 + (tid >> 4)] = ls;
            // context partials: 16-lane sums of ex[i] * memory[i][c] (normalised below)
#pragma unroll
            for (int c = 0; c < MAXEPW; ++c) {
                if (c < EPW) {
                    const float s = row16_sum(ex * memr[c]);
                    if ((lane & 15) == 0) red_s[32 + c * 16 + (tid >> 4)] = s;
                }
            }
            __syncthreads();
            float gsum;
            {
                const float4 s0 = *reinterpret_cast<const float4*>(&red_s[16]), s1 = *reinterpret_cast<const float4*>(&red_s[20]);
                const float4 s2 = *reinterpret_cast<const float4*>(&red_s[24]), s3 = *reinterpret_cast<const float4*>(&red_s[28]);
                gsum = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                       (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
            }
            const float inv = 1.0f / gsum;
            const float w = ex * inv;
            if (tid < Ti) {
                win_s[PB_HALO + tid] = w;                            // windows of the next step: w(t), cum(t) = cum(t-1) + w(t)
                win_s[TIP + PB_HALO + tid] += w;
                if (k == 0) a.ALIGN[(long long)t * Ti + tid] = w;    // alignment row of this step (reference model.py:447)
            }
            if (ct >= 0 && ct < EPW) {
                const float* r = &red_s[32 + ct * 16];
                const float4 s0 = *reinterpret_cast<const float4*>(r), s1 = *reinterpret_cast<const float4*>(r + 4);
                const float4 s2 = *reinterpret_cast<const float4*>(r + 8), s3 = *reinterpret_cast<const float4*>(r + 12);
                const float cx = ((((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                                  (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)))) * inv;
                pb_publish(G_ctx + k * EPW + ct, tag, cx);
                if (trace) trace[(long long)t * TRW + H + k * EPW + ct] = cx;
            }
        }
        PB_T(5);
        PB_T(6);
        // (7) ctx(t)
        PB_DELAY(3);
        fail = pb_sweep<true, 2, false>(G_ctx, E, tag, xctx_s, E, a.status, tid, p.poll_sleep, p.timeout_ticks);
        if (__syncthreads_or(fail)) break;
        PB_T(7);

        // (8) decoder LSTM: accD holds the h_a(t) and h_d(t-1) parts
        PB_DOT_D(1, H, E, xctx_s);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = row16_sum(accD[u]);
            if ((lane & 15) == 0) os_s[(wave * 4 + u) * 4 + (lane >> 4)] = s;
            accD[u] = 0.f;
        }
        __syncthreads();
        if (ct >= 0 && ct < 4) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q4 = *reinterpret_cast<const float4*>(&os_s[(g * 4 + ct) * 4]);
                pre[g] = ((q4.x + q4.y) + (q4.z + q4.w)) + bd[g];
            }
            const float gi = t2_sigmoid_fast(pre[0]), gf = t2_sigmoid_fast(pre[1]), gg = t2_tanh(pre[2]), go = t2_sigmoid_fast(pre[3]);
            c_d = gf * c_d + gi * gg;
            const float h = go * t2_tanh(c_d);
            pb_publish(G_hd + 4 * k + ct, tag, h);
            if (trace) trace[(long long)t * TRW + H + E + 4 * k + ct] = h;
        }
        PB_T(8);
        PB_DOT_A(P, E, xctx_s);                                       // ctx(t) part of the next attention LSTM
        PB_T(9);

        // (9) h_d(t)
        PB_DELAY(4);
        fail = pb_sweep<true, 4, false>(G_hd, H, tag, xhd_s, H, a.status, tid, p.poll_sleep, p.timeout_ticks);
        if (__syncthreads_or(fail)) break;
        PB_T(10);

        // (10) rows of [W1 Wp ; Wp ; Wg] . [h_d ; ctx]: p1(t+1) rows, frame rows, gate row (with the stop test)
        {
            float xr[PB_MAXKPT];
#pragma unroll
            for (int i = 0; i < PB_MAXKPT; ++i) {
                const int e = tid + PB_NT * i;
                xr[i] = e < H ? xhd_s[pb_xoff(e, H)] : (e < KF ? xctx_s[pb_xoff(e - H, E)] : 0.f);
            }
#pragma unroll
            for (int j = 0; j < MAXFR; ++j) {
                if (k + j * NWG < NF) {                               // workgroup-uniform: rows this workgroup owns
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < PB_MAXKPT; ++i) s = fmaf(wf[j][i], xr[i], s);
                    s = row16_sum(s);
                    if ((lane & 15) == 0) red_s[96 + j * 16 + (tid >> 4)] = s;
                }
            }
            __syncthreads();
            if (ct >= 0 && ct < MAXFR) {
                const int r = k + ct * NWG;
                if (r < NF) {
                    float y;
                    {
                        const float* rr = &red_s[96 + ct * 16];
                        const float4 s0 = *reinterpret_cast<const float4*>(rr), s1 = *reinterpret_cast<const float4*>(rr + 4);
                        const float4 s2 = *reinterpret_cast<const float4*>(rr + 8), s3 = *reinterpret_cast<const float4*>(rr + 12);
                        y = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                            (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
                    }
                    // bias of row tid lives in register bf_[tid] of every thread: select without dynamic indexing
                    float b = bf_[0];
#pragma unroll
                    for (int j = 1; j < MAXFR; ++j) b = (ct == j) ? bf_[j] : b;
                    y += b;
                    if (r < P) {
                        float v = fmaxf(y, 0.f);
                        if (t + 1 < a.max_steps) v = keep1 ? v * 2.0f : 0.f;
                        pb_publish(G_p1 + r, tag, v);
                        if (trace) trace[(long long)t * TRW + H + E + H + r] = v;
                    } else {
                        a.PG[(long long)t * (C + 1) + (r - P)] = y;
                        if (r == NF - 1) {
                            // stop test after the frame is emitted: sigmoid(gate) > threshold (strict); the stopping frame
                            // is part of the output (reference model.py:439-444)
                            const float sg = 1.0f / (1.0f + expf(-y));
                            const bool stop = (sg > thr) || (t + 1 >= a.max_steps);
                            if (stop) *a.out_length = t + 1;
                            pb_publish(G_p1 + P, tag, stop ? 1.0f : 0.f);
                        }
                    }
                }
            }
        }
        PB_T(11);
        PB_DOT_D(2, H + E, H, xhd_s);                                 // h_d(t) part of the next decoder LSTM
        PB_T(12);

        // (11) p1(t+1) and the stop flag; prenet layer 2
        PB_DELAY(5);
        {
            const bool mine = tid < P;
            const pb_u64* gx = G_p1 + (mine ? tid : P - 1);
            const pb_u64* gy = G_p1 + P;                              // the stop granule: every thread polls it (one address)
            const long long t0 = wall_clock64();
            unsigned spins = 0;
            pb_u64 xa, ya, xb, yb;
            xa = __hip_atomic_load(gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ya = __hip_atomic_load(gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                xb = __hip_atomic_load(gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                yb = __hip_atomic_load(gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(xa >> 32) == tag && (unsigned)(ya >> 32) == tag)) break;
                xa = __hip_atomic_load(gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ya = __hip_atomic_load(gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(xb >> 32) == tag && (unsigned)(yb >> 32) == tag)) { xa = xb; ya = yb; break; }
                if ((++spins & 31u) == 0 && pb_give_up(t0, a.status, p.timeout_ticks)) { fail = true; break; }
            }
            if (!fail) {
                if (mine) xp1_s[tid] = __uint_as_float((unsigned)xa);
                if (tid == 0) xp1_s[P] = __uint_as_float((unsigned)ya);
            }
        }
        if (__syncthreads_or(fail)) break;
        PB_T(13);
        if (xp1_s[P] != 0.f) { ++t; break; }                           // every workgroup reads the same flag
        {
            const float x = tid < P ? xp1_s[tid] : 0.f;
#pragma unroll
            for (int j = 0; j < MAXP2R; ++j) {
                if (k + j * NWG < P) {
                    const float s = row16_sum(w2[j] * x);
                    if ((lane & 15) == 0) red_s[192 + j * 16 + (tid >> 4)] = s;
                }
            }
            __syncthreads();
            if (ct >= 0 && ct < MAXP2R) {
                const int r = k + ct * NWG;
                if (r < P) {
                    float y;
                    {
                        const float* rr = &red_s[192 + ct * 16];
                        const float4 s0 = *reinterpret_cast<const float4*>(rr), s1 = *reinterpret_cast<const float4*>(rr + 4);
                        const float4 s2 = *reinterpret_cast<const float4*>(rr + 8), s3 = *reinterpret_cast<const float4*>(rr + 12);
                        y = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                            (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
                    }
                    y = fmaxf(y, 0.f);
                    y = keep2 ? y * 2.0f : 0.f;
                    pb_publish(G_p2 + r, tag, y);
                    if (trace) trace[(long long)t * TRW + H + E + H + P + r] = y;
                }
            }
        }
        PB_T(14);
    }
    if (tim) {
#pragma unroll
        for (int i = 0; i < 15; ++i) tim[i] = tacc[i];
    }
    if (k == 0 && tid == 0) a.steps_done[0] = t;
}

static int g_persist_lds = 0;

extern "C" long long t2amd_decoder_persist_mailbox_bytes(int Ti, int E, int H, int P) {
    const long long TiP4 = (Ti + 3) & ~3;
    return 8ll * (P + H + PB_TEAMS * TiP4 + E + H + P + 4);
}

static long long persist_lds_bytes(const t2amd_dec_persist* a, int tip) {
    const long long Ka = a->P + a->E + a->H, Kd = 2ll * a->H + a->E;
    const long long TiP4 = (a->Ti + 3) & ~3;
    // f32: the attention rows and 12 of the 16 ctx segments of the decoder rows (the rest of the decoder rows: registers)
    const long long rows = a->weights_f32 ? 4 * (16 * Ka + 12ll * a->E) : 2 * 16 * (Ka + Kd);
    return rows + 4 * (a->P + a->E + 2ll * a->H + a->P + 4 + TiP4 + 2ll * tip + 64 + 256 + 256 + 16);
}

// 0 = this geometry can run on the persistent kernel; otherwise the reason is left in t2amd_last_error()
extern "C" int t2amd_decoder_persist_supported(const t2amd_dec_persist* a) {
    T2_REQUIRE(a != nullptr, "dec_persist: null args");
    T2_REQUIRE(a->H > 0 && a->H % 64 == 0 && a->H / 4 <= 256, "dec_persist: H must be a multiple of 64 and H/4 <= 256 workgroups");
    const int nwg = a->H / 4;
    T2_REQUIRE(a->E % 8 == 0 && a->P % 8 == 0 && a->P <= PB_NT, "dec_persist: E, P multiples of 8, P <= 256");
    T2_REQUIRE(a->E % nwg == 0 && a->E / nwg <= PB_MAXEPW, "dec_persist: E must split into <= 4 channels per workgroup");
    T2_REQUIRE(nwg % PB_TEAMS == 0, "dec_persist: the workgroups must split into 8 attention teams (H a multiple of 32)");
    T2_REQUIRE(a->Ti > 0 && a->Ti <= PB_PPW * (nwg / PB_TEAMS) && a->Ti <= PB_NT, "dec_persist: Ti must be <= H/4 (<= 256)");
    T2_REQUIRE((a->H / 16) % 8 == 0, "dec_persist: H must be a multiple of 128");
    T2_REQUIRE(a->E <= 2 * PB_POLLERS && a->H <= 4 * PB_POLLERS && a->P <= PB_POLLERS, "dec_persist: E <= 512, H <= 1024 (granules per polling thread)");
    T2_REQUIRE((a->P + a->C + 1 + nwg - 1) / nwg <= PB_MAXFR, "dec_persist: too many projection rows per workgroup");
    T2_REQUIRE((a->P + nwg - 1) / nwg <= PB_MAXP2R, "dec_persist: too many prenet rows per workgroup");
    T2_REQUIRE((a->H + a->E + PB_NT - 1) / PB_NT <= PB_MAXKPT, "dec_persist: H + E too wide");
    T2_REQUIRE(a->H / 16 <= PB_MAXQ, "dec_persist: H too wide for the query slice");
    const int tip = ((a->Ti + 2 * PB_HALO + 2) + 3) & ~3;
    T2_REQUIRE(!a->weights_f32 || (a->H % 256 == 0 && a->E % 256 == 0 && a->H <= 256 * PB_RJ && a->E <= 256 * PB_RJ && a->P % 8 == 0),
               "dec_persist: f32 weights need H and E multiples of 256, <= 1024 (register-resident decoder-LSTM rows)");
    T2_REQUIRE(!a->weights_f32 || ((a->P + a->C + 1 + nwg - 1) / nwg <= 2 && (a->P + nwg - 1) / nwg <= 1 && a->E / nwg <= 2),
               "dec_persist: f32 weights leave registers for 2 projection rows, 1 prenet row and 2 context channels per workgroup");
    T2_REQUIRE(persist_lds_bytes(a, tip) <= 160 * 1024 - 1024, "dec_persist: the LSTM rows of one workgroup do not fit in 160 KB of LDS");
    return T2AMD_OK;
}

extern "C" int t2amd_decoder_infer_persistent_f32(const t2amd_dec_persist* a, void* stream) {
    T2_PROPAGATE(t2amd_decoder_persist_supported(a));
    T2_REQUIRE(a->Wa16 && a->Wd16 && a->bias_a && a->bias_d && a->Wq && a->U && a->v && a->Wf && a->bias_f && a->W2 &&
                   a->memory && a->pm && a->keep_prenet,
               "dec_persist: null weights/inputs");
    T2_REQUIRE(a->PG && a->ALIGN && a->out_length && a->status && a->steps_done && a->mailbox, "dec_persist: null outputs/state");
    T2_REQUIRE(t2_aligned16(a->Wa16) && t2_aligned16(a->Wd16) && (reinterpret_cast<uintptr_t>(a->mailbox) & 7u) == 0,
               "dec_persist: the LSTM weights must be 16-byte aligned, the mailbox 8-byte aligned");
    T2_REQUIRE(a->max_steps > 0, "dec_persist: max_steps");
    PersistParams p;
    p.a = *a;
    p.nwg = a->H / 4;
    p.tip = ((a->Ti + 2 * PB_HALO + 2) + 3) & ~3;
    {
        // defaults from the round-2 sweep (profiles/r02_d_decode_b1_delays.txt: 16.6 us per step without, 12.3 with); T2AMD_PB_DELAYS="p2,ha,pe,ctx,hd,p1,poll"
        // overrides them (tools only)
        static int cfg[7] = {16, 16, 0, 16, 8, 8, 1};
        static bool parsed = false;
        if (!parsed) {
            parsed = true;
            if (const char* e = getenv("T2AMD_PB_DELAYS")) {
                int v[7], n = sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]);
                for (int i = 0; i < n && i < 7; ++i) cfg[i] = v[i] < 0 ? 0 : (v[i] > 400 ? 400 : v[i]);
            }
        }
        for (int i = 0; i < 6; ++i) p.delay[i] = cfg[i];
        p.poll_sleep = cfg[6];
        // T2AMD_PB_TIMEOUT_TICKS (tests of the give-up path only): e.g. 1 makes the first wait that has to spin give up
        const char* te = getenv("T2AMD_PB_TIMEOUT_TICKS");
        p.timeout_ticks = te ? atoll(te) : PB_TIMEOUT_TICKS;
        if (p.timeout_ticks < 1) p.timeout_ticks = 1;
    }
    const long long lds = persist_lds_bytes(a, p.tip);
    hipStream_t s = (hipStream_t)stream;
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    if (lds > 64 * 1024 && lds > g_persist_lds) {
        // exactly what is needed: the kernel also owns a few hundred bytes of static LDS (the compiler's scratch for
        // __syncthreads_or), so asking for all 160 KB of dynamic LDS is refused
        if (hipFuncSetAttribute((const void*)decode_persistent_b1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_persistent_b1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            T2_FAIL("dec_persist: cannot raise the dynamic LDS limit");
        g_persist_lds = (int)lds;
    }
    // every polled word starts at zero (tags are step + 1, never 0); re-initialised on every call
    if (hipMemsetAsync(a->mailbox, 0, (size_t)t2amd_decoder_persist_mailbox_bytes(a->Ti, a->E, a->H, a->P), s) != hipSuccess ||
        hipMemsetAsync(a->status, 0, sizeof(int), s) != hipSuccess || hipMemsetAsync(a->out_length, 0, sizeof(int), s) != hipSuccess ||
        hipMemsetAsync(a->steps_done, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("dec_persist: memset failed");
    if (a->weights_f32) hipLaunchKernelGGL(decode_persistent_b1_kernel<true>, dim3(p.nwg), dim3(PB_NT), (size_t)lds, s, p);
    else hipLaunchKernelGGL(decode_persistent_b1_kernel<false>, dim3(p.nwg), dim3(PB_NT), (size_t)lds, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}


// =====================================================================================================================
// Encoder bi-LSTM of ONE utterance as one persistent launch (reference model.py:192-201 Encoder.inference, the nn.LSTM
// call :197-199).  The launch chain spends T dependent launches of ~6 us on it (0.6 ms of a 14 ms single-utterance
// call); here 2 x H/4 co-resident workgroups (both directions in one launch) keep their 16 rows of W_hh in REGISTERS
// (4 KB per wave at H = 256) and exchange h as {step + 1, f32} granules, double-buffered by step parity (a producer
// may run one step ahead of its slowest reader, never two: it needs every workgroup's h of the previous step first).
// Inference only: writes the outputs, not the gate / cell slabs the training backward reads.  Bounded spins; on a
// timeout *status != 0 and the caller must run t2amd_lstm_seq_fwd2_f32 instead.
// =====================================================================================================================
#define EP_NT 256
#define EP_MAXJ 4                        // H / 64 h values per lane: H <= 256
struct EncPersistParams {
    t2amd_lstm_seq d[2];
    int ndir, nwg_dir;
    pb_u64* mailbox;                     // [ndir][2][H]
    int* status;
    long long timeout_ticks;
};

__global__ __launch_bounds__(EP_NT, 1) void encoder_bilstm_persistent_kernel(EncPersistParams p) {
    __shared__ float os_s[16];
    __shared__ int fail_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = (int)blockIdx.x / p.nwg_dir, k = (int)blockIdx.x % p.nwg_dir;
    const t2amd_lstm_seq& a = p.d[dir];
    const int H = a.H, T = a.T, nj = H >> 6;
    pb_u64* const box = p.mailbox + (size_t)dir * 2 * H;
    // this wave's gate rows (gate = wave, units 4k .. 4k+3): element lane + 64 j of each, in registers
    float wr[4][EP_MAXJ];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float* row = a.Whh + ((long long)wave * H + 4 * k + u) * H;
#pragma unroll
        for (int j = 0; j < EP_MAXJ; ++j) wr[u][j] = j < nj ? row[lane + 64 * j] : 0.f;
    }
    float c = 0.f;
    if (tid == 0) fail_s = 0;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = a.reverse ? T - 1 - s : s;
        // this unit's four pre-activation addends (hoisted x . W_ih^T + biases): issued before the wait
        float gin[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < 4) {
            const float* g = a.GX + (long long)t * 4 * H + 4 * k + tid;
#pragma unroll
            for (int q = 0; q < 4; ++q) gin[q] = g[q * H];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // h(s-1) of the whole direction: lane polls granules lane + 64 j of buffer (s-1) & 1 until they carry tag s
            const pb_u64* g = box + (size_t)((s - 1) & 1) * H;
            pb_u64 x[EP_MAXJ];
            const unsigned tag = (unsigned)s;
#pragma unroll
            for (int j = 0; j < EP_MAXJ; ++j) x[j] = __hip_atomic_load(g + (j < nj ? lane + 64 * j : lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            unsigned spins = 0;
            bool bad = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < EP_MAXJ; ++j) ok = ok && (j >= nj || (unsigned)(x[j] >> 32) == tag);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int j = 0; j < EP_MAXJ; ++j) x[j] = __hip_atomic_load(g + (j < nj ? lane + 64 * j : lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((++spins & 31u) == 0 && pb_give_up(t0, p.status, p.timeout_ticks)) { bad = true; break; }
            }
            if (bad && lane == 0) fail_s = 1;
#pragma unroll
            for (int j = 0; j < EP_MAXJ; ++j) {
                const float h = j < nj ? __uint_as_float((unsigned)x[j]) : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fmaf(wr[u][j], h, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v = wave_reduce_sum(acc[u]);
            if (lane == 0) os_s[wave * 4 + u] = v;
        }
        __syncthreads();
        if (fail_s) return;
        if (tid < 4) {
            const float gi = t2_sigmoid(os_s[tid] + gin[0]), gf = t2_sigmoid(os_s[4 + tid] + gin[1]);
            const float gg = tanhf(os_s[8 + tid] + gin[2]), go = t2_sigmoid(os_s[12 + tid] + gin[3]);
            c = gf * c + gi * gg;
            const float h = go * tanhf(c);
            pb_publish(box + (size_t)(s & 1) * H + 4 * k + tid, (unsigned)s + 1u, h);
            a.out[(long long)t * a.ld_out + 4 * k + tid] = h;
        }
        __syncthreads();                                            // os_s is rewritten by the next step
    }
}

extern "C" long long t2amd_lstm_seq_persistent_mailbox_bytes(int H, int ndir) { return 8ll * ndir * 2 * H; }

// 0 = this geometry can run persistently (one utterance, H a multiple of 64 up to 256); else T2AMD_ERR_ARG with the reason
extern "C" int t2amd_lstm_seq_persistent_supported(const t2amd_lstm_seq* p) {
    T2_REQUIRE(p != nullptr, "lstm_seq_persistent: null args");
    T2_REQUIRE(p->B == 1, "lstm_seq_persistent: one utterance only (a batch would exchange B x H granules per step)");
    T2_REQUIRE(p->H % 64 == 0 && p->H >= 64 && p->H <= 64 * EP_MAXJ, "lstm_seq_persistent: H must be a multiple of 64, <= 256");
    T2_REQUIRE(p->T > 0, "lstm_seq_persistent: T");
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_seq_fwd2_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned long long* mailbox,
                                                  int* status, void* stream) {
    T2_PROPAGATE(t2amd_lstm_seq_persistent_supported(p));
    T2_REQUIRE(p->Whh && p->GX && p->out && mailbox && status, "lstm_seq_persistent: null pointer");
    if (q) {
        T2_PROPAGATE(t2amd_lstm_seq_persistent_supported(q));
        T2_REQUIRE(q->T == p->T && q->H == p->H && q->Whh && q->GX && q->out, "lstm_seq_persistent: the two directions must match");
    }
    T2_REQUIRE((reinterpret_cast<uintptr_t>(mailbox) & 7u) == 0, "lstm_seq_persistent: the mailbox must be 8-byte aligned");
    EncPersistParams e;
    e.d[0] = *p;
    e.d[1] = q ? *q : *p;
    e.ndir = q ? 2 : 1;
    e.nwg_dir = p->H / 4;
    e.mailbox = mailbox;
    e.status = status;
    const char* te = getenv("T2AMD_PB_TIMEOUT_TICKS");
    e.timeout_ticks = te ? atoll(te) : PB_TIMEOUT_TICKS;
    if (e.timeout_ticks < 1) e.timeout_ticks = 1;
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(mailbox, 0, (size_t)t2amd_lstm_seq_persistent_mailbox_bytes(p->H, e.ndir), s) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("lstm_seq_persistent: memset failed");
    hipLaunchKernelGGL(encoder_bilstm_persistent_kernel, dim3(e.ndir * e.nwg_dir), dim3(EP_NT), 0, s, e);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// The same recurrence for a BATCH (training forward at B = 64, batched inference) as one persistent launch.
// Workgroup (direction, row group of 32 utterances, unit group k of 4 hidden units) keeps its 16 gate rows of W_hh in
// registers as MFMA B-fragments for the whole sequence and owns the cell state of its 32 x 4 cells.  Per step it needs
// h(s-1) of its 32 rows -- all H units, produced by the H/4 workgroups of its own (direction, row group) -- multiplies
// [32 x H] . [H x 16] on the exact-f32 MFMA (each wave a quarter of K, partial sums through LDS) and runs the cell.
// Hand-off = the guide's R1 form (Guideline 16): h is stored write-through (sc1), every storing wave drains its stores,
// ONE lane stores the workgroup's step counter; consumers poll the H/4 counters of their group (one wave, relaxed), then
// read h with sc1 loads.  h(s) lives in the output slab itself: every row is written once, nothing is double-buffered.
// Rows of a batch never interact, so a row group only ever waits for its own H/4 workgroups.  Bounded spins as above.
// ---------------------------------------------------------------------------------------
#define EB_NT 256
#define EB_ROWS 32

struct EncBatchParams {
    t2amd_lstm_seq d[2];
    int ndir, nrg, nk;                   // directions, row groups, unit groups (H / 4)
    unsigned* flags;                     // [ndir][nrg][nk] step counters, zeroed by the call
    int* status;
    long long timeout_ticks;
    int delay;                           // s_sleep units (64 clocks) before the first poll of a step
};

__device__ __forceinline__ void eb_store_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 eb_load_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ __launch_bounds__(EB_NT) void encoder_bilstm_batch_persistent_kernel(EncBatchParams p) {
    __shared__ __attribute__((aligned(16))) float red_s[4][2][16][17];     // [k quarter][row tile][row][gate col (+1 pad)]
    __shared__ int fail_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_dir = p.nrg * p.nk;
    const int dir = (int)blockIdx.x / per_dir, rg = ((int)blockIdx.x % per_dir) / p.nk, k = (int)blockIdx.x % p.nk;
    const t2amd_lstm_seq& a = p.d[dir];
    const int H = a.H, T = a.T, B = a.B;
    unsigned* const flags = p.flags + ((size_t)dir * p.nrg + rg) * p.nk;
    const int r0 = rg * EB_ROWS;
    // B-fragments of this wave's k quarter [64 wave, 64 wave + 64): MFMA (i, e) multiplies k = 64 wave + 16 i + 4 (lane >> 4) + e;
    // gate column n = lane & 15 = 4 u + g  ->  row g H + 4 k + u of W_hh
    const int n = lane & 15, kq = lane >> 4;
    float wb[4][4];
    {
        const float* row = a.Whh + ((long long)(n & 3) * H + 4 * k + (n >> 2)) * H + 64 * wave + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = (64 * wave + 16 * i + 4 * kq + 3 < H) ? *reinterpret_cast<const f32x4*>(row + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            wb[i][0] = v[0]; wb[i][1] = v[1]; wb[i][2] = v[2]; wb[i][3] = v[3];
        }
    }
    // the cell this thread owns (threads 0..127): row r0 + (tid >> 2), unit 4 k + (tid & 3)
    const int crow = r0 + (tid >> 2), cu = 4 * k + (tid & 3);
    const bool cell = tid < 128 && crow < B;
    const int len = cell ? a.lens[crow] : 0;
    float c = 0.f;
    if (tid == 0) fail_s = 0;
    __syncthreads();
    // A-fragment rows of this lane: r0 + 16 mt + (lane & 15), clamped (padding rows are never stored)
    int arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { const int r = r0 + 16 * mt + (lane & 15); arow[mt] = r < B ? r : B - 1; }

    for (int s = 0; s < T; ++s) {
        const int t = a.reverse ? T - 1 - s : s;
        const int tp = a.reverse ? t + 1 : t - 1;
        float gin[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            const float* g = a.GX + ((long long)crow * T + t) * 4 * H + cu;
#pragma unroll
            for (int q = 0; q < 4; ++q) gin[q] = g[(long long)q * H];
        }
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (s > 0) {
            if (wave == 0) {                                      // one wave polls the group's counters (relaxed, agent scope)
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                bool bad = false;
                // a short pause first: 64 workgroups polling the two cache lines of their group's counters delay the very
                // stores they wait for (the decode kernel's lesson)
                for (int d_ = 0; d_ < p.delay; ++d_) __builtin_amdgcn_s_sleep(1);
                for (;;) {
                    bool ok = true;
                    for (int j = lane; j < p.nk; j += 64)
                        ok = ok && __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 31u) == 0 && pb_give_up(t0, p.status, p.timeout_ticks)) { bad = true; break; }
                }
                if (bad && lane == 0) fail_s = 1;
            }
            __syncthreads();
            if (fail_s) return;
            // h(s-1) of this lane's two rows, this wave's k quarter: 4 x 16 bytes per row tile, straight from L2 (sc1)
            f32x4 hv[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float* hp = a.out + ((long long)arow[mt] * T + tp) * a.ld_out + 64 * wave + 4 * kq;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hv[mt][i] = (64 * wave + 16 * i + 4 * kq + 3 < H) ? eb_load_sc1(hp + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // the wait is tied to the loaded registers: the MFMAs below cannot be scheduled in front of it
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(hv[0][0]), "+v"(hv[0][1]), "+v"(hv[0][2]), "+v"(hv[0][3]), "+v"(hv[1][0]), "+v"(hv[1][1]), "+v"(hv[1][2]),
                           "+v"(hv[1][3]) : : "memory");
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[mt][i][e], wb[i][e], acc[mt], 0, 0, 0);
        }
        // D layout of the 16 x 16 MFMA: lane -> column n, register r -> row 4 (lane >> 4) + r
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red_s[wave][mt][4 * kq + r][n] = acc[mt][r];
        __syncthreads();
        float h = 0.f, sg[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            const int lr = tid >> 2, u = tid & 3;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = ((red_s[0][lr >> 4][lr & 15][4 * u + g] + red_s[1][lr >> 4][lr & 15][4 * u + g]) +
                          (red_s[2][lr >> 4][lr & 15][4 * u + g] + red_s[3][lr >> 4][lr & 15][4 * u + g])) + gin[g];
            float gi = t2_sigmoid(pre[0]), gf = t2_sigmoid(pre[1]), gg = tanhf(pre[2]), go = t2_sigmoid(pre[3]);
            float cn = gf * c + gi * gg;
            h = go * tanhf(cn);
            if (t >= len) { gi = gf = gg = go = 0.f; cn = 0.f; h = 0.f; }     // packed-sequence semantics (model.py:180-188)
            c = cn;
            sg[0] = gi; sg[1] = gf; sg[2] = gg; sg[3] = go;
        }
        // the four units of a row sit in four neighbouring lanes: one 16-byte write-through store per row (R1 payload)
        {
            const float h1 = __shfl_down(h, 1), h2 = __shfl_down(h, 2), h3 = __shfl_down(h, 3);
            if (cell && (tid & 3) == 0)
                eb_store_sc1(a.out + ((long long)crow * T + t) * a.ld_out + 4 * k, f32x4{h, h1, h2, h3});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores (R1) ...
        __syncthreads();                                             // ... (and red_s may be rewritten)
        if (tid == 0) __hip_atomic_store(flags + k, (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cell) {                                                  // what only the backward reads goes out behind the flag
            float* g = a.GX + ((long long)crow * T + t) * 4 * H + cu;
            g[0] = sg[0]; g[(long long)H] = sg[1]; g[2ll * H] = sg[2]; g[3ll * H] = sg[3];
            a.C[((long long)t * B + crow) * H + cu] = c;
        }
    }
}

// Give-ups of the launch above since the last reset.  The training step does not read `status` back (a host sync at the
// top of every step makes the step time follow the host's enqueue speed); instead a one-thread launch behind the kernel
// turns a give-up into a NaN in the step's data -- the step goes non-finite, is skipped by the finite-norm check like an
// abandoned attention hand-off, and engine.handle_nonfinite_step() finds the reason here.
__device__ unsigned int t2_enc_batch_timeouts = 0u;
__global__ void encoder_batch_poison_kernel(const int* status, float* poison) {
    if (*status != 0) {
        atomicAdd(&t2_enc_batch_timeouts, 1u);
        *poison = __builtin_nanf("");
    }
}
// tests: the finishing launch alone, on a status word the caller sets
extern "C" int t2amd_debug_encoder_poison_(const int* status, float* poison, void* stream) {
    T2_REQUIRE(status && poison, "debug_encoder_poison: null args");
    hipLaunchKernelGGL(encoder_batch_poison_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, status, poison);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
extern "C" int t2amd_encoder_handoff_timeouts(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(t2_enc_batch_timeouts), sizeof(v)) != hipSuccess) return -1;
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(t2_enc_batch_timeouts), &z, sizeof(z));
    }
    return (int)v;
}

// ---------------------------------------------------------------------------------------
// BPTT of the same recurrence for a BATCH as one persistent launch (round 4; reference model.py:181-188
// under autograd).  The chain runs a step as two dependent launches -- the pointwise cell backward (rnn.hip) and the recurrent
// data gradient dh(s-1) = dG(s) . W_hh (64 x 4H x H, split-K 4) -- 2 T launches of 5.0 + 6.3 us.  Here workgroup (direction,
// row group of 32 utterances, unit group k of 4 hidden units) owns the same 32 x 4 cells as in the forward launch: it keeps
// the cells' dc carry in registers, writes their four gate gradients straight into the DG slab the weight-gradient GEMMs read
// -- write-through, R1 hand-off through per-workgroup step counters exactly as the forward hands h on -- and forms its four
// columns of the NEXT step's recurrent gradient, dh[b][4k + u] = sum_n dG[b][n] W_hh[n][4k + u] over all 4H gate columns of its
// 32 rows, which the H/4 workgroups of its own (direction, row group) have just produced.  W_hh^T rows of its four units live in
// registers as split-bf16 MFMA B-fragments (hi + lo), the dG rows are split on the way in, and the product is
// Ah.Bh + Ah.Bl + Al.Bh on v_mfma_f32_16x16x32_bf16 with f32 accumulation: ~2^-17 relative per product (what the engine's other
// gradient GEMMs use in the parity mode: gemm_bf16x3), each wave a quarter of K, partial sums through LDS.  Per step and
// workgroup: one poll of H/4 counters, 32 x 4H x 4 B = 128 KB of gate gradients from L2, 48 MFMAs per wave.
// ---------------------------------------------------------------------------------------
typedef __bf16 eb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned eb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void eb_split8(const f32x4& a, const f32x4& b, eb_u32x4& hi, eb_u32x4& lo) {
    hi.x = t2_cvt_pk_bf16(a[0], a[1]); hi.y = t2_cvt_pk_bf16(a[2], a[3]);
    hi.z = t2_cvt_pk_bf16(b[0], b[1]); hi.w = t2_cvt_pk_bf16(b[2], b[3]);
    lo.x = t2_cvt_pk_bf16(a[0] - __uint_as_float(hi.x << 16), a[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = t2_cvt_pk_bf16(a[2] - __uint_as_float(hi.y << 16), a[3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = t2_cvt_pk_bf16(b[0] - __uint_as_float(hi.z << 16), b[1] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = t2_cvt_pk_bf16(b[2] - __uint_as_float(hi.w << 16), b[3] - __uint_as_float(hi.w & 0xffff0000u));
}
#define EBB_KS 8          // k-steps of 32 per wave: 4 waves x 8 x 32 = 4H = 1024 gate columns (H = 256); fewer for smaller H

// FULL: H = 256, every wave runs all EBB_KS k-steps -- row tile 0 is complete while exactly 16 loads of row tile 1 are
// outstanding (a counted wait); smaller H issues fewer loads and waits for all of them (a runtime choice between the two
// waits cost 64 registers and the second workgroup per CU).
template <bool FULL>
__global__ __launch_bounds__(EB_NT) void encoder_bilstm_batch_persistent_bwd_kernel(EncBatchParams p) {
    __shared__ __attribute__((aligned(16))) float red_s[4][2][16][4];     // [k quarter][row tile][row][unit]
    __shared__ int fail_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_dir = p.nrg * p.nk;
    const int dir = (int)blockIdx.x / per_dir, rg = ((int)blockIdx.x % per_dir) / p.nk, k = (int)blockIdx.x % p.nk;
    const t2amd_lstm_seq& a = p.d[dir];
    const int H = a.H, T = a.T, B = a.B, G4 = 4 * H;
    unsigned* const flags = p.flags + ((size_t)dir * p.nrg + rg) * p.nk;
    const int r0 = rg * EB_ROWS;
    const int l15 = lane & 15, lg = lane >> 4;
    const int kq = G4 / 4;                       // gate columns per wave (256 at H = 256)
    const int nks = kq / 32;                     // k-steps of this wave (<= EBB_KS)
    // B fragments: MFMA k index 8 lg + e of k-step ks <-> gate column n = kq wave + 32 ks + 8 lg + e; column n' = l15 < 4 <-> unit
    // 4 k + l15 (the other twelve columns of the tile are zero): W_hh[n][4k + u] = WhhT[4k + u][n]
    eb_u32x4 bh[EBB_KS], bl[EBB_KS];
    {
        const float* wrow = a.WhhT + (long long)(4 * k + (l15 & 3)) * G4 + kq * wave + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < EBB_KS; ++ks) {
            f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
            if (ks < nks && l15 < 4) {
                w0 = *reinterpret_cast<const f32x4*>(wrow + 32 * ks);
                w1 = *reinterpret_cast<const f32x4*>(wrow + 32 * ks + 4);
            }
            eb_split8(w0, w1, bh[ks], bl[ks]);
        }
    }
    // the cell this thread owns (threads 0..127): row r0 + (tid >> 2), unit 4 k + (tid & 3)
    const int crow = r0 + (tid >> 2), cu = 4 * k + (tid & 3);
    const bool cell = tid < 128 && crow < B;
    const int len = cell ? a.lens[crow] : 0;
    float dc = 0.f;                               // dL/dc carried to the previous step (processing order)
    if (tid == 0) fail_s = 0;
    __syncthreads();
    int arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { const int r = r0 + 16 * mt + l15; arow[mt] = r < B ? r : B - 1; }

    for (int s = T - 1; s >= 0; --s) {            // reverse of the processing order
        const int t = a.reverse ? T - 1 - s : s;
        const int tp = a.reverse ? t + 1 : t - 1; // the step before this one in processing order
        // ---- this step's cell operands: none of them depends on another workgroup (issued ahead of the wait below) ----
        float g4[4] = {0.f, 0.f, 0.f, 0.f}, cc = 0.f, cp = 0.f, dho = 0.f;
        if (cell) {
            const float* g = a.GX + ((long long)crow * T + t) * G4 + cu;
#pragma unroll
            for (int q = 0; q < 4; ++q) g4[q] = g[(long long)q * H];
            cc = a.C[((long long)t * B + crow) * H + cu];
            if (s > 0) cp = a.C[((long long)tp * B + crow) * H + cu];
            dho = a.dout[((long long)crow * T + t) * a.ld_dout + cu];
        }
        // ---- the recurrent gradient into this step: dG of step s+1 (all 4H columns of my 32 rows) . my W_hh columns ----
        float dhrec = 0.f;
        if (s < T - 1) {
            if (wave == 0) {
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                bool bad = false;
                const unsigned target = (unsigned)(T - 1 - s);       // every workgroup of the group has finished step s+1
                for (int d_ = 0; d_ < p.delay; ++d_) __builtin_amdgcn_s_sleep(1);
                for (;;) {
                    bool ok = true;
                    for (int j = lane; j < p.nk; j += 64)
                        ok = ok && __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 31u) == 0 && pb_give_up(t0, p.status, p.timeout_ticks)) { bad = true; break; }
                }
                if (bad && lane == 0) fail_s = 1;
            }
            __syncthreads();
            if (fail_s) return;
            const int tn = a.reverse ? t - 1 : t + 1;                 // time index of processing step s+1
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            // the gate gradients of BOTH row tiles in one round trip (32 x 16 B per lane in flight: one workgroup per CU has the
            // whole register file; as two waited halves a step measured 7.75 us, profiles/r04_microbench_encoder_bwd_persistent.json)
            f32x4 x0[2][EBB_KS], x1[2][EBB_KS];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float* grow = a.DG + ((long long)arow[mt] * T + tn) * G4 + kq * wave + 8 * lg;
#pragma unroll
                for (int ks = 0; ks < EBB_KS; ++ks) {
                    x0[mt][ks] = (ks < nks) ? eb_load_sc1(grow + 32 * ks) : f32x4{0.f, 0.f, 0.f, 0.f};
                    x1[mt][ks] = (ks < nks) ? eb_load_sc1(grow + 32 * ks + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                // the waits are tied to the loaded registers (the MFMAs below cannot be scheduled in front of them); loads
                // return in order, so row tile 0 is complete while the 16 loads of row tile 1 are still outstanding
                if (mt == 0 && FULL)
                    asm volatile("s_waitcnt vmcnt(16)"
                                 : "+v"(x0[0][0]), "+v"(x0[0][1]), "+v"(x0[0][2]), "+v"(x0[0][3]), "+v"(x0[0][4]), "+v"(x0[0][5]), "+v"(x0[0][6]), "+v"(x0[0][7]),
                                   "+v"(x1[0][0]), "+v"(x1[0][1]), "+v"(x1[0][2]), "+v"(x1[0][3]), "+v"(x1[0][4]), "+v"(x1[0][5]), "+v"(x1[0][6]), "+v"(x1[0][7])
                                 : : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(x0[1][0]), "+v"(x0[1][1]), "+v"(x0[1][2]), "+v"(x0[1][3]), "+v"(x0[1][4]), "+v"(x0[1][5]), "+v"(x0[1][6]), "+v"(x0[1][7]),
                                   "+v"(x1[1][0]), "+v"(x1[1][1]), "+v"(x1[1][2]), "+v"(x1[1][3]), "+v"(x1[1][4]), "+v"(x1[1][5]), "+v"(x1[1][6]), "+v"(x1[1][7])
                                 : : "memory");
#pragma unroll
                for (int ks = 0; ks < EBB_KS; ++ks) {
                    if (ks < nks) {
                        eb_u32x4 ah, al;
                        eb_split8(x0[mt][ks], x1[mt][ks], ah, al);
#define EBB_M(A_, B_) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(eb_bf16x8, (A_)), __builtin_bit_cast(eb_bf16x8, (B_)), acc[mt], 0, 0, 0)
                        EBB_M(al, bh[ks]);
                        EBB_M(ah, bl[ks]);
                        EBB_M(ah, bh[ks]);
#undef EBB_M
                    }
                }
            }
            // D layout: lane -> column n' = l15 (unit, < 4 meaningful), register r -> row 4 lg + r
            if (l15 < 4) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red_s[wave][mt][4 * lg + r][l15] = acc[mt][r];
            }
            __syncthreads();
            if (cell) {
                const int lr = tid >> 2, u = tid & 3;
                dhrec = (red_s[0][lr >> 4][lr & 15][u] + red_s[1][lr >> 4][lr & 15][u]) +
                        (red_s[2][lr >> 4][lr & 15][u] + red_s[3][lr >> 4][lr & 15][u]);
            }
        }
        // ---- the cell backward (arithmetic of cell_bwd_finish, csrc/cell_bwd.h, without dropout) ----
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        if (cell) {
#pragma clang fp contract(off)
            if (t < len) {
                const float dh = (dho + dhrec) + 0.f;
                const float tc = tanhf(cc);
                const float d_o = dh * tc;
                const float dcc = dc + dh * g4[3] * (1.f - tc * tc);
                o0 = dcc * g4[2] * g4[0] * (1.f - g4[0]);
                o1 = dcc * cp * g4[1] * (1.f - g4[1]);
                o2 = dcc * g4[0] * (1.f - g4[2] * g4[2]);
                o3 = d_o * g4[3] * (1.f - g4[3]);
                dc = dcc * g4[1];
            } else {
                dc = 0.f;                        // packed-sequence semantics: steps behind the utterance carry nothing
            }
        }
        // the four units of a row sit in four neighbouring lanes: one 16-byte write-through store per row and gate (R1 payload)
        {
            float* dg = a.DG + ((long long)crow * T + t) * G4 + 4 * k;
            const float v[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y1 = __shfl_down(v[q], 1), y2 = __shfl_down(v[q], 2), y3 = __shfl_down(v[q], 3);
                if (cell && (tid & 3) == 0) eb_store_sc1(dg + (long long)q * H, f32x4{v[q], y1, y2, y3});
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores (R1) ...
        __syncthreads();                                             // ... (and red_s may be rewritten)
        if (tid == 0) __hip_atomic_store(flags + k, (unsigned)(T - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int t2amd_lstm_seq_bwd2_batch_persistent_supported(const t2amd_lstm_seq* p, int ndir, int cus) {
    T2_REQUIRE(p != nullptr, "lstm_seq_bwd_batch_persistent: null args");
    T2_REQUIRE(p->H % 64 == 0 && p->H >= 64 && p->H <= 256, "lstm_seq_bwd_batch_persistent: H must be a multiple of 64, <= 256");
    T2_REQUIRE(p->T > 0 && p->B > 0, "lstm_seq_bwd_batch_persistent: B, T");
    const long long wgs = (long long)ndir * ((p->B + EB_ROWS - 1) / EB_ROWS) * (p->H / 4);
    int per_cu = 2;
    if (!t2amd_validate_only_flag_()) {
        static int occ = 0;
        if (occ == 0) {
            int n = 0;
            occ = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, encoder_bilstm_batch_persistent_bwd_kernel<true>, EB_NT, 0) == hipSuccess && n > 0) ? n : -1;
        }
        if (occ > 0) per_cu = occ;
    }
    T2_REQUIRE(4 * wgs <= 3ll * per_cu * cus, "lstm_seq_bwd_batch_persistent: more workgroups than 3/4 of what can be co-resident");
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_seq_bwd2_batch_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned* flags, int* status,
                                                        float* poison, void* stream) {
    T2_REQUIRE(p && p->WhhT && p->GX && p->C && p->lens && p->dout && p->DG && flags && status, "lstm_seq_bwd_batch_persistent: null pointer");
    T2_REQUIRE(p->H % 64 == 0 && p->H >= 64 && p->H <= 256 && p->T > 0 && p->B > 0, "lstm_seq_bwd_batch_persistent: geometry");
    T2_REQUIRE((reinterpret_cast<uintptr_t>(p->DG) & 15u) == 0 && (reinterpret_cast<uintptr_t>(p->WhhT) & 15u) == 0,
               "lstm_seq_bwd_batch_persistent: DG and WhhT must be 16-byte aligned");
    if (q) T2_REQUIRE(q->T == p->T && q->H == p->H && q->B == p->B && q->WhhT && q->GX && q->C && q->lens && q->dout && q->DG &&
                          (reinterpret_cast<uintptr_t>(q->DG) & 15u) == 0 && (reinterpret_cast<uintptr_t>(q->WhhT) & 15u) == 0,
                      "lstm_seq_bwd_batch_persistent: the two directions must match");
    EncBatchParams e;
    e.d[0] = *p;
    e.d[1] = q ? *q : *p;
    e.ndir = q ? 2 : 1;
    e.nrg = (p->B + EB_ROWS - 1) / EB_ROWS;
    e.nk = p->H / 4;
    e.flags = flags;
    e.status = status;
    const char* te = getenv("T2AMD_PB_TIMEOUT_TICKS");
    e.timeout_ticks = te ? atoll(te) : PB_TIMEOUT_TICKS;
    if (e.timeout_ticks < 1) e.timeout_ticks = 1;
    const char* de = getenv("T2AMD_EBB_DELAY");
    e.delay = de ? atoi(de) : 8;          // measured: 1.21 ms at 4-8, 1.23 at 0 / 16, 1.44 at 64 (B = 64, T = 177)
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(flags, 0, (size_t)t2amd_lstm_seq_batch_persistent_flag_bytes(p->B, p->H, e.ndir), s) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("lstm_seq_bwd_batch_persistent: memset failed");
    if (4 * p->H / 4 / 32 == EBB_KS) hipLaunchKernelGGL(encoder_bilstm_batch_persistent_bwd_kernel<true>, dim3(e.ndir * e.nrg * e.nk), dim3(EB_NT), 0, s, e);
    else hipLaunchKernelGGL(encoder_bilstm_batch_persistent_bwd_kernel<false>, dim3(e.ndir * e.nrg * e.nk), dim3(EB_NT), 0, s, e);
    T2_LAUNCH_CHECK();
    if (poison) {
        hipLaunchKernelGGL(encoder_batch_poison_kernel, dim3(1), dim3(1), 0, s, status, poison);
        T2_LAUNCH_CHECK();
    }
    return T2AMD_OK;
}

extern "C" long long t2amd_lstm_seq_batch_persistent_flag_bytes(int B, int H, int ndir) {
    return 4ll * ndir * ((B + EB_ROWS - 1) / EB_ROWS) * (H / 4);
}

// 0 = this geometry can run as one persistent launch here; else T2AMD_ERR_ARG with the reason
extern "C" int t2amd_lstm_seq_batch_persistent_supported(const t2amd_lstm_seq* p, int ndir, int cus) {
    T2_REQUIRE(p != nullptr, "lstm_seq_batch_persistent: null args");
    T2_REQUIRE(p->H % 64 == 0 && p->H >= 64 && p->H <= 256, "lstm_seq_batch_persistent: H must be a multiple of 64, <= 256");
    T2_REQUIRE(p->T > 0 && p->B > 0, "lstm_seq_batch_persistent: B, T");
    const long long wgs = (long long)ndir * ((p->B + EB_ROWS - 1) / EB_ROWS) * (p->H / 4);
    // Co-residency is a property of THIS build's register allocation, so it is asked of the runtime, not assumed (ADVICE
    // r03); and the launch only takes 3/4 of what fits, so that a kernel running beside it on another stream (an RCCL
    // bucket, a side stream) does not turn into a 30 ms give-up.  Without a device (validate-only runs) the geometry alone
    // is checked against the 4 per CU the kernel is written for.
    int per_cu = 4;
    if (!t2amd_validate_only_flag_()) {
        static int occ = 0;
        if (occ == 0) {
            int n = 0;
            occ = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, encoder_bilstm_batch_persistent_kernel, EB_NT, 0) == hipSuccess && n > 0) ? n : -1;
        }
        if (occ > 0) per_cu = occ;
    }
    T2_REQUIRE(4 * wgs <= 3ll * per_cu * cus, "lstm_seq_batch_persistent: more workgroups than 3/4 of what can be co-resident");
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_seq_fwd2_batch_persistent_f32(const t2amd_lstm_seq* p, const t2amd_lstm_seq* q, unsigned* flags, int* status,
                                                        float* poison, void* stream) {
    T2_REQUIRE(p && p->Whh && p->GX && p->out && p->C && p->lens && flags && status, "lstm_seq_batch_persistent: null pointer");
    T2_REQUIRE(p->H % 64 == 0 && p->H >= 64 && p->H <= 256 && p->T > 0 && p->B > 0, "lstm_seq_batch_persistent: geometry");
    T2_REQUIRE(p->ld_out % 4 == 0 && (reinterpret_cast<uintptr_t>(p->out) & 15u) == 0, "lstm_seq_batch_persistent: out must be 16-byte aligned rows");
    if (q) T2_REQUIRE(q->T == p->T && q->H == p->H && q->B == p->B && q->Whh && q->GX && q->out && q->C && q->lens && q->ld_out % 4 == 0 &&
                          (reinterpret_cast<uintptr_t>(q->out) & 15u) == 0, "lstm_seq_batch_persistent: the two directions must match");
    EncBatchParams e;
    e.d[0] = *p;
    e.d[1] = q ? *q : *p;
    e.ndir = q ? 2 : 1;
    e.nrg = (p->B + EB_ROWS - 1) / EB_ROWS;
    e.nk = p->H / 4;
    e.flags = flags;
    e.status = status;
    const char* te = getenv("T2AMD_PB_TIMEOUT_TICKS");
    e.timeout_ticks = te ? atoll(te) : PB_TIMEOUT_TICKS;
    if (e.timeout_ticks < 1) e.timeout_ticks = 1;
    const char* de = getenv("T2AMD_EB_DELAY");
    e.delay = de ? atoi(de) : 16;
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(flags, 0, (size_t)t2amd_lstm_seq_batch_persistent_flag_bytes(p->B, p->H, e.ndir), s) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("lstm_seq_batch_persistent: memset failed");
    hipLaunchKernelGGL(encoder_bilstm_batch_persistent_kernel, dim3(e.ndir * e.nrg * e.nk), dim3(EB_NT), 0, s, e);
    T2_LAUNCH_CHECK();
    if (poison) {
        hipLaunchKernelGGL(encoder_batch_poison_kernel, dim3(1), dim3(1), 0, s, status, poison);
        T2_LAUNCH_CHECK();
    }
    return T2AMD_OK;
}

