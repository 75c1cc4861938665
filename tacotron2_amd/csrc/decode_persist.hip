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
//     p2 -> [LSTM_a] -> h_a -> [energies, 8 team CUs] -> partial energies -> [softmax + context slice, every CU]
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

#define PB_NT 256
#define PB_TEAMS 8                       // attention teams: 16 of the 128 attention dims each
#define PB_TDIM (T2AMD_ATT_DIM / PB_TEAMS)
#define PB_MAXR 16                       // position rounds of an energy thread: Ti <= 16 * 16
#define PB_MAXFR 6                       // rows of the folded projection per workgroup
#define PB_MAXKPT 6                      // (H + E) / 256 elements of such a row per thread
#define PB_MAXP2R 2                      // prenet layer-2 rows per workgroup
#define PB_MAXEPW 4                      // context channels per workgroup
#define PB_MAXQ 64                       // H / 16 W_q elements per energy thread
#define PB_HALO 15
#define PB_TIMEOUT_TICKS 3000000ll       // 30 ms of the 100 MHz wall clock per wait

typedef unsigned long long pb_u64;

struct PersistParams {
    t2amd_dec_persist a;
    int nwg, tip;
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
__device__ __forceinline__ bool pb_give_up(long long t0, int* status) {
    const bool late = wall_clock64() - t0 > PB_TIMEOUT_TICKS;
    const bool other = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (!__any(late || other)) return false;
    if (late && !other && (threadIdx.x & 63) == 0) atomicCAS(status, 0, T2AMD_PERSIST_TIMEOUT);
    return true;
}

// Every thread polls its own granules (tid, tid + 256, ...) until all of a wave's carry `tag`; values go to LDS.
// Returns true on failure (timeout or another workgroup already failed).  Wave-uniform control flow.
template <bool SPLIT>
__device__ __forceinline__ bool pb_sweep(const pb_u64* __restrict__ g, int n, unsigned tag, float* __restrict__ dst, int len,
                                         int* status, int tid) {
    bool fail = false;
    for (int base = 0; base < n; base += PB_NT) {
        const int i = base + tid;
        const bool mine = i < n;
        pb_u64 x = 0;
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            x = mine ? __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((pb_u64)tag << 32);
            if (__all((unsigned)(x >> 32) == tag)) break;
            if ((++spins & 63u) == 0 && pb_give_up(t0, status)) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (fail) break;
        if (mine) dst[SPLIT ? pb_xoff(i, len) : i] = __uint_as_float((unsigned)x);
    }
    return fail;
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

__global__ __launch_bounds__(PB_NT, 1) void decode_persistent_b1_kernel(PersistParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const t2amd_dec_persist& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x, NWG = p.nwg;
    const int H = a.H, E = a.E, P = a.P, C = a.C, Ti = a.Ti, TIP = p.tip;
    const int Ka = P + E + H, Kd = H + E + H, KF = H + E;
    const int NF = P + C + 1;                       // rows of the folded projection: p1 rows, frame rows, gate row
    const int EPW = E / NWG;
    const bool team = k < PB_TEAMS;

    // ---- LDS carve (all offsets multiples of 16 bytes) -------------------------------------------------------
    unsigned short* Wa_s = reinterpret_cast<unsigned short*>(smem_raw);            // [16][Ka] bf16
    unsigned short* Wd_s = Wa_s + (size_t)16 * Ka;                                 // [16][Kd] bf16
    float* xp2_s = reinterpret_cast<float*>(Wd_s + (size_t)16 * Kd);               // [P]   split layout
    float* xctx_s = xp2_s + P;                                                     // [E]   split layout
    float* xha_s = xctx_s + E;                                                     // [H]   split layout
    float* xhd_s = xha_s + H;                                                      // [H]   split layout
    float* xp1_s = xhd_s + H;                                                      // [P + 4] linear (+ stop flag)
    float* w_s = xp1_s + P + 4;                                                    // [TiP4] attention weights of the step
    const int TiP4 = (Ti + 3) & ~3;
    float* win_s = w_s + TiP4;                                                     // [2][TIP] halo windows (teams)
    float* os_s = win_s + 2 * TIP;                                                 // [16] gate pre-activations
    float* red_s = os_s + 16;                                                      // [64] block-reduction scratch
    float* qpart_s = red_s + 64;                                                   // [16][16] (teams)
    float* q_s = qpart_s + 256;                                                    // [16] (teams)

    pb_u64* const G_p2 = a.mailbox;
    pb_u64* const G_ha = G_p2 + P;
    pb_u64* const G_pe = G_ha + H;                  // [PB_TEAMS][TiP4]
    pb_u64* const G_ctx = G_pe + (size_t)PB_TEAMS * TiP4;
    pb_u64* const G_hd = G_ctx + E;
    pb_u64* const G_p1 = G_hd + H;                  // [P] + stop granule at [P]

    // ---- one-time: this workgroup's LSTM rows -> LDS (row g*4+u of the image = row g*H + 4k + u of the matrix) ----
    {
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
    float wf[PB_MAXFR][PB_MAXKPT];
    float bf_[PB_MAXFR];
#pragma unroll
    for (int j = 0; j < PB_MAXFR; ++j) {
        const int r = k + j * NWG;
        bf_[j] = (r < NF) ? a.bias_f[r] : 0.f;
#pragma unroll
        for (int i = 0; i < PB_MAXKPT; ++i) {
            const int e = tid + PB_NT * i;
            wf[j][i] = (r < NF && e < KF) ? a.Wf[(long long)r * KF + e] : 0.f;
        }
    }
    // prenet layer-2 rows r = k + j*NWG < P: element tid (P <= 256)
    float w2[PB_MAXP2R];
#pragma unroll
    for (int j = 0; j < PB_MAXP2R; ++j) {
        const int r = k + j * NWG;
        w2[j] = (r < P && tid < P) ? a.W2[(long long)r * P + tid] : 0.f;
    }
    // context: thread i < Ti keeps memory[i][EPW channels of this workgroup]
    float memr[PB_MAXEPW];
#pragma unroll
    for (int c = 0; c < PB_MAXEPW; ++c) memr[c] = (tid < Ti && c < EPW) ? a.memory[(long long)tid * E + k * EPW + c] : 0.f;
    // teams: W_q slice (thread (d = tid & 15, part = tid >> 4) keeps H/16 elements of row k*16 + d), the 62 taps of
    // U row d, v[d], processed-memory entries of its positions
    float wq[PB_MAXQ], ureg[T2AMD_LOC_TAPS], pmr[PB_MAXR], vd = 0.f;
    const int td = tid & 15, tpg = tid >> 4, HQ = H >> 4;
    if (team) {
        const int drow = k * PB_TDIM + td;
#pragma unroll
        for (int j = 0; j < PB_MAXQ; ++j) wq[j] = (j < HQ) ? a.Wq[(long long)drow * H + tpg * HQ + j] : 0.f;
#pragma unroll
        for (int j = 0; j < T2AMD_LOC_TAPS; ++j) ureg[j] = a.U[(long long)drow * T2AMD_LOC_TAPS + j];
#pragma unroll
        for (int r = 0; r < PB_MAXR; ++r) {
            const int i = tpg + 16 * r;
            pmr[r] = (i < Ti) ? a.pm[(long long)i * T2AMD_ATT_DIM + drow] : 0.f;
        }
        vd = a.v[drow];
    }
    // cell state of unit 4k + tid (threads 0..3), biases of its four gates
    float c_a = 0.f, c_d = 0.f, ba[4] = {0.f, 0.f, 0.f, 0.f}, bd[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ba[g] = a.bias_a[g * H + 4 * k + tid];
            bd[g] = a.bias_d[g * H + 4 * k + tid];
        }
    }
    float accA[4] = {0.f, 0.f, 0.f, 0.f}, accD[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* WaW = Wa_s + (size_t)wave * 4 * Ka;      // this wave's gate: rows wave*4 .. wave*4+3
    const unsigned short* WdW = Wd_s + (size_t)wave * 4 * Kd;
    const float thr = a.gate_threshold;
    float* const trace = a.trace;
    const int TRW = H + E + H + P + P;
    __syncthreads();

    int t = 0;
    for (; t < a.max_steps; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        bool fail = false;
        // (1) p2(t): the prenet output for this step (published with tag t by step t-1; zeros at t = 0)
        if (t > 0) fail = pb_sweep<true>(G_p2, P, (unsigned)t, xp2_s, P, a.status, tid);
        if (__syncthreads_or(fail)) break;

        // (2) attention LSTM: accA already holds the ctx(t-1) and h_a(t-1) parts
        pb_dot_seg(WaW, Ka, 0, P, xp2_s, accA, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_reduce_sum(accA[u]);
            if (lane == 0) os_s[wave * 4 + u] = s;
            accA[u] = 0.f;
        }
        __syncthreads();
        if (tid < 4) {
            const float gi = t2_sigmoid(os_s[0 * 4 + tid] + ba[0]), gf = t2_sigmoid(os_s[1 * 4 + tid] + ba[1]);
            const float gg = tanhf(os_s[2 * 4 + tid] + ba[2]), go = t2_sigmoid(os_s[3 * 4 + tid] + ba[3]);
            c_a = gf * c_a + gi * gg;
            const float h = go * tanhf(c_a);
            pb_publish(G_ha + 4 * k + tid, tag, h);
            if (trace) trace[(long long)t * TRW + 4 * k + tid] = h;
        }

        // (3) h_a(t) from every workgroup
        fail = pb_sweep<true>(G_ha, H, tag, xha_s, H, a.status, tid);
        if (__syncthreads_or(fail)) break;

        // (4) teams: q for 16 attention dims, partial energies of all positions over those dims
        if (team) {
            float qp = 0.f;
#pragma unroll
            for (int j = 0; j < PB_MAXQ; ++j)
                if (j < HQ) qp = fmaf(wq[j], xha_s[pb_xoff(tpg * HQ + j, H)], qp);
            qpart_s[tpg * 16 + td] = qp;
            __syncthreads();
            if (tid < 16) {
                float q = 0.f;
#pragma unroll
                for (int pp = 0; pp < 16; ++pp) q += qpart_s[pp * 16 + tid];
                q_s[tid] = q;
            }
            __syncthreads();
            const float qd = q_s[td];
#pragma unroll
            for (int r = 0; r < PB_MAXR; ++r) {
                const int i = tpg + 16 * r;
                if (16 * r < Ti) {                                   // wave-uniform round guard
                    const int ic = i < Ti ? i : Ti - 1;
                    float loc = 0.f;
#pragma unroll
                    for (int j = 0; j < T2AMD_LOC_KERNEL; ++j) loc = fmaf(ureg[j], win_s[ic + j], loc);
#pragma unroll
                    for (int j = 0; j < T2AMD_LOC_KERNEL; ++j) loc = fmaf(ureg[T2AMD_LOC_KERNEL + j], win_s[TIP + ic + j], loc);
                    float e = vd * t2_tanh(qd + loc + pmr[r]);
                    e = row16_sum(e);                                // over the team's 16 dims (lanes td = 0..15)
                    if (td == 0 && i < Ti) pb_publish(G_pe + (size_t)k * TiP4 + i, tag, e);
                }
            }
        }

        // (5) the h_a(t) parts of the decoder LSTM of this step and of the attention LSTM of the next one
        pb_dot_seg(WdW, Kd, 0, H, xha_s, accD, lane);
        pb_dot_seg(WaW, Ka, P + E, H, xha_s, accA, lane);

        // (6) energies = fixed-order sum of the 8 team partials; softmax; this workgroup's context channels
        {
            float e = -INFINITY;
            {
                const bool mine = tid < Ti;
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < PB_TEAMS; ++q) {
                        const pb_u64 x = mine ? __hip_atomic_load(G_pe + (size_t)q * TiP4 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : ((pb_u64)tag << 32);
                        ok = ok && ((unsigned)(x >> 32) == tag);
                        s += __uint_as_float((unsigned)x);
                    }
                    if (__all(ok)) { if (mine) e = s; break; }
                    if ((++spins & 63u) == 0 && pb_give_up(t0, a.status)) { fail = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            float m = wave_reduce_max(e);
            if (lane == 0) red_s[wave] = m;
            if (__syncthreads_or(fail)) break;
            m = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
            const float ex = (tid < Ti) ? expf(e - m) : 0.f;
            const float ls = wave_reduce_sum(ex);
            if (lane == 0) red_s[4 + wave] = ls;
            __syncthreads();
            const float inv = 1.0f / (((red_s[4] + red_s[5]) + red_s[6]) + red_s[7]);
            const float w = ex * inv;
            // context partials: wave sums of w[i] * memory[i][c]
#pragma unroll
            for (int c = 0; c < PB_MAXEPW; ++c) {
                const float s = wave_reduce_sum(w * memr[c]);
                if (lane == 0) red_s[8 + c * 4 + wave] = s;
            }
            if (tid < Ti) {
                if (k == 0) a.ALIGN[(long long)t * Ti + tid] = w;
                if (team) {                                          // windows of the next step: w(t), cum(t) = cum(t-1) + w(t)
                    win_s[PB_HALO + tid] = w;
                    win_s[TIP + PB_HALO + tid] += w;
                }
            }
            __syncthreads();
            if (tid < EPW) {
                const float cx = ((red_s[8 + tid * 4] + red_s[8 + tid * 4 + 1]) + red_s[8 + tid * 4 + 2]) + red_s[8 + tid * 4 + 3];
                pb_publish(G_ctx + k * EPW + tid, tag, cx);
                if (trace) trace[(long long)t * TRW + H + k * EPW + tid] = cx;
            }
        }

        // (7) ctx(t)
        fail = pb_sweep<true>(G_ctx, E, tag, xctx_s, E, a.status, tid);
        if (__syncthreads_or(fail)) break;

        // (8) decoder LSTM: accD holds the h_a(t) and h_d(t-1) parts
        pb_dot_seg(WdW, Kd, H, E, xctx_s, accD, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_reduce_sum(accD[u]);
            if (lane == 0) os_s[wave * 4 + u] = s;
            accD[u] = 0.f;
        }
        __syncthreads();
        if (tid < 4) {
            const float gi = t2_sigmoid(os_s[0 * 4 + tid] + bd[0]), gf = t2_sigmoid(os_s[1 * 4 + tid] + bd[1]);
            const float gg = tanhf(os_s[2 * 4 + tid] + bd[2]), go = t2_sigmoid(os_s[3 * 4 + tid] + bd[3]);
            c_d = gf * c_d + gi * gg;
            const float h = go * tanhf(c_d);
            pb_publish(G_hd + 4 * k + tid, tag, h);
            if (trace) trace[(long long)t * TRW + H + E + 4 * k + tid] = h;
        }
        pb_dot_seg(WaW, Ka, P, E, xctx_s, accA, lane);                // ctx(t) part of the next attention LSTM

        // (9) h_d(t)
        fail = pb_sweep<true>(G_hd, H, tag, xhd_s, H, a.status, tid);
        if (__syncthreads_or(fail)) break;

        // (10) rows of [W1 Wp ; Wp ; Wg] . [h_d ; ctx]: p1(t+1) rows, frame rows, gate row (with the stop test)
        {
            float xr[PB_MAXKPT];
#pragma unroll
            for (int i = 0; i < PB_MAXKPT; ++i) {
                const int e = tid + PB_NT * i;
                xr[i] = e < H ? xhd_s[pb_xoff(e, H)] : (e < KF ? xctx_s[pb_xoff(e - H, E)] : 0.f);
            }
#pragma unroll
            for (int j = 0; j < PB_MAXFR; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < PB_MAXKPT; ++i) s = fmaf(wf[j][i], xr[i], s);
                s = wave_reduce_sum(s);
                if (lane == 0) red_s[32 + j * 4 + wave] = s;
            }
            __syncthreads();
            if (tid < PB_MAXFR) {
                const int r = k + tid * NWG;
                if (r < NF) {
                    float y = ((red_s[32 + tid * 4] + red_s[32 + tid * 4 + 1]) + red_s[32 + tid * 4 + 2]) + red_s[32 + tid * 4 + 3];
                    // bias of row tid lives in register bf_[tid] of every thread: select without dynamic indexing
                    float b = bf_[0];
#pragma unroll
                    for (int j = 1; j < PB_MAXFR; ++j) b = (tid == j) ? bf_[j] : b;
                    y += b;
                    if (r < P) {
                        float v = fmaxf(y, 0.f);
                        if (t + 1 < a.max_steps) v = a.keep_prenet[((long long)(t + 1) * 2 + 0) * P + r] ? v * 2.0f : 0.f;
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
        pb_dot_seg(WdW, Kd, H + E, H, xhd_s, accD, lane);             // h_d(t) part of the next decoder LSTM

        // (11) p1(t+1) and the stop flag; prenet layer 2
        {
            const bool mine = tid < P;
            const long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                const pb_u64 x = mine ? __hip_atomic_load(G_p1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((pb_u64)tag << 32);
                const pb_u64 y = tid == 0 ? __hip_atomic_load(G_p1 + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((pb_u64)tag << 32);
                if (__all((unsigned)(x >> 32) == tag && (unsigned)(y >> 32) == tag)) {
                    if (mine) xp1_s[tid] = __uint_as_float((unsigned)x);
                    if (tid == 0) xp1_s[P] = __uint_as_float((unsigned)y);
                    break;
                }
                if ((++spins & 63u) == 0 && pb_give_up(t0, a.status)) { fail = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (__syncthreads_or(fail)) break;
        if (xp1_s[P] != 0.f) { ++t; break; }                           // every workgroup reads the same flag
        {
            const float x = tid < P ? xp1_s[tid] : 0.f;
#pragma unroll
            for (int j = 0; j < PB_MAXP2R; ++j) {
                const float s = wave_reduce_sum(w2[j] * x);
                if (lane == 0) red_s[56 + j * 4 + wave] = s;
            }
            __syncthreads();
            if (tid < PB_MAXP2R) {
                const int r = k + tid * NWG;
                if (r < P) {
                    float y = ((red_s[56 + tid * 4] + red_s[56 + tid * 4 + 1]) + red_s[56 + tid * 4 + 2]) + red_s[56 + tid * 4 + 3];
                    y = fmaxf(y, 0.f);
                    y = a.keep_prenet[((long long)(t + 1) * 2 + 1) * P + r] ? y * 2.0f : 0.f;
                    pb_publish(G_p2 + r, tag, y);
                    if (trace) trace[(long long)t * TRW + H + E + H + P + r] = y;
                }
            }
        }
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
    return 2 * 16 * (Ka + Kd) + 4 * (a->P + a->E + 2ll * a->H + a->P + 4 + TiP4 + 2ll * tip + 16 + 64 + 256 + 16);
}

// 0 = this geometry can run on the persistent kernel; otherwise the reason is left in t2amd_last_error()
extern "C" int t2amd_decoder_persist_supported(const t2amd_dec_persist* a) {
    T2_REQUIRE(a != nullptr, "dec_persist: null args");
    T2_REQUIRE(a->H > 0 && a->H % 64 == 0 && a->H / 4 <= 256, "dec_persist: H must be a multiple of 64 and H/4 <= 256 workgroups");
    const int nwg = a->H / 4;
    T2_REQUIRE(a->E % 8 == 0 && a->P % 8 == 0 && a->P <= PB_NT, "dec_persist: E, P multiples of 8, P <= 256");
    T2_REQUIRE(a->E % nwg == 0 && a->E / nwg <= PB_MAXEPW, "dec_persist: E must split into <= 4 channels per workgroup");
    T2_REQUIRE(nwg >= PB_TEAMS, "dec_persist: fewer workgroups than attention teams");
    T2_REQUIRE(a->Ti > 0 && a->Ti <= 16 * PB_MAXR, "dec_persist: Ti must be <= 256");
    T2_REQUIRE((a->P + a->C + 1 + nwg - 1) / nwg <= PB_MAXFR, "dec_persist: too many projection rows per workgroup");
    T2_REQUIRE((a->P + nwg - 1) / nwg <= PB_MAXP2R, "dec_persist: too many prenet rows per workgroup");
    T2_REQUIRE((a->H + a->E + PB_NT - 1) / PB_NT <= PB_MAXKPT, "dec_persist: H + E too wide");
    T2_REQUIRE(a->H / 16 <= PB_MAXQ, "dec_persist: H too wide for the query slice");
    const int tip = ((a->Ti + 2 * PB_HALO + 2) + 3) & ~3;
    T2_REQUIRE(persist_lds_bytes(a, tip) <= 160 * 1024, "dec_persist: the LSTM rows of one workgroup do not fit in 160 KB of LDS");
    return T2AMD_OK;
}

extern "C" int t2amd_decoder_infer_persistent_f32(const t2amd_dec_persist* a, void* stream) {
    T2_PROPAGATE(t2amd_decoder_persist_supported(a));
    T2_REQUIRE(a->Wa16 && a->Wd16 && a->bias_a && a->bias_d && a->Wq && a->U && a->v && a->Wf && a->bias_f && a->W2 &&
                   a->memory && a->pm && a->keep_prenet,
               "dec_persist: null weights/inputs");
    T2_REQUIRE(a->PG && a->ALIGN && a->out_length && a->status && a->steps_done && a->mailbox, "dec_persist: null outputs/state");
    T2_REQUIRE(t2_aligned16(a->Wa16) && t2_aligned16(a->Wd16) && (reinterpret_cast<uintptr_t>(a->mailbox) & 7u) == 0,
               "dec_persist: bf16 weights must be 16-byte aligned, the mailbox 8-byte aligned");
    T2_REQUIRE(a->max_steps > 0, "dec_persist: max_steps");
    PersistParams p;
    p.a = *a;
    p.nwg = a->H / 4;
    p.tip = ((a->Ti + 2 * PB_HALO + 2) + 3) & ~3;
    const long long lds = persist_lds_bytes(a, p.tip);
    hipStream_t s = (hipStream_t)stream;
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    if (lds > 64 * 1024 && lds > g_persist_lds) {
        if (hipFuncSetAttribute((const void*)decode_persistent_b1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            T2_FAIL("dec_persist: cannot raise the dynamic LDS limit");
        g_persist_lds = (int)lds;
    }
    // every polled word starts at zero (tags are step + 1, never 0); re-initialised on every call
    if (hipMemsetAsync(a->mailbox, 0, (size_t)t2amd_decoder_persist_mailbox_bytes(a->Ti, a->E, a->H, a->P), s) != hipSuccess ||
        hipMemsetAsync(a->status, 0, sizeof(int), s) != hipSuccess || hipMemsetAsync(a->out_length, 0, sizeof(int), s) != hipSuccess ||
        hipMemsetAsync(a->steps_done, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("dec_persist: memset failed");
    hipLaunchKernelGGL(decode_persistent_b1_kernel, dim3(p.nwg), dim3(PB_NT), (size_t)lds, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
