// LSTM cell backward, pointwise part, as device functions shared by the stand-alone kernel (rnn.hip:
// lstm_pointwise_bwd_kernel) and by the attention-backward kernel, which runs the two decoder cells of a BPTT step as
// its closing phase (attention.hip, CELL form).  One definition = the same arithmetic, in the same order, in both.
//
// Given dL/dh' (dropped-out hidden) and the carried dL/dc, produce the gate pre-activation gradients and the new
// dL/dc carry.  Reference: torch.nn.LSTMCell under autograd (model.py:351-352, 366-370) and F.dropout (:353, :371).
#pragma once
#include "common.h"

// An addend's partial slabs at (row, col..col+3) are added in index order.  Up to four slabs are fetched by
// independent loads (addend_issue4) and only summed later (addend_finish4), after every other operand load of the
// kernel has been issued: a runtime-trip-count loop made every slab a separate, fully waited L2 round trip (seven
// in a row for the decoder cells).  More than four slabs fall back to a loop.
struct Slab4 { float4 v0, v1, v2, v3; };
// SC1 (the persistent backward loop, attention.hip): the slabs were written by OTHER workgroups of the SAME launch (the previous
// time step's dgrad tiles): device-scope loads (two 8-byte relaxed agent-scope loads per float4: sc1, tracked by the compiler's
// own wait counting, unlike an inline-asm load), never an L2 line that predates them.  Same values, same order of additions.
template <bool SC1>
__device__ __forceinline__ float4 cell_ld4(const float* q) {
    if constexpr (SC1) {
        typedef unsigned long long u64;
        const u64 lo = __hip_atomic_load(reinterpret_cast<const u64*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 hi = __hip_atomic_load(reinterpret_cast<const u64*>(q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                           __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    } else {
        return *reinterpret_cast<const float4*>(q);
    }
}
template <bool SC1 = false>
__device__ __forceinline__ Slab4 addend_issue4(const t2amd_addend& ad, int row, int col) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    Slab4 r = {z, z, z, z};
    if (!ad.p) return r;
    const float* q = ad.p + (long long)row * ad.ld + col;
    const int n = ad.nsplit;
    const long long st = ad.split_stride;
    r.v0 = cell_ld4<SC1>(q);
    if (n > 1) r.v1 = cell_ld4<SC1>(q + st);
    if (n > 2) r.v2 = cell_ld4<SC1>(q + 2 * st);
    if (n > 3) r.v3 = cell_ld4<SC1>(q + 3 * st);
    return r;
}
template <bool SC1 = false>
__device__ __forceinline__ float4 addend_finish4(const Slab4& r, const t2amd_addend& ad, int row, int col) {
    if (!ad.p) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = ad.nsplit;
    float4 s = make_float4(0.f + r.v0.x, 0.f + r.v0.y, 0.f + r.v0.z, 0.f + r.v0.w);
    if (n > 1) { s.x += r.v1.x; s.y += r.v1.y; s.z += r.v1.z; s.w += r.v1.w; }
    if (n > 2) { s.x += r.v2.x; s.y += r.v2.y; s.z += r.v2.z; s.w += r.v2.w; }
    if (n > 3) { s.x += r.v3.x; s.y += r.v3.y; s.z += r.v3.z; s.w += r.v3.w; }
    if (n > 4) {
        const float* q = ad.p + (long long)row * ad.ld + col;
        for (int k = 4; k < n; ++k) {
            const float4 v = cell_ld4<SC1>(q + (long long)k * ad.split_stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    return s;
}

// Operands of the cell at (row b, units j..j+3) that do not depend on the gradient of h: issued as independent loads.
struct CellOperands {
    float4 gi, gf, gg, go, c, cprev, dc_in;
    unsigned kp;
};
__device__ __forceinline__ CellOperands cell_bwd_issue(const t2amd_lstm_bwd& a, int b, int j) {
    const int H = a.H;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    CellOperands r;
    const float* g = a.gates + (long long)b * a.ld_gates + j;
    r.gi = *reinterpret_cast<const float4*>(g);
    r.gf = *reinterpret_cast<const float4*>(g + H);
    r.gg = *reinterpret_cast<const float4*>(g + 2 * H);
    r.go = *reinterpret_cast<const float4*>(g + 3 * H);
    r.c = *reinterpret_cast<const float4*>(a.c + (long long)b * a.ld_c + j);
    r.cprev = z4;      // (not `cond ? *p : z4`: a conditional of two lvalues selects between ADDRESSES and puts z4 in scratch)
    if (a.c_prev) r.cprev = *reinterpret_cast<const float4*>(a.c_prev + (long long)b * a.ld_cprev + j);
    r.dc_in = *reinterpret_cast<const float4*>(a.dc + (long long)b * a.ld_dc + j);
    r.kp = 0x01010101u;
    if (a.keep) r.kp = *reinterpret_cast<const unsigned*>(a.keep + (long long)b * a.ld_keep + j);
    return r;
}

#define T2_PK4(o) make_uint2((unsigned)t2_f32_to_bf16(o[0]) | ((unsigned)t2_f32_to_bf16(o[1]) << 16), \
                             (unsigned)t2_f32_to_bf16(o[2]) | ((unsigned)t2_f32_to_bf16(o[3]) << 16))
// Where the operand copy of gate column k (k % 4 == 0) of row b starts, in bf16 units: a plain bf16 row, or (dgates16_x3, round 6:
// the 'bf16x3' mode) the split image of t2amd_split_bf16x3_f32 -- the four hi values there, the four lo values 16 further.
__device__ __forceinline__ unsigned short* cell_d16_at(const t2amd_lstm_bwd& a, int b, int k) {
    unsigned short* const base = reinterpret_cast<unsigned short*>(a.dgates16);
    return a.dgates16_x3 ? base + (long long)b * a.ld_dgates16 * 2 + t2_x3_pos(k) : base + (long long)b * a.ld_dgates16 + k;
}
// four consecutive gate gradients -> their operand copy (bf16, or hi + lo of the split image)
template <bool SC1>
__device__ __forceinline__ void cell_d16_store(const t2amd_lstm_bwd& a, unsigned short* d, const float (&o)[4]) {
    typedef unsigned long long u64;
    const uint2 h = T2_PK4(o);
    if constexpr (SC1) __hip_atomic_store(reinterpret_cast<u64*>(d), (u64)h.x | ((u64)h.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<uint2*>(d) = h;
    if (a.dgates16_x3) {
        const float r[4] = {o[0] - __uint_as_float(h.x << 16), o[1] - __uint_as_float(h.x & 0xffff0000u),
                            o[2] - __uint_as_float(h.y << 16), o[3] - __uint_as_float(h.y & 0xffff0000u)};
        const uint2 l = T2_PK4(r);
        if constexpr (SC1) __hip_atomic_store(reinterpret_cast<u64*>(d + 16), (u64)l.x | ((u64)l.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *reinterpret_cast<uint2*>(d + 16) = l;
    }
}
// dh = (d0 + d1) + d2 in that order (the three addends of t2amd_lstm_bwd.dh, each already summed over its slabs)
// SC1: the bf16 copy of the gate gradients is what the dgrad tiles of the SAME launch read next: write-through stores.
template <bool SC1 = false>
__device__ __forceinline__ void cell_bwd_finish(const t2amd_lstm_bwd& a, const CellOperands& r, const float4& d0,
                                                const float4& d1, const float4& d2, int b, int j) {
    // no mul+add contraction in here: which products the compiler fuses depends on the surrounding kernel, and the two
    // kernels that share this function must agree bit for bit (tests/test_kernels_gpu.py, folded cells)
#pragma clang fp contract(off)
    const int H = a.H;
    float* dcp = a.dc + (long long)b * a.ld_dc + j;
    const float gi_[4] = {r.gi.x, r.gi.y, r.gi.z, r.gi.w}, gf_[4] = {r.gf.x, r.gf.y, r.gf.z, r.gf.w};
    const float gg_[4] = {r.gg.x, r.gg.y, r.gg.z, r.gg.w}, go_[4] = {r.go.x, r.go.y, r.go.z, r.go.w};
    const float c_[4] = {r.c.x, r.c.y, r.c.z, r.c.w}, cp_[4] = {r.cprev.x, r.cprev.y, r.cprev.z, r.cprev.w};
    const float dci[4] = {r.dc_in.x, r.dc_in.y, r.dc_in.z, r.dc_in.w};
    const float dh_[4] = {(d0.x + d1.x) + d2.x, (d0.y + d1.y) + d2.y, (d0.z + d1.z) + d2.z, (d0.w + d1.w) + d2.w};
    float o0[4], o1[4], o2[4], o3[4], dcn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float dh = dh_[e];
        if (a.keep) dh = ((r.kp >> (8 * e)) & 0xffu) ? dh * a.keep_scale : 0.f;
        const float tc = tanhf(c_[e]);
        const float d_o = dh * tc;
        const float dc = dci[e] + dh * go_[e] * (1.f - tc * tc);
        o0[e] = dc * gg_[e] * gi_[e] * (1.f - gi_[e]);
        o1[e] = dc * cp_[e] * gf_[e] * (1.f - gf_[e]);
        o2[e] = dc * gi_[e] * (1.f - gg_[e] * gg_[e]);
        o3[e] = d_o * go_[e] * (1.f - go_[e]);
        dcn[e] = dc * gf_[e];
    }
    if (a.dgates) {         // NULL (round 6): the operand copy below is this cell's only output -- the bf16 mode's whole-sequence slabs
        float* dg = a.dgates + (long long)b * a.ld_dgates + j;
        *reinterpret_cast<float4*>(dg) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(dg + H) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<float4*>(dg + 2 * H) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        *reinterpret_cast<float4*>(dg + 3 * H) = make_float4(o3[0], o3[1], o3[2], o3[3]);
    }
    if (a.dgates16) {       // operand copy for the dgrad GEMM's MFMA: bf16 (bf16 mode) or the split hi/lo image ('bf16x3' mode)
        cell_d16_store<SC1>(a, cell_d16_at(a, b, j), o0);
        cell_d16_store<SC1>(a, cell_d16_at(a, b, H + j), o1);
        cell_d16_store<SC1>(a, cell_d16_at(a, b, 2 * H + j), o2);
        cell_d16_store<SC1>(a, cell_d16_at(a, b, 3 * H + j), o3);
    }
    *reinterpret_cast<float4*>(dcp) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
}
#undef T2_PK4
