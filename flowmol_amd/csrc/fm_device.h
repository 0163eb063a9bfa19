// Device-side building blocks for the FlowMol3 sampling hot path on gfx950 (MI355X, CDNA4).
//
// Everything here is written for 64-wide wavefronts and the f32-input matrix instruction
// v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, 32-cycle issue per SIMD, the f32
// vector rate).  A workgroup is 512 threads = 8 waves (2 per SIMD) working on a tile of TM rows
// (edges, nodes or pairs) whose activations stay in LDS between the fused stages; weights are
// pre-packed on the host into MFMA B-fragment order and streamed from L2.  TM is a template
// parameter of every kernel: the GVP kernels (edge message, node update) run 32-row tiles -- 77 KB
// of LDS, TWO workgroups per CU, i.e. 4 waves per SIMD -- or 16-row tiles while a batch is too small
// to give every CU a 32-row tile; EdgeUpdate runs 32-row tiles (4 workgroups per CU); the two-layer
// MLPs and projections run 64- or 16-row tiles (FM_TM below is THEIR default).  64-row GVP tiles
// (154 KB, one workgroup per CU) exist behind fm_config.tile_edge / tile_node for A/B runs only.
//
// Layout conventions
//   * LDS activation tiles are row-major [row][ld] f32 with (ld/4) odd, so that the A-fragment read
//     (lane (i=l&15,h=l>>4) reads float2 at row i, col 8*ks+2*h -> one ds_read_b64) is bank-conflict
//     free: rows i=0..15 start on distinct multiples of 4 banks and the two lane halves cover
//     disjoint 4-bank groups (MI355X_MICROARCH.md §LDS, ds_read_b64 = 2 x 32-lane groups, 64 banks).
//   * packed weights Wp[(ks*NT + nt)*64 + lane] = (W[8ks+2h][16nt+j], W[8ks+2h+1][16nt+j]),
//     j=lane&15, h=lane>>4: one coalesced 512-B global_load_dwordx2 per (ks, nt) per wave.
//   * the MFMA k-slots of lane-group h within k-superstep ks are k = 8ks+2h (x) and 8ks+2h+1 (y);
//     A and B use the same assignment, so the dot product is complete (summation order is a
//     permutation of k, which f32 parity tolerates: SURVEY.md §7 "hard parts").
//   * C/D fragment: acc[r] <-> row = 4*(lane>>4)+r, col = lane&15 (cdna_hip_programming.md §3).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Profiling aid, compiled out unless -DFM_PHASE_TIMING: thread 0 of every workgroup accumulates the shader
// cycles spent between consecutive FM_MARK(id) points into fm_tlog[id] (read back with fm_tlog_read, see
// tools/phase_timing.py).  This is how profiles/r01c_phase_cycles_edge_message.json was produced.
#ifdef FM_TRACE
// dev-only (-DFM_TRACE): absolute timestamps of selected marks per workgroup (first 16384 workgroups of launches with
// more than 20000 workgroups, i.e. fm_k_edge_message) plus the hardware id, to see how the phases of workgroups that
// share a CU line up (tools/trace_phases.py; profiles/r01e_trace_coresident.txt)
__device__ unsigned long long fm_trace[16384 * 16];
__device__ __forceinline__ int fm_trace_slot(int id) {
    return id == 0 ? 0 : id == 1 ? 1 : id == 12 ? 2 : id == 13 ? 3 : id == 22 ? 4 : id == 23 ? 5 : id == 32 ? 6 : id == 33 ? 7 : id == 41 ? 8 : -1;
}
#define FM_MARK_DECL if (threadIdx.x == 0 && blockIdx.x < 16384 && gridDim.x > 20000) { \
        fm_trace[blockIdx.x * 16 + 10] = __builtin_readcyclecounter(); \
        fm_trace[blockIdx.x * 16 + 9] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11))) | \
                                        ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32); }
#define FM_MARK_ARG , int fm_tb_
#define FM_MARK_PASS(b) , (b)
#define FM_MARK(id) do { if (threadIdx.x == 0 && blockIdx.x < 16384 && gridDim.x > 20000) { const int sl_ = fm_trace_slot(id); \
        if (sl_ >= 0) fm_trace[blockIdx.x * 16 + sl_] = __builtin_readcyclecounter(); } } while (0)
#define FM_MARKB(k) FM_MARK(fm_tb_ + (k))
#elif defined(FM_PHASE_TIMING)
__device__ unsigned long long fm_tlog[64];
#define FM_MARK_DECL unsigned long long fm_tl_ = __builtin_readcyclecounter();
#define FM_MARK_ARG , unsigned long long& fm_tl_, int fm_tb_
#define FM_MARK_PASS(b) , fm_tl_, (b)
#define FM_MARK(id) do { if (threadIdx.x == 0) { unsigned long long now_ = __builtin_readcyclecounter(); \
        atomicAdd(&fm_tlog[(id)], now_ - fm_tl_); fm_tl_ = now_; } } while (0)
#define FM_MARKB(k) FM_MARK(fm_tb_ + (k))
#else
#define FM_MARK_DECL
#define FM_MARK_ARG
#define FM_MARK_PASS(b)
#define FM_MARK(id) ((void)0)
#define FM_MARKB(k) ((void)0)
#endif

// Dev-only timing ablations (-DFM_ABLATE=<bit mask>; results are garbage, only the kernel time is meaningful): which part of
// the GVP kernels costs what.  1: no SiLU math, 2: no gate GEMM / gating, 4: no vector path ([Wh|Wcp] GEMM, cross products,
// norms, Wu GEMM), 8: no aggregation epilogue, 16: no prologue gathers, 32: no scalar-GEMM MFMAs (tools/ablate.sh, profiles/r02a_*).
#ifndef FM_ABLATE
#define FM_ABLATE 0
#endif

#define FM_TM 64          // rows per workgroup tile of the non-GVP kernels (MLPs, edge update, projections)
#define FM_THREADS 512    // threads per workgroup of the non-GVP kernels (8 waves)
#define FM_LDX 300        // scalar tile leading dim: >= 296, (300/4)=75 odd
#define FM_LDG 33         // gate tile leading dim

// SiLU / sigmoid on the hardware transcendental path: v_exp_f32 (2^x) and v_rcp_f32, ~1 ulp each.  Measured on
// MI355X (profiles/r01b_ab.jsonl): -8 % on the edge-message kernel vs expf + IEEE division, with unchanged parity
// (per-evaluation output error vs the CPU oracle 3-5e-7 either way).
__device__ __forceinline__ float fm_exp_neg(float x) { return __builtin_amdgcn_exp2f(x * -1.44269504088896341f); }
__device__ __forceinline__ float fm_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fm_exp_neg(x)); }
__device__ __forceinline__ float fm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + fm_exp_neg(x)); }

// Floating-point contraction is OFF for the whole library (flowmol_amd/build.py: -ffp-contract=off): the compiler never fuses a
// multiply with an add on its own, so the last bits of every result follow from the SOURCE, not from how a kernel's loops happen to be
// arranged (round 4 measured the same source giving different output fingerprints after a fill loop was rearranged: hipcc's default
// -ffp-contract=fast decides per call site).  Where a fused multiply-add is wanted -- one VALU instruction instead of two, and VALU
// instructions cost matrix-pipe issue slots on gfx950 -- it is written out: fm_fma(a, b, c) = a * b + c with ONE rounding (v_fma_f32).
__device__ __forceinline__ float fm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Correctly rounded, separately rounded f32 operations for the few places whose results must equal the
// reference's torch ops bit for bit (Euler step, purity-sampling probabilities).  With contraction off these are the plain
// operators; the pragmas keep them so under any build flags.
__device__ __forceinline__ float fm_mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float fm_add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float fm_sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float fm_div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

// Gaussian RBF, reference flowmol/utils/embedding.py:19-34: mu_k = k*Dmax/(R-1), sigma = Dmax/R
__device__ __forceinline__ float fm_rbf(float d, int k, float mu_step, float inv_sigma) {
    const float z = fm_fma(-(float)k, mu_step, d) * inv_sigma;      // d - k mu_step, one rounding
    return __builtin_amdgcn_exp2f(z * z * -1.44269504088896341f);
}

// distance with the reference's clamps: sqrt(max(|dx|^2,1e-8)) (+1e-8 added by the caller where the reference does)
__device__ __forceinline__ float fm_norm3(float dx, float dy, float dz) {
    const float s = fm_fma(dz, dz, fm_fma(dy, dy, dx * dx));
    return __builtin_amdgcn_sqrtf(fmaxf(s, 1e-8f));     // v_sqrt_f32 (~1 ulp); the argument is >= 1e-8, far from denormals, so
                                                        // sqrtf()'s scaling / fix-up sequence (8 more VALU per call) buys nothing
}

// ---------------------------------------------------------------------------------------------
// wave-level GEMM: acc[MT][NT] += A(lds rows m0.., K) * Wp(cols of tiles nt0..nt0+NT-1)
// ---------------------------------------------------------------------------------------------
// Packed-weight fragment load through a buffer descriptor (raw_buffer_load_b64): the wave-uniform part of the address
// (tile / k-superstep) travels in the SGPR soffset, the per-lane part is the constant VGPR voffset = lane*8, so the
// unrolled GEMM loops contain no VALU address arithmetic at all.  That matters on gfx950: VALU instructions take issue
// slots from the matrix pipe (tools/ubench/mfma_share.cpp: 4 VALU per 8 MFMAs cost 8 %, 16 cost 23 %, even with 4
// waves per SIMD), and 64-bit flat addresses cost 2 VALU per load.
//
// fm_buf(base, bytes) builds the descriptor (4 SGPRs) from a WAVE-UNIFORM base; the typed accessors take
// (voffset: per-lane bytes, soffset: wave-uniform bytes).  Raw-buffer range checking is part of the contract:
// an access with voffset (+ size) beyond `bytes` reads 0 / is dropped, which replaces compare+select pairs for
// ragged tiles and invalid rows (callers pass FM_BUF_OOB as voffset for "no row").
#define FM_BUF_OOB ((int)0x80000000)
__device__ __forceinline__ auto fm_buf(const void* base, unsigned bytes = 0x7fffffffu) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
template <class R>
__device__ __forceinline__ float fm_buf_f32(R rs, int voff, int soff) {
    const unsigned r = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
    return __builtin_bit_cast(float, r);
}
template <class R>
__device__ __forceinline__ float2 fm_buf_f32x2(R rs, int voff, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    static_assert(sizeof(r) == 8, "raw_buffer_load_b64 must return two dwords");
    const unsigned r0 = r[0], r1 = r[1];   // copy out first: __builtin_bit_cast on a vector-element lvalue reads element 0 (clang 19/ROCm 7.2)
    float2 out;
    out.x = __builtin_bit_cast(float, r0);
    out.y = __builtin_bit_cast(float, r1);
    return out;
}
template <class R>
__device__ __forceinline__ float4 fm_buf_f32x4(R rs, int voff, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    static_assert(sizeof(r) == 16, "raw_buffer_load_b128 must return four dwords");
    const unsigned r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    return make_float4(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1), __builtin_bit_cast(float, r2), __builtin_bit_cast(float, r3));
}
template <class R>
__device__ __forceinline__ void fm_buf_store_f32(R rs, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}
template <class R>
__device__ __forceinline__ void fm_buf_store_f32x2(R rs, int voff, int soff, float x, float y) {
    typedef unsigned fm_u32x2 __attribute__((ext_vector_type(2)));
    fm_u32x2 d;
    d[0] = __builtin_bit_cast(unsigned, x); d[1] = __builtin_bit_cast(unsigned, y);
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff, soff, 0);
}
template <class R>
__device__ __forceinline__ void fm_buf_store_f32x4(R rs, int voff, int soff, float4 v) {
    typedef unsigned fm_u32x4 __attribute__((ext_vector_type(4)));
    fm_u32x4 d;
    d[0] = __builtin_bit_cast(unsigned, v.x); d[1] = __builtin_bit_cast(unsigned, v.y);
    d[2] = __builtin_bit_cast(unsigned, v.z); d[3] = __builtin_bit_cast(unsigned, v.w);
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, soff, 0);
}
__device__ __forceinline__ float2 fm_wload(const float2* __restrict__ base /*wave-uniform*/, int byte_off /*wave-uniform*/, int lane) {
    return fm_buf_f32x2(fm_buf(base), lane * 8, byte_off);
}

// `wp` is the wave-uniform base of the wave's first column tile.
template <int MT, int NT>
__device__ __forceinline__ void fm_frag_load(float2 (&a)[MT], float2 (&b)[NT], const float* ap, int lda,
                                             const float2* __restrict__ wp, size_t wstep, int ks, int lane) {
    // volatile LDS-address-space load: keeps each A fragment a single ds_read_b64 (conflict-free with (ld/4) odd, 256 B/clk).  Without it
    // hipcc pairs them into ds_read2_b64, which is serviced in 16-lane groups over 32 banks at half the rate and
    // 2-way conflicts on this layout (MI355X_MICROARCH.md LDS table; SQ_LDS_BANK_CONFLICT was 40 % of LDS cycles).
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x2 t = *(const volatile __attribute__((address_space(3))) f32x2*)(ap + mt * 16 * lda + 8 * ks);
        a[mt].x = t[0]; a[mt].y = t[1];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = fm_wload(wp, (int)(((size_t)ks * wstep + (size_t)nt * 64) * sizeof(float2)), lane);
}

template <int MT, int NT>
__device__ __forceinline__ void fm_frag_mma(f32x4 (&acc)[MT][NT], const float2 (&a)[MT], const float2 (&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
}

// only the first of a k-superstep's two MFMA passes: k-slots 8ks + {0, 2, 4, 6}.  For a LAST superstep whose odd slots are zero by construction
// (fm_gvp_core lays the four cross-product norms out on the even slots of its last superstep) -- the all-zero second pass is not issued.
template <int MT, int NT>
__device__ __forceinline__ void fm_frag_mma_x(f32x4 (&acc)[MT][NT], const float2 (&a)[MT], const float2 (&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
}

// Software-pipelined over k-supersteps: the A (LDS) and B (L2) fragments of step ks+1 are requested
// before the MFMAs of step ks are issued, so one wave covers its own load latency (only 2 waves share
// a SIMD).  Two named register sets, manually unrolled by 2 (no runtime-indexed register arrays).
// TAILX: the odd k-slots of the LAST superstep are zero in A and in the packed weights -- its second MFMA pass is skipped (fm_frag_mma_x).
template <int MT, int NT, bool TAILX = false>
__device__ __forceinline__ void fm_wave_gemm(f32x4 (&acc)[MT][NT], const float* A, int lda, int K8,
                                             const float2* __restrict__ Wp, int ntiles, int nt0, int lane) {
    const float* ap = A + (lane & 15) * lda + 2 * (lane >> 4);
    const float2* wp = Wp + (size_t)nt0 * 64;          // wave-uniform
    const size_t wstep = (size_t)ntiles * 64;
    // Three register sets, prefetch distance 2: while the MFMAs of step k issue, the fragments of steps k+1 and k+2
    // are in flight (measured on MI355X: a fully unrolled variant with distance 2/4/6 was 2/7/12 % slower; distance 3 with four sets and buffer loads: no change).  sched_barrier(0) pins "request, then issue": without it hipcc's scheduler sinks every load down
    // to its first use (s_waitcnt vmcnt(0) right behind the global_load) and the pipelining is lost.
    float2 a0[MT], b0[NT], a1[MT], b1[NT], a2[MT], b2[NT];
    fm_frag_load<MT, NT>(a0, b0, ap, lda, wp, wstep, 0, lane);
    if (K8 > 1) fm_frag_load<MT, NT>(a1, b1, ap, lda, wp, wstep, 1, lane);
    int ks = 0;
    for (; ks + 3 <= K8; ks += 3) {
        if (ks + 2 < K8) fm_frag_load<MT, NT>(a2, b2, ap, lda, wp, wstep, ks + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        fm_frag_mma<MT, NT>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 3 < K8) fm_frag_load<MT, NT>(a0, b0, ap, lda, wp, wstep, ks + 3, lane);
        __builtin_amdgcn_sched_barrier(0);
        fm_frag_mma<MT, NT>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 4 < K8) fm_frag_load<MT, NT>(a1, b1, ap, lda, wp, wstep, ks + 4, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (TAILX && ks + 3 == K8) fm_frag_mma_x<MT, NT>(acc, a2, b2);
        else fm_frag_mma<MT, NT>(acc, a2, b2);
        __builtin_amdgcn_sched_barrier(0);
    }
    // tail: K8 - ks in {0, 1, 2} steps are already loaded in (a0,b0), (a1,b1)
    if (ks < K8) { if (TAILX && ks + 1 == K8) fm_frag_mma_x<MT, NT>(acc, a0, b0); else fm_frag_mma<MT, NT>(acc, a0, b0); }
    if (ks + 1 < K8) { if (TAILX) fm_frag_mma_x<MT, NT>(acc, a1, b1); else fm_frag_mma<MT, NT>(acc, a1, b1); }
}

// ---------------------------------------------------------------------------------------------
// 4-row GEMM on v_mfma_f32_4x4x1_16B_f32 for the node kernels of small batches (RG instances of fm_k_node_update, fm_k_mlp4): a tile of 4 RG nodes (RG groups
// of four rows), so that the nodes spread over more CUs and a 4-node tile's scalar GEMM is ~1 us of matrix time instead of 3.9 (the instruction runs at the
// full f32 rate: 8.5 cycles per 512 FLOP, profiles/r04h).  Sixteen 4x4 blocks per instruction; with A_b = X[0..3][k] for every block and B_b = W[k][4b..4b+3]
// one instruction adds X[0..3][k] (x) W[k][0..63] to a 4 x 64 output tile held as acc[r] = out[r][lane] (layout verified on the device:
// tools/ubench/mfma_4x4_layout.cpp).  Weights are quad-row packed, Wq4[(kq * GS + g) * 64 + lane] = W[4kq .. 4kq+3][64g + lane] (GS column groups): one 1-KB
// buffer_load_dwordx4 per four k and wave.  A operands: one ds_read_b128 (four k of row lane & 3; the 16 lanes of a row read the same address).
//
// THE SUMMATION ORDER IS THE REGULAR TILES' (round 6): a 16x16x4 accumulator of fm_wave_gemm is ONE fma chain from its initial value over the k-slots
// 8s + {0, 2, 4, 6} (first pass of superstep s: lanes 16 j .. 16 j + 15 hold k = 8s + 2j) and then 8s + {1, 3, 5, 7} (second pass) -- fm_frag_mma; measured on the
// MI355X: the hardware's 16x16x4 and 4x4x1 chains give identical bits (profiles/r06m_*, r06n_*).  Here the same chain runs with one k per instruction: quads 2s and
// 2s + 1, elements 0, 2 of both, then 1, 3 of both.  One accumulator per row group (no even / odd pair, no K split over waves -- rounds 4-5 had both: another order, faster by
// <= 1 us per launch): a 4 RG-node tile gives the bits of the regular 16- / 32-row tiles, so the tile height may follow the batch size in
// canonical mode.  All-zero products of padded k-slots add exactly 0: the padded tail needs no special case.  KQ even; PD quads of weights in flight.
template <int KQ, int RG, int PD = 8, int GS = 4>
__device__ __forceinline__ void fm_wave_gemm4(f32x4 (&acc)[RG], const float* X, int ldx, const void* Wq4, int g, int lane) {
    static_assert(KQ % 2 == 0 && PD % 2 == 0, "whole k-supersteps (two quads each)");
    constexpr int PA = 4;                            // A operands (LDS): two supersteps in flight
    const float* ap = X + (lane & 3) * ldx;
    const auto rs = fm_buf(Wq4);
    const int s0 = g * 1024;                         // wave-uniform byte offset of the first fragment; the others are compile-time multiples of GS KB away (GS column groups in the packing)
    f32x4 a[PA][RG];
    float4 b[PD];
    auto load_a = [&](int q, int k) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) a[q][rg] = *(const volatile __attribute__((address_space(3))) f32x4*)(ap + rg * 4 * ldx + 4 * k);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) if (q < KQ) b[q] = fm_buf_f32x4(rs, lane * 16, s0 + q * GS * 1024);
#pragma unroll
    for (int q = 0; q < PA; ++q) if (q < KQ) load_a(q, q);
#pragma unroll
    for (int k = 0; k < KQ; k += 2) {                // fully unrolled: every offset an immediate, exact wait counts
        const float4 b0 = b[k % PD], b1 = b[(k + 1) % PD];
        f32x4 a0[RG], a1[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) { a0[rg] = a[k % PA][rg]; a1[rg] = a[(k + 1) % PA][rg]; }
        if (k + PD < KQ) { b[k % PD] = fm_buf_f32x4(rs, lane * 16, s0 + (k + PD) * GS * 1024); b[(k + 1) % PD] = fm_buf_f32x4(rs, lane * 16, s0 + (k + 1 + PD) * GS * 1024); }
        if (k + PA < KQ) { load_a(k % PA, k + PA); load_a((k + 1) % PA, k + 1 + PA); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[rg][0], b0.x, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[rg][2], b0.z, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[rg][0], b1.x, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[rg][2], b1.z, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[rg][1], b0.y, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[rg][3], b0.w, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[rg][1], b1.y, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[rg][3], b1.w, acc[rg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// out[4][64 G] = X[4][4 KQ] * W for a 512-thread workgroup: wave g < G owns column group g and runs ONE fma chain per output element from 0 over all of K in the
// regular MLP tiles' order (fm_block_gemm starts at 0 and adds the bias in its epilogue, as epi does here) -- the bits of fm_mlp2_tile, so the 4-row tiles may
// be chosen by batch size in canonical mode.  epi(row, column, sum) runs on the accumulator registers; ends with a barrier.
template <int KQ, int G, class Epi>
__device__ __forceinline__ void fm_rows4_linear(const float* X, int ldx, const void* Wq4, Epi epi) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < G) {
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        fm_wave_gemm4<KQ, 1, 16, G>(acc, X, ldx, Wq4, wave, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(r, 64 * wave + lane, acc[0][r]);
    }
    __syncthreads();
}


// Single-tile (16x16) GEMM with compile-time K and chunked prefetch: all A/B fragments of chunk c+1 (CH k-supersteps)
// are requested before the MFMAs of chunk c.  A 1x1 tile has only 2 MFMAs (64 cycles) per k-superstep, far less
// than the L2 latency of its B fragment, so the one-step pipelining of fm_wave_gemm leaves it latency-bound
// (profiles/r01c: the 256->V gate GEMM took 6-10k cycles for 2k cycles of MFMA work).  Two accumulators (even/odd
// steps) break the 40-cycle dependent-accumulator chain; they are added at the end.
template <int K8, int CH>
__device__ __forceinline__ f32x4 fm_wave_gemm_1x1(const float* A, int lda, const float2* __restrict__ Wp, int ntiles, int nt, int lane) {
    static_assert(K8 % CH == 0, "K8 must be a multiple of the chunk size");
    constexpr int NCH = K8 / CH;
    const float* ap = A + (lane & 15) * lda + 2 * (lane >> 4);
    const float2* wp = Wp + (size_t)nt * 64;            // wave-uniform
    const size_t wstep = (size_t)ntiles * 64;
    float2 a[NCH > 1 ? 2 : 1][CH], b[NCH > 1 ? 2 : 1][CH];
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const f32x2 t = *(const volatile __attribute__((address_space(3))) f32x2*)(ap + 8 * q);
        a[0][q].x = t[0]; a[0][q].y = t[1];
        b[0][q] = fm_wload(wp, (int)((size_t)q * wstep * sizeof(float2)), lane);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int ks = (c + 1) * CH + q;
                const f32x2 t = *(const volatile __attribute__((address_space(3))) f32x2*)(ap + 8 * ks);
                a[(c + 1) & 1][q].x = t[0]; a[(c + 1) & 1][q].y = t[1];
                b[(c + 1) & 1][q] = fm_wload(wp, (int)((size_t)ks * wstep * sizeof(float2)), lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q + 1 < CH; q += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][q].x, b[c & 1][q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][q + 1].x, b[c & 1][q + 1].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][q].y, b[c & 1][q].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][q + 1].y, b[c & 1][q + 1].y, acc1, 0, 0, 0);
        }
        if (CH & 1) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][CH - 1].x, b[c & 1][CH - 1].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][CH - 1].y, b[c & 1][CH - 1].y, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

// block-level GEMM over an (mtiles x ntiles) grid of 16x16 output tiles: super-tiles of MT x NT tiles
// are dealt round-robin to the 8 waves; epi(row, col, value) is called for every output element.
// No barriers inside: the caller orders LDS hazards.
template <int MT, int NT, class Epi>
__device__ __forceinline__ void fm_block_gemm(const float* X, int ldx, int mtiles, int K8,
                                              const float2* __restrict__ Wp, int ntiles, Epi epi) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR: weight addresses become scalar
    const int sm = mtiles / MT, sn = ntiles / NT;
    const int nwaves = (int)(blockDim.x >> 6);
    for (int st = wave; st < sm * sn; st += nwaves) {
        const int m0 = (st / sn) * MT, n0 = (st % sn) * NT;
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        fm_wave_gemm<MT, NT>(acc, X + (size_t)m0 * 16 * ldx, ldx, K8, Wp, ntiles, n0, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    epi((m0 + i) * 16 + 4 * (lane >> 4) + r, (n0 + j) * 16 + (lane & 15), acc[i][j][r]);
    }
}

// ---------------------------------------------------------------------------------------------
// Opt-in split precision ("bf16x3", never the default): the two big GEMMs of an edge GVP (scalar linear, gates) on the bf16 matrix
// cores, every f32 value v carried as hi + lo with hi = bf16(v), lo = bf16(v - hi) (both round-to-nearest-even, residual <= 2^-18 |v|)
// and every product as hi*hi + hi*lo + lo*hi in f32 accumulators (dropped lo*lo <= 2^-18 |ab|).  This is NOT f32 arithmetic -- per-stage
// errors are ~10x the f32 path's (tests quote them) -- so it is reported separately and never used for the headline number.
//   * activations are split ONCE, when they are written to LDS, into two bf16 planes XH / XL [TM][FM_LDP] (4 bytes per element together,
//     like f32); splitting fragments at read time would cost ~24 VALU per MFMA triple;
//   * an A fragment of v_mfma_f32_16x16x32_bf16 is 8 consecutive k of one row = one ds_read_b128 per plane; FM_LDP * 2 bytes = 32 * odd
//     keeps the b128 lane groups on distinct banks (MI355X_MICROARCH.md LDS table);
//   * weights are split on the host and packed per (k32-block, column tile, plane, lane) as 16-byte entries: one coalesced 1-KB
//     buffer_load_dwordx4 per fragment, as many bytes per weight as f32.
// ---------------------------------------------------------------------------------------------
typedef short fm_h8 __attribute__((ext_vector_type(8)));          // eight bf16 bit patterns = one A / B fragment
typedef __bf16 fm_bf16x8 __attribute__((ext_vector_type(8)));
#define FM_LDP 336        // bf16 elements per row of a plane: >= 320 (ten k32 blocks), 336 * 2 B = 32 B * 21

__device__ __forceinline__ f32x4 fm_mfma_bf16(fm_h8 a, fm_h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fm_bf16x8, a), __builtin_bit_cast(fm_bf16x8, b), c, 0, 0, 0);
}
// Plane formats of the two-plane split modes: FMT 0 = bf16 (8 + 8 mantissa bits: "bf16x3"), FMT 1 = IEEE half (11 + 11 bits: "f16x3", round 5).  Half carries 22
// of f32's 24 mantissa bits in the same 4 bytes per element and the same three products on v_mfma_f32_16x16x32_f16 (subnormal inputs are kept by the
// instruction: tools/ubench/mfma_f16_denorm.cpp) -- its price is RANGE: |v| is clamped to the largest half, 65504, before the split (a finite, wrong result for
// activations beyond that instead of an infinity), and the low parts of values below 2^-3 are subnormal halves with an absolute resolution of 2^-25.
// The half-plane WEIGHTS are packed times 2^6 (fm_engine.cpp:pack_sp): the low part of a weight of magnitude 0.04 is 5e-6, a subnormal half with an absolute
// resolution of 3e-8 (7.5e-7 of the weight) -- times 64 it is a normal half and the pair carries 22 bits; accumulators start at (bias) * 2^6 and the sums are
// multiplied by 2^-6 (both exact).  With it a K = 296 product measures the f32 chain's own error (3.07e-7 rms against float64; 4.3e-7 without).
#define FM_F16_WSCALE 64.0f
template <int FMT> __device__ __forceinline__ constexpr float fm_sp_wscale() { return FMT == 1 ? FM_F16_WSCALE : 1.0f; }
typedef _Float16 fm_f16x8 __attribute__((ext_vector_type(8)));
template <int FMT>
__device__ __forceinline__ f32x4 fm_mfma_16(fm_h8 a, fm_h8 b, f32x4 c) {
    if constexpr (FMT == 1) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fm_f16x8, a), __builtin_bit_cast(fm_f16x8, b), c, 0, 0, 0);
    else return fm_mfma_bf16(a, b, c);
}
template <int FMT = 0>
__device__ __forceinline__ void fm_split(float v, unsigned short& hi, unsigned short& lo) {
    if constexpr (FMT == 1) {
        const float cl = fminf(fmaxf(v, -65504.f), 65504.f);        // one v_med3_f32
        v = (v != v) ? v : cl;                          // NaN stays NaN (fmaxf(NaN, x) = x would turn a blown-up network into finite +-65504: ADVICE r5)
        const _Float16 h = (_Float16)v;                // round-to-nearest-even
        const _Float16 l = (_Float16)(v - (float)h);   // v - hi is exact in f32
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const __bf16 h = (__bf16)v;                    // round-to-nearest-even
        const __bf16 l = (__bf16)(v - (float)h);       // v - hi is exact in f32
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    }
}
template <int LDP = FM_LDP, int FMT = 0>
__device__ __forceinline__ void fm_split_store(unsigned short* XH, unsigned short* XL, int row, int col, float v) {
    unsigned short hi, lo;
    fm_split<FMT>(v, hi, lo);
    XH[row * LDP + col] = hi;
    XL[row * LDP + col] = lo;
}
template <class R>
__device__ __forceinline__ fm_h8 fm_buf_h8(R rs, int voff, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return __builtin_bit_cast(fm_h8, r);
}

// acc[MT][NT] += A(planes, rows m0.., KB k32-blocks) * W(column tiles nt0..nt0+NT-1); Wsp: packed planes, wave-uniform base.
// LDP = row pitch of the planes in bf16 elements (16 * odd: 32-byte units odd)
template <int MT, int NT, int LDP>
__device__ __forceinline__ void fm_sp_frag_load(fm_h8 (&ah)[MT], fm_h8 (&al)[MT], fm_h8 (&bh)[NT], fm_h8 (&bl)[NT], const unsigned short* aph, const unsigned short* apl,
                                                const void* wsp, int ntiles, int nt0, int kb, int lane) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *(const volatile __attribute__((address_space(3))) fm_h8*)(aph + mt * 16 * LDP + kb * 32);      // one ds_read_b128
        al[mt] = *(const volatile __attribute__((address_space(3))) fm_h8*)(apl + mt * 16 * LDP + kb * 32);
    }
    const auto rs = fm_buf(wsp);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int e = ((kb * ntiles + nt0 + nt) * 2) * 64 * 16;          // bytes: entry (kb, tile, plane 0)
        bh[nt] = fm_buf_h8(rs, lane * 16, e);
        bl[nt] = fm_buf_h8(rs, lane * 16, e + 64 * 16);
    }
}
template <int MT, int NT, int FMT = 0>
__device__ __forceinline__ void fm_sp_frag_mma(f32x4 (&acc)[MT][NT], const fm_h8 (&ah)[MT], const fm_h8 (&al)[MT], const fm_h8 (&bh)[NT], const fm_h8 (&bl)[NT]) {
    // the two small products first, the big one last; three passes over the accumulators keep dependent MFMAs apart
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = fm_mfma_16<FMT>(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = fm_mfma_16<FMT>(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = fm_mfma_16<FMT>(ah[mt], bh[nt], acc[mt][nt]);
}
template <int MT, int NT, int LDP = FM_LDP, int FMT = 0>
__device__ __forceinline__ void fm_wave_gemm_sp(f32x4 (&acc)[MT][NT], const unsigned short* XH, const unsigned short* XL, int row0, int KB,
                                                const void* wsp, int ntiles, int nt0, int lane) {
    static_assert(LDP % 16 == 0 && (LDP / 16) % 2 == 1, "plane pitch must be 32 bytes * odd");
    const unsigned short* aph = XH + (row0 + (lane & 15)) * LDP + 8 * (lane >> 4);
    const unsigned short* apl = XL + (row0 + (lane & 15)) * LDP + 8 * (lane >> 4);
    fm_h8 ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
    // One accumulator for all three products, in both plane formats.  (A separate accumulator for the half mode's two cross products was measured: no
    // better on the hardware -- worst stage 1.42 x the f32 kernels' error against float64 vs 1.38 x with one -- and its 16 VGPRs push the edge kernel
    // from 118 to 134, i.e. to ONE workgroup per CU: 93 instead of 107 molecules/s; profiles/r05f_*.)
    auto mma = [&](const fm_h8 (&ah)[MT], const fm_h8 (&al)[MT], const fm_h8 (&bh)[NT], const fm_h8 (&bl)[NT]) { fm_sp_frag_mma<MT, NT, FMT>(acc, ah, al, bh, bl); };
    fm_sp_frag_load<MT, NT, LDP>(ah0, al0, bh0, bl0, aph, apl, wsp, ntiles, nt0, 0, lane);
    int kb = 0;
    for (; kb + 2 <= KB; kb += 2) {          // double-buffered: the fragments of block kb+1 are requested before the MFMAs of block kb issue
        fm_sp_frag_load<MT, NT, LDP>(ah1, al1, bh1, bl1, aph, apl, wsp, ntiles, nt0, kb + 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 2 < KB) fm_sp_frag_load<MT, NT, LDP>(ah0, al0, bh0, bl0, aph, apl, wsp, ntiles, nt0, kb + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kb < KB) mma(ah0, al0, bh0, bl0);
}

// ---------------------------------------------------------------------------------------------
// Opt-in THREE-term split ("bf16x6", edge-message kernel only; VERDICT r4 #5: can the f32 roof be beaten at f32-class accuracy?): every f32 value v is
// carried as hi + mid + lo, three bf16 = all 24 mantissa bits (the residual after three round-to-nearest steps is <= 2^-27 |v|), every product as
// hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi) in f32 accumulators; the three dropped products are <= 3 * 2^-24 |ab| -- f32 rounding class (a
// K = 296 GEMM measured on the host: rms error 1.2e-7 against f64, plain f32 3.1e-7, the two-term mode 4.4e-6).  The price: six matrix instructions per
// k32 block instead of three, three planes per operand -- 6 bytes per weight in the stream and per activation in LDS, which at 32-row tiles leaves room
// for ONE workgroup per CU.  Planes are consecutive: plane k of a tile = XH + k * (XL - XH).
// ---------------------------------------------------------------------------------------------
template <int LDP = FM_LDP>
__device__ __forceinline__ void fm_split3_store(unsigned short* XH, unsigned short* XL, int row, int col, float v) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;                  // exact
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);       // exact subtraction, one rounding
    unsigned short* X2 = XL + (XL - XH);
    XH[row * LDP + col] = __builtin_bit_cast(unsigned short, h);
    XL[row * LDP + col] = __builtin_bit_cast(unsigned short, m);
    X2[row * LDP + col] = __builtin_bit_cast(unsigned short, l);
}
template <int MT, int NT, int LDP>
__device__ __forceinline__ void fm_sp3_frag_load(fm_h8 (&a)[3][MT], fm_h8 (&b)[3][NT], const unsigned short* ap, int pstride, const void* wsp, int ntiles, int nt0, int kb, int lane) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[p][mt] = *(const volatile __attribute__((address_space(3))) fm_h8*)(ap + p * pstride + mt * 16 * LDP + kb * 32);
    const auto rs = fm_buf(wsp);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int e = ((kb * ntiles + nt0 + nt) * 3) * 64 * 16;          // bytes: entry (kb, tile, plane 0); the planes of an entry are 1 KB apart
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p][nt] = fm_buf_h8(rs, lane * 16, e + p * 64 * 16);
    }
}
// The five correction products -- 2^-8 ... 2^-16 of the result -- go to an accumulator of their own (accs), only hi * hi to the main one: a rounding of accs is
// 2^-8 of a rounding of the full sum, so the split result carries one full-magnitude rounding per k32 block like an f32 chain carries one per k
template <int MT, int NT>
__device__ __forceinline__ void fm_sp3_frag_mma(f32x4 (&acc)[MT][NT], f32x4 (&accs)[MT][NT], const fm_h8 (&a)[3][MT], const fm_h8 (&b)[3][NT]) {
    constexpr int PA[5] = {2, 0, 1, 1, 0}, PB[5] = {0, 2, 1, 0, 1};          // (lo, hi), (hi, lo), (mid, mid), (mid, hi), (hi, mid): smallest first
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) accs[mt][nt] = fm_mfma_bf16(a[PA[q]][mt], b[PB[q]][nt], accs[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = fm_mfma_bf16(a[0][mt], b[0][nt], acc[mt][nt]);
}
// acc += A (three planes XH, XL, XL + (XL - XH); rows row0.., KB k32 blocks) * W (three-plane packing, column tiles nt0 .. nt0 + NT - 1)
template <int MT, int NT, int LDP = FM_LDP>
__device__ __forceinline__ void fm_wave_gemm_sp3(f32x4 (&acc)[MT][NT], const unsigned short* XH, const unsigned short* XL, int row0, int KB,
                                                 const void* wsp, int ntiles, int nt0, int lane) {
    const unsigned short* ap = XH + (row0 + (lane & 15)) * LDP + 8 * (lane >> 4);
    const int pstride = (int)(XL - XH);
    fm_h8 a0[3][MT], b0[3][NT], a1[3][MT], b1[3][NT];
    f32x4 accs[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accs[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    fm_sp3_frag_load<MT, NT, LDP>(a0, b0, ap, pstride, wsp, ntiles, nt0, 0, lane);
    int kb = 0;
    for (; kb + 2 <= KB; kb += 2) {
        fm_sp3_frag_load<MT, NT, LDP>(a1, b1, ap, pstride, wsp, ntiles, nt0, kb + 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        fm_sp3_frag_mma<MT, NT>(acc, accs, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 2 < KB) fm_sp3_frag_load<MT, NT, LDP>(a0, b0, ap, pstride, wsp, ntiles, nt0, kb + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        fm_sp3_frag_mma<MT, NT>(acc, accs, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kb < KB) fm_sp3_frag_mma<MT, NT>(acc, accs, a0, b0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += accs[mt][nt];
}

// ---------------------------------------------------------------------------------------------
// One Geometric Vector Perceptron on a 64-row tile (reference flowmol/models/gvp.py:90-133)
// ---------------------------------------------------------------------------------------------
struct FmGvpW {
    const float2* Wv1;   // [Wh | Wcp] packed, K = V, N = V+16          (unused when FIRST)
    const float2* Wu;    // packed, K = V+8, N = VOUT padded to 16
    const float2* Ws;    // scalar linear packed, K = (FIRST ? 160 : 256) + V + 8, N = 256
    const float* bs;     // (256)
    const float2* Wg;    // gates packed, K = 256, N = VOUT padded to 16
    const float* bg;     // (VOUT padded)
    const void* Ws4;     // Ws quad-row packed for fm_wave_gemm4 (node-side GVPs; RG instances)
    const void* Ws_sp;   // split-precision builds only: Ws / Wg as hi/lo bf16 planes in v_mfma_f32_16x16x32_bf16 B-fragment order
    const void* Wg_sp;
};

// LDS tile geometry shared by every GVP-based kernel; TM = rows (edges or nodes) per workgroup tile:
// 64 -> 154 KB of LDS, one workgroup per CU; 32 -> 77 KB, two workgroups per CU whose MFMA and
// VALU/LDS/gather phases overlap each other.
// HX = extra hidden vector channels of the FIRST edge-message GVP: the reduced destination-node vectors of
// use_dst_feats models (gvp.py:527-529; V/4 with configs/dev.yml), 0 otherwise.
template <int V, int TM, int HX = 0>
struct FmGvpTile {
    static constexpr int H0 = V + 1 + HX;                     // hidden vectors of the first edge GVP: [x_diff | v_src | v_dst_msg]
    static constexpr int KU0 = (H0 + 4 + 7) / 8 * 8;          // K of its Wu GEMM: hidden + 4 cross products, padded
    static constexpr int PVW = (KU0 + 8 + 15) / 16 * 16;      // its hidden-vector row: [hidden | cp | 0 .. KU0) | Vcp sources (8) | 0]
    static constexpr int LDVI = V + 4;       // Vin  [3*TM][LDVI]   (36 | 20: /4 odd)
    static constexpr int LDVH = PVW + 4;     // Vh   [3*TM][LDVH]   (HX = 0: V+20 = 52 | 36; /4 odd in every instance)
    static constexpr int KU = V + 8;         // K of the other GVPs' Wu GEMM (hidden V + 4 cp (+pad))
    static_assert(((PVW + 4) / 4) % 2 == 1 && PVW >= V + 16, "Vh leading dimension must keep ds_read_b64 conflict-free");
    static constexpr int X_FLOATS = TM * FM_LDX;
    static constexpr int VIN_FLOATS = 3 * TM * LDVI;
    static constexpr int VH_FLOATS = 3 * TM * LDVH;
    static constexpr int G_FLOATS = TM * FM_LDG;
    static constexpr int TOTAL_FLOATS = X_FLOATS + VIN_FLOATS + VH_FLOATS + G_FLOATS;
};

// State on entry
//   FIRST : Vh[xyz*TM+r][0..H0)   = hidden vectors (H0 = V+1+HX), Vh[..][H0..KU0) = 0, Vh[..][KU0..KU0+8) = Vcp (8)   (HX = 0: KU0 = V+8),
//           X[r][0..159]          = [rbf(32) | ef(128)]
//   !FIRST: Vin[xyz*TM+r][0..V-1] = input vectors,  X[r][0..255] = input scalars
// State on exit: X[r][0..255] = scalar output (SiLU), Vin[xyz*TM+r][0..VOUT-1] = gated vector output.
// `pre`: per-accumulator-element addend of the scalar linear, used by FIRST only (fm_gather_pre); the bias is added here.
// All 512 threads must call it (it contains barriers); it ends with a barrier.
// SP = 1 (split precision, edge message only): X is NOT an f32 tile but the two bf16 planes XH = (u16*)X, XL = XH + TM*FM_LDP; the
// scalar GEMM and the gate GEMM run on v_mfma_f32_16x16x32_bf16 with hi/lo operands; with LAST the f32 scalar output is kept in
// registers and written as a plain f32 [TM][FM_LDX] tile over the (then dead) planes for the aggregation.  G must then alias Vh + TM*FM_LDG.
// PQ (FIRST only): the [rbf | ef] slab of the scalar linear arrives inside `pre` (per-pair table Q, FmMlpArgs::slabQ0): X holds only the hidden-vector
// norms sh at columns [0, KU0) and the scalar GEMM has K = KU0.
// RG > 0 (node kernels of small batches): only rows 0 .. 4 RG - 1 of the TM-row frame are nodes; the scalar GEMM -- the one phase whose cost
// scales with the tile height -- runs on those rows with fm_wave_gemm4 (64 columns on each of waves 0..3, one fma chain per output element in the k-order of the
// regular tiles' 16x16x4 MFMAs: the bits of the regular 16- / 32-row tiles); every other phase is the TM-row code over the frame.
template <int V, int VOUT, bool FIRST, bool SIGMOID, int TM, int NTH, int HX = 0, int SP = 0, bool LAST = false, bool PQ = false, int RG = 0>
__device__ __forceinline__ void fm_gvp_core(float* X, float* Vin, float* Vh, float* G, const FmGvpW& w,
                                            float (&pre)[TM / 16][1024 / NTH][4] FM_MARK_ARG) {
    typedef FmGvpTile<V, TM, HX> T;
    unsigned short* const XH = reinterpret_cast<unsigned short*>(X);
    unsigned short* const XL = XH + TM * FM_LDP;
    constexpr int FMT = SP == 3 ? 1 : 0;                 // SP: 1 = two bf16 planes, 2 = three bf16 planes, 3 = two half planes
    constexpr int MT = TM / 16;                          // row tiles of the scalar GEMM
    constexpr int NW = NTH / 64;                         // waves per workgroup
    constexpr int NTW = 16 / NW;                         // column tiles of the scalar GEMM per wave (16 tiles = 256 columns)
    constexpr int H = FIRST ? T::H0 : V;                 // hidden vector channels
    constexpr int KUC = FIRST ? T::KU0 : T::KU;          // K of this GVP's Wu GEMM = width of [hidden | cp | pad] in Vh and of sh in X
    static_assert(!PQ || (FIRST && !SP), "PQ is a variant of the first f32 edge GVP");
    static_assert(RG == 0 || (!FIRST && !SP && 4 * RG <= TM && NTH == 512), "RG is a variant of the non-first f32 GVP: 4 RG nodes in a TM-row frame");
    constexpr int SOFF = PQ ? 0 : (FIRST ? 160 : 256);   // where sh goes in X
    constexpr int K8S = (SOFF + KUC) / 8;
    constexpr int VOP = VOUT < 16 ? 16 : VOUT;           // padded vector-out width
    constexpr int CPSRC = FIRST ? T::KU0 : V;            // where the 8 Vcp channels sit in Vh
    // sh layout in X: [norms of the H hidden channels | the 4 cross-product norms | 0].  Non-first GVPs: H is a multiple of 8, so the last
    // k-superstep of the scalar GEMM holds only the four cross-product norms -- laid out on its EVEN k-slots (H, H+2, H+4, H+6; zeros between,
    // weights packed to match: pack_gvp) so that the superstep's second MFMA pass is all zero and is skipped (TAILX): 1 of 74 passes per GEMM.
    constexpr int CPS = FIRST ? 1 : 2;                   // column stride of the cross-product norms
    static_assert(FIRST || (H % 8 == 0 && KUC == H + 8), "the interleaved tail needs the cross-product norms alone in the last k-superstep");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // biases of this lane's accumulator columns, requested first: their L2 latency hides behind the vector phases
    float bias_s[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bias_s[j] = w.bs[(NTW * wave + j) * 16 + (lane & 15)];
    // ... and the gate bias of this thread's channel in the gating loop (NTH is a multiple of VOUT: the channel is the same in every iteration):
    // a global load inside that loop sat on the critical path of every GVP of a small batch (profiles/r04i: 0.5 us of an 8.8-us GVP)
    static_assert(NTH % VOUT == 0, "one gate channel per thread");
    const float bias_g = w.bg[tid % VOUT];

    if (!FIRST && !(FM_ABLATE & 4)) {
        // Vh[:, 0..V+15] = Vin(3TM x V) * [Wh | Wcp | 0]
        constexpr int RT = 3 * TM / 16, CT = (V + 16) / 16;
        if constexpr (CT == 3) {
            // V = 32: column tiles {0, 1} as one job, {2} ([Wcp | 0]) as another: 2 RT jobs instead of 3 RT single-tile jobs -- one round of the
            // eight waves for a 16-row tile instead of two (each job is four k-steps behind one L2 round trip: the phase is latency, not MFMA time)
            for (int job = wave; job < 2 * RT; job += NW) {
                const int rt = job >> 1;
                const float* A = Vin + (size_t)rt * 16 * T::LDVI;
                float* D = Vh + (rt * 16 + 4 * (lane >> 4)) * T::LDVH + (lane & 15);
                if (job & 1) {
                    f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
                    fm_wave_gemm<1, 1>(acc, A, T::LDVI, V / 8, w.Wv1, CT, 2, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[r * T::LDVH + 32] = acc[0][0][r];
                } else {
                    f32x4 acc[1][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
                    fm_wave_gemm<1, 2>(acc, A, T::LDVI, V / 8, w.Wv1, CT, 0, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { D[r * T::LDVH] = acc[0][0][r]; D[r * T::LDVH + 16] = acc[0][1][r]; }
                }
            }
        } else {
            fm_block_gemm<1, 1>(Vin, T::LDVI, RT, V / 8, w.Wv1, CT, [&](int row, int col, float v) { Vh[row * T::LDVH + col] = v; });
        }
        __syncthreads();
    }
    FM_MARKB(0);
    // One phase, no barrier inside: (a) threads < 4*TM form the cross products cp_p = a_p x b_p,
    // (a,b) = Vcp[0..3], Vcp[4..7] (gvp.py:105-112), store them at columns H..H+3 of Vh (each thread touches
    // only its own columns) and their norms into X; (b) all threads compute the norms sh of the H plain hidden
    // channels (gvp.py:116, _norm_no_nan clamp) -> X[:, SOFF..SOFF+H) and clear the K padding of X.
    static_assert(TM * 4 <= NTH, "cross-product phase needs 4 threads per row");
    if (!(FM_ABLATE & 4)) {
    if (tid < TM * 4) {
        const int r = tid >> 2, p = tid & 3;
        float a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = Vh[(c * TM + r) * T::LDVH + CPSRC + p];
            b[c] = Vh[(c * TM + r) * T::LDVH + CPSRC + 4 + p];
        }
        const float cx = fm_fma(a[1], b[2], -(a[2] * b[1]));
        const float cy = fm_fma(a[2], b[0], -(a[0] * b[2]));
        const float cz = fm_fma(a[0], b[1], -(a[1] * b[0]));
        if (!FIRST) {   // the b-channels sit inside the K range of the Wu GEMM: clear them
#pragma unroll
            for (int c = 0; c < 3; ++c) Vh[(c * TM + r) * T::LDVH + V + 4 + p] = 0.f;
        }
        Vh[(0 * TM + r) * T::LDVH + H + p] = cx;
        Vh[(1 * TM + r) * T::LDVH + H + p] = cy;
        Vh[(2 * TM + r) * T::LDVH + H + p] = cz;
        if (SP == 2) fm_split3_store(XH, XL, r, SOFF + H + CPS * p, fm_norm3(cx, cy, cz));
        else if (SP) fm_split_store<FM_LDP, FMT>(XH, XL, r, SOFF + H + CPS * p, fm_norm3(cx, cy, cz));
        else X[r * FM_LDX + SOFF + H + CPS * p] = fm_norm3(cx, cy, cz);
    }
    // thread -> (row, 16-column group): no integer division by V+8 in the index math
    for (int r = tid >> 4; r < TM; r += NTH / 16) {
#pragma unroll
        for (int j = 0; j < (KUC + 15) / 16; ++j) {
            const int c = (tid & 15) + 16 * j;
            if (c < H) {
                const float vx = Vh[(0 * TM + r) * T::LDVH + c];
                const float vy = Vh[(1 * TM + r) * T::LDVH + c];
                const float vz = Vh[(2 * TM + r) * T::LDVH + c];
                if (SP == 2) fm_split3_store(XH, XL, r, SOFF + c, fm_norm3(vx, vy, vz));
                else if (SP) fm_split_store<FM_LDP, FMT>(XH, XL, r, SOFF + c, fm_norm3(vx, vy, vz));
                else X[r * FM_LDX + SOFF + c] = fm_norm3(vx, vy, vz);
            } else if (FIRST ? (c >= H + 4 && c < KUC) : (c < KUC && ((c - H) & 1))) {      // K padding (first GVP) / the odd slots between the cross-product norms
                if (SP) { XH[r * FM_LDP + SOFF + c] = 0; XL[r * FM_LDP + SOFF + c] = 0; if (SP == 2) XL[TM * FM_LDP + r * FM_LDP + SOFF + c] = 0; }
                else X[r * FM_LDX + SOFF + c] = 0.f;
            }
        }
    }
    __syncthreads();
    FM_MARKB(1);
    // Vu = Vh_full * Wu -> Vin (the input vectors are dead by now)
    fm_block_gemm<1, 1>(Vh, T::LDVH, 3 * TM / 16, KUC / 8, w.Wu, VOP / 16,
                        [&](int row, int col, float v) { Vin[row * T::LDVI + col] = v; });
    }
    // scalar linear: TM x K -> 256, wave w owns column tiles NTW*w .. NTW*w+NTW-1 for all MT row tiles
    float keep[(SP && LAST) ? MT : 1][(SP && LAST) ? NTW : 1][4];     // split precision, last GVP: the f32 scalar output for the aggregation
    if constexpr (RG > 0) {
        constexpr int KQ = (SOFF + KUC) / 4;
        const int g = wave & 3;
        FM_MARKB(2);
        f32x4 acc[RG];
        if (wave < 4) {
            const float b4 = w.bs[64 * g + lane];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{b4, b4, b4, b4};
            if (!(FM_ABLATE & 32)) fm_wave_gemm4<KQ, RG>(acc, X, FM_LDX, w.Ws4, g, lane);
        }
        FM_MARKB(3);
        __syncthreads();                      // every wave has finished reading X (and Vh)
        FM_MARKB(4);
        if (wave < 4) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(4 * rg + r) * FM_LDX + 64 * g + lane] = fm_silu(acc[rg][r]);
        }
        __syncthreads();
    } else {
        // The accumulators start at bias (+ `pre` for the FIRST GVP of an edge tile: the hoisted W_s*s[src] term, requested
        // long before so its latency is hidden) instead of zero: no separate add in the epilogue (VALU instructions cost
        // matrix-pipe issue slots).
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float bias = bias_s[j];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (FIRST ? pre[i][j][r] + bias : bias) * fm_sp_wscale<FMT>();
        }
        FM_MARKB(2);
        if (SP == 2) fm_wave_gemm_sp3<MT, NTW>(acc, XH, XL, 0, (SOFF + KUC + 31) / 32, w.Ws_sp, 16, NTW * wave, lane);
        else if (SP) { if (!(FM_ABLATE & 32)) fm_wave_gemm_sp<MT, NTW, FM_LDP, FMT>(acc, XH, XL, 0, (SOFF + KUC + 31) / 32, w.Ws_sp, 16, NTW * wave, lane); }
        else if (!(FM_ABLATE & 32)) fm_wave_gemm<MT, NTW, !FIRST>(acc, X, FM_LDX, K8S, w.Ws, 16, NTW * wave, lane);
        FM_MARKB(3);
        __syncthreads();                      // every wave has finished reading X (and Vh)
        FM_MARKB(4);
        if (SP) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = fm_silu(acc[i][j][r] * (1.0f / fm_sp_wscale<FMT>()));
                        if constexpr (SP && LAST) keep[i][j][r] = y;
                        if (SP == 2) fm_split3_store(XH, XL, i * 16 + 4 * (lane >> 4) + r, (NTW * wave + j) * 16 + (lane & 15), y);
                        else fm_split_store<FM_LDP, FMT>(XH, XL, i * 16 + 4 * (lane >> 4) + r, (NTW * wave + j) * 16 + (lane & 15), y);
                    }
        } else {
            float* xo = X + (4 * (lane >> 4)) * FM_LDX + NTW * wave * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xo[(i * 16 + r) * FM_LDX + j * 16] = (FM_ABLATE & 1) ? acc[i][j][r] : fm_silu(acc[i][j][r]);
        }
        __syncthreads();
    }
    FM_MARKB(5);
    // gates = Linear(256 -> VOUT)(scalar out)   (gvp.py:122-128): NJ = (TM/16) x (VOP/16) single-tile jobs, K = 256.
    // K is split in two: jobs [0,NJ) take k < 128 and write G, jobs [NJ,2NJ) take k >= 128 and write a partial tile in Vh (dead since the
    // scalar GEMM); the gating loop adds the parts and the bias.
    // (The 64-MFMA dependent chain of an unsplit job was the critical path of this phase: profiles/r01c.)
    constexpr int NJ = (TM / 16) * (VOP / 16);
    // CANONICAL ORDER (round 6): two K halves for EVERY tile height, so that a row's gates -- like every other sum of this file -- do not depend on whether
    // its batch ran 16-, 32- or 64-row tiles (a molecule alone and the same molecule inside a 1024-batch give the same bits).  Round 4's four K quarters
    // for 16-row tiles were worth 1.4 of 68 us per node tile (profiles/r04j); 64-row tiles (A/B only) take two rounds of eight jobs.
    constexpr int KS = 2;
    // partial q >= 1 lives in slot (q - 1) of Vh (dead since the scalar GEMM), in units of a gate tile; split-precision instances keep G itself in
    // slot 1 of Vh, so their partials 2, 3 move up one slot
    auto gpart = [&](int q) { return Vh + (SP && q >= 2 ? q : q - 1) * TM * FM_LDG; };
    static_assert((KS - 1 + (SP ? 1 : 0)) * TM * FM_LDG <= T::VH_FLOATS, "gate partial sums must fit into Vh");
    if (FM_ABLATE & 2) return;
    for (int jw = wave; jw < NJ * KS; jw += NW) {
        const int job = jw % NJ, half = jw / NJ;
        const int m0 = job / (VOP / 16), n0 = job % (VOP / 16);
        f32x4 g;
        if (SP) {
            f32x4 ga[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
            if (SP == 2) fm_wave_gemm_sp3<1, 1>(ga, XH + half * (256 / KS), XL + half * (256 / KS), m0 * 16, 8 / KS,
                                                static_cast<const char*>(w.Wg_sp) + (size_t)half * (8 / KS) * (VOP / 16) * 3 * 64 * 16, VOP / 16, n0, lane);
            else fm_wave_gemm_sp<1, 1, FM_LDP, FMT>(ga, XH + half * (256 / KS), XL + half * (256 / KS), m0 * 16, 8 / KS,
                                  static_cast<const char*>(w.Wg_sp) + (size_t)half * (8 / KS) * (VOP / 16) * 2 * 64 * 16, VOP / 16, n0, lane);
            g = ga[0][0] * (1.0f / fm_sp_wscale<FMT>());
        } else {
            g = fm_wave_gemm_1x1<32 / KS, 8>(X + (size_t)m0 * 16 * FM_LDX + half * (256 / KS), FM_LDX,
                                             w.Wg + (size_t)half * (32 / KS) * (VOP / 16) * 64, VOP / 16, n0, lane);
        }
        float* go = (half ? gpart(half) : G) + (m0 * 16 + 4 * (lane >> 4)) * FM_LDG + n0 * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) go[r * FM_LDG] = g[r];
    }
    FM_MARKB(6);
    __syncthreads();
    // gating: one (row, channel) pair per thread and iteration, its gate applied to the three spatial components
    for (int idx = tid; idx < TM * VOUT; idx += NTH) {
        const int r = idx / VOUT, u = idx % VOUT;
        float gv = G[r * FM_LDG + u] + bias_g;
#pragma unroll
        for (int q = 1; q < KS; ++q) gv += gpart(q)[r * FM_LDG + u];
        if (SIGMOID) gv = fm_sigmoid(gv);
#pragma unroll
        for (int c = 0; c < 3; ++c) Vin[(c * TM + r) * T::LDVI + u] *= gv;
    }
    if constexpr (SP && LAST) {      // every gate GEMM has read the planes (barrier above): the region now receives the plain f32 scalar messages
        float* xo = X + (4 * (lane >> 4)) * FM_LDX + NTW * wave * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[(i * 16 + r) * FM_LDX + j * 16] = keep[i][j][r];
    }
    __syncthreads();
    FM_MARKB(7);
}

// per-element addend of the first edge GVP's scalar linear: addend[row of rowoff[..]][col] for this lane's accumulator elements.
// rowoff[] holds ready BYTE offsets of the (nrows, 256) fp32 table rows (row * 1024), FM_BUF_OOB for "no row" (reads 0 through the
// buffer range check; adding the lane's column offset keeps it out of range): one v_add per row instead of compare + scale + select.
template <int TM, int NTH, bool ACCUM>
__device__ __forceinline__ void fm_gather_pre(float (&pre)[TM / 16][1024 / NTH][4], const float* __restrict__ addend, int nrows, const int* rowoff) {
    constexpr int NTW = 1024 / NTH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = fm_buf(addend, (unsigned)nrows * 1024u);
#pragma unroll
    for (int i = 0; i < TM / 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int voff = rowoff[i * 16 + 4 * (lane >> 4) + r] + (lane & 15) * 4;
#pragma unroll
            for (int j = 0; j < NTW; ++j) { const float t = fm_buf_f32(rs, voff, (NTW * wave + j) * 64); pre[i][j][r] = ACCUM ? pre[i][j][r] + t : t; }
        }
}

// Sum over a group of LPR consecutive lanes (LPR = 2 .. 32, group-aligned), every lane receiving the total: bit for bit the xor butterfly
// `for (o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o)`.  After each level all lanes of a 2^k group hold the same value (IEEE addition commutes), so
// the partner of the next level may be ANY lane of the neighbouring group: quad_perm [1,0,3,2] / [2,3,0,1] for o = 1, 2, row_half_mirror for 4,
// row_mirror for 8 -- one v_add_f32 with a DPP operand per level instead of address arithmetic (4 VALU) + ds_bpermute + add; o = 16 swaps the
// two rows of 16 with ds_swizzle (no address register).
template <int CTRL>
__device__ __forceinline__ float fm_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPR>
__device__ __forceinline__ float fm_group_sum(float s) {
    static_assert(LPR == 2 || LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32, "group of 2 .. 32 lanes");
    s += fm_dpp<0xB1>(s);
    if constexpr (LPR > 2) s += fm_dpp<0x4E>(s);
    if constexpr (LPR > 4) s += fm_dpp<0x141>(s);
    if constexpr (LPR > 8) s += fm_dpp<0x140>(s);
    if constexpr (LPR > 16) s += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s), 0x401F));   // lane ^ 16
    return s;
}

// LayerNorm statistics of one LDS row handled by a group of LPR consecutive lanes (LPR = 8 or 16):
// two-pass mean / biased variance like torch.nn.functional.layer_norm.  All lanes must call it.
// CANONICAL ORDER (round 6): a row owned by 32 lanes (16-row tiles) is summed in the 16-lane order of the 32-row tiles -- lanes sub and sub ^ 16 form the
// same strided partial sums and each aligned group of 16 runs the same butterfly, so both halves hold identical statistics -- which makes every LayerNorm
// independent of the tile height its batch happened to run (FM_LN_LANES<LPR> / fm_ln_sub).  8-lane rows (64-row tiles: A/B configurations only) keep their own order.
template <int LPR> struct FmLnLanes { static constexpr int value = LPR > 16 ? 16 : LPR; };
template <int LPR> __device__ __forceinline__ int fm_ln_sub(int sub) { return LPR > 16 ? (sub & 15) : sub; }
template <int LPR>
__device__ __forceinline__ void fm_row_stats(const float* row, int n, int sub, float& mean, float& rstd) {
    constexpr int L = FmLnLanes<LPR>::value;
    const int s0 = fm_ln_sub<LPR>(sub);
    float s = 0.f;
    for (int c = s0; c < n; c += L) s += row[c];
    s = fm_group_sum<L>(s);
    const float inv_n = 1.0f / (float)n;          // n is a power of two in every use: s * inv_n == s / n bit for bit
    mean = s * inv_n;
    float q = 0.f;
    for (int c = s0; c < n; c += L) { const float d = row[c] - mean; q = fm_fma(d, d, q); }
    q = fm_group_sum<L>(q);
    rstd = __builtin_amdgcn_rsqf(fm_fma(q, inv_n, 1e-5f));     // v_rsq_f32 (~1 ulp) instead of IEEE 1/sqrt (about 25 VALU less per row lane)
}
__device__ __forceinline__ void fm_row_stats8(const float* row, int n, int sub, float& mean, float& rstd) {
    fm_row_stats<8>(row, n, sub, mean, rstd);
}

// Counter-based noise for sharded sampling (SURVEY.md §8e "performance mode"): Philox4x32-10 keyed by the run's seed, counter =
// (global molecule id, row inside the molecule, step * 4 + modality, draw block).  A molecule's draws depend on nothing else,
// so its trajectory is the same on 1 or 8 GPUs, in any batch composition; no noise tensors exist in HBM.
struct FmPhilox4 { unsigned v[4]; };
__device__ __forceinline__ FmPhilox4 fm_philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    FmPhilox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__device__ __forceinline__ float fm_u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }              // [0, 1), 24 bits like torch.rand
__device__ __forceinline__ float fm_exp1(unsigned x) { return fmaxf(-logf((float)((x >> 8) + 1u) * 5.9604644775390625e-08f), 1e-30f); }   // Exp(1): -log(u), u in (0, 1]; never -0

