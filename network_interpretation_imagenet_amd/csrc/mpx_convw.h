// mpx_convw.h -- EXPANDING 1x1 stride-1 conv + BN + residual + ReLU with K = 256 (the last conv of a layer3 bottleneck:
// 256 -> 1024) as ONE persistent 4-wave workgroup per CU whose WEIGHTS LIVE IN REGISTERS (tile id 14; f16x3 arithmetic of
// mpx_conv.h), with a COLUMN-MAJOR K loop and the epilogue BETWEEN its MFMAs (round 5).
//
// Round 4: what bounds mpx_convx.h (tile 10) on this layer class is that two thirds of its LDS-DMA pieces re-stream the
// SAME 256 KB of weights through L2 -> LDS for every pixel tile a workgroup walks.  A persistent workgroup keeps its cout tile for the
// whole launch, and with K = 256 the weights of 64 output channels are 64 x 256 x (hi + lo) = 64 KB = the 256 AGPRs of one wave:
//   * tile 256 (cout) x 64 (pixels), 4 waves (one per SIMD, 512 registers each), wave w owns channels [64 w, 64 w + 64) of the cout
//     tile for ALL pixels: its 8 x 4 x (hi, lo) weight fragments are loaded ONCE per launch, straight from the piece-major planes in
//     MFMA operand layout, and never touch the LDS;
//   * the LDS holds only pixels: two buffers of a whole 64-pixel x 256-channel tile (8 K steps x [X_hi 4 KB | X_lo 4 KB] = 64 KB);
//   * epilogue from the accumulator registers as mpx_convx.h (v_permlane16_swap + DPP row_ror:8 regrouping into whole 128-B lines).
//
// Round 5 (profiles/r05_convw_instruction_stream_and_btail_dead_columns.txt, profiles/r05_convwc_variants.txt): that kernel
// ran a tile as a K loop of 384 MFMAs (6.1 k cycles) and THEN an epilogue of ~ 500 VALU + 32 memory instructions with no MFMA beside it;
// its instruction stream alone was 73 % of the layer's time, and so was its memory traffic alone.  The weights being in registers, nothing
// forces the K loop to sweep all four pixel columns per K step: here it finishes column b (8 K steps x 12 MFMAs into acc[.][b]) before it
// starts column b + 1, and the epilogue slice of column b -- regrouping, BatchNorm, residual add, ReLU, hi / lo split, 4 stores -- is
// issued one or two instructions at a time in the gaps behind the 96 MFMAs of column b + 1 (an MFMA holds the matrix pipe 16 cycles and
// the wave's issue for 8 of them).  Slice 3 of a tile runs under column 0 of the next one.  No second accumulator set: a finished
// column's registers are next written when the NEXT tile reaches that column.  Per accumulator the products are summed in the same order
// (K steps ascending; hi*lo, lo*hi, hi*hi) and the epilogue computes the same values: bit-identical to round 4's kernel and to tiles 7
// and 10.  In the network at batch 2340: 21.3 -> 18.7 ms for the 23 layers (5.2 TB/s: the rate of a bare copy kernel).
//
// Memory instructions of a tile, in program order (all retire in issue order, so every wait is a counted immediate; "gap T" = the slot
// behind MFMA number T of a column; ConvWC below holds the measured placement):
//   column b, gaps 8..14   b = 0, 1 only: 8 of the 16 LDS-DMA pieces of the NEXT tile, into the buffer the previous tile released
//   column b, gaps 32..35  the 4 residual loads of slice b + 1 (column 3: slice 0 of the next tile), 1.7 columns before their use --
//                          EARLIER is slower, as in round 4's kernel (at gap 0: 20.1 ms; 32: 18.8; 48: 18.8; 80: 19.3)
//   column b, gaps 88, 89  the 2 + 2 stores of slice b - 1 (its values are complete behind gaps 39 and 64)
//   tile end               s_waitcnt for the next tile's pieces (the younger loads and stores may stay in flight), s_barrier
// hipcc's own wait insertion places the vmcnt in front of the first use of a residual line (it sees every load, piece and store of the
// loop; checked in the ISA: vmcnt(26) / vmcnt(34)): no manual wait there.
#pragma once
#include <type_traits>
#include "mpx_conv.h"

// Placement of a tile's memory instructions (gap = the slot behind MFMA number T of a column); the defaults are the measured best
// (tools/ab_variants.sh with -DCWC_*=..., profiles/r05_convwc_variants.txt)
#ifndef CWC_DMA_COL
#define CWC_DMA_COL 0       // first column that requests pieces of the next tile
#endif
#ifndef CWC_DMA_NCOL
#define CWC_DMA_NCOL 2      // columns that share the 16 pieces (1: all in one column, 2: 8 each)
#endif
#ifndef CWC_DMA_GAP
#define CWC_DMA_GAP 8       // gap of a column's first piece pair; one pair every second gap
#endif
#ifndef CWC_RES_GAP
#define CWC_RES_GAP 32      // gaps CWC_RES_GAP .. + 3: the four residual loads a column issues
#endif
#ifndef CWC_RES_LEAD
#define CWC_RES_LEAD 1      // column b requests the lines of slice b + CWC_RES_LEAD (consumed under column b + CWC_RES_LEAD + 1)
#endif
#ifndef CWC_ST_GAP0
#define CWC_ST_GAP0 88     // stores of a slice's first half (its values are complete behind gap 39) ...
#endif
#ifndef CWC_ST_GAP1
#define CWC_ST_GAP1 89     // ... and of its second half (complete behind gap 64)
#endif

namespace mpx {

// The MFMAs are inline assembly so that the weight operand can be CONSTRAINED to the accumulation half of the register file ("a"):
// hipcc keeps MFMA A operands in v0..v255 and parks what does not fit in AGPRs behind v_accvgpr_read copies (four copies and an
// s_nop per MFMA, measured in the first build of this kernel).  With the weights in a0..a255 everything else -- accumulators, pixel
// fragments, residual lines, epilogue temporaries -- shares v0..v255.  The compiler knows nothing about what an asm statement
// executes, so the hazards are handled here: a column's accumulators are first read by VALU code behind MFMA number 4 of the NEXT column
// (five MFMAs = 80 cycles after the last one that wrote them), consecutive MFMAs never chain through the same accumulator (four apart),
// fragments are written by ds_read only (s_waitcnt lgkmcnt is inserted by the compiler, which sees the asm operands).
__device__ __forceinline__ void mfma_w(f4& d, const h8& w, const h8& x) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_w0(f4& d, const h8& w, const h8& x) {         // first product of a tile: C = 0
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "a"(w), "v"(x));
}

// Epilogue arithmetic of a 1-wave-per-SIMD kernel, where every VALU instruction is exposed: the forms hipcc does not pick by itself
// (its SLP pass packs these chains into v_pk_fma_f32 / v_pk_add_f32 behind separate conversions).  Same values, same bits:
//   hi + lo of a residual element is exact in fp32, so ONE v_fma_mix_f32 (both fp16 -> fp32 conversions ride in the instruction) gives
//   what (float)hi + (float)lo gives; v - (float)fp16(v) is exact, so v_fma_mixlo/hi_f16 round the same number split_f32 rounds.
__device__ __forceinline__ float cw_pair_lo(unsigned h, float one, unsigned l) {      // (float)lo16(h) + (float)lo16(l)
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(one), "v"(l));
    return d;
}
__device__ __forceinline__ float cw_pair_hi(unsigned h, float one, unsigned l) {      // (float)hi16(h) + (float)hi16(l)
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(one), "v"(l));
    return d;
}
__device__ __forceinline__ float cw_relu(float v) {
    float d;
    asm("v_max_f32 %0, 0, %1" : "=v"(d) : "v"(v));
    return d;
}
__device__ __forceinline__ unsigned cw_pack_hi(float a, float b) {                    // {fp16(a), fp16(b)}
    unsigned d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned cw_pack_lo(unsigned hi, float one, float a, float b) {   // {fp16(a - lo16(hi)), fp16(b - hi16(hi))}
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -%2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -%2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d) : "v"(hi), "v"(one), "v"(a), "v"(b));
    return d;
}

template <int KK>
struct ConvWK {
    static_assert(KK == 256, "64 output channels x K x (hi + lo) fill a wave's 256 AGPRs.  K = 128 was built and measured in round 5 (a tie with convx, DESIGN 5e) and is NOT "
                             "shipped: with NK = 4 an epilogue stage starts 3 MFMAs behind the column's last accumulator write, not the 5 the hazard comment in the kernel relies on");
    static constexpr int TC = 256, TP = 64, NW = 4, NT = 256;
    static constexpr int K = KK, NK = K / 32;
    static constexpr int STAGE = 8192;                  // one K step of the pixel tile: [X_hi 64 rows x 64 B | X_lo 64 rows x 64 B]
    static constexpr int TILE = NK * STAGE;             // 64 KB (K = 256), 32 KB (K = 128)
    static constexpr int OFF_SCALE = 2 * TILE;          // f32[256] scale, f32[256] shift of the workgroup's cout tile
    static constexpr int LDS = 2 * TILE + 2048;
};
typedef ConvWK<256> ConvW;

// Where a tile's memory instructions sit.  A column has NK * 12 MFMAs; the epilogue slice is written as 96 STAGES of one or two instructions
// (K = 256: one stage per gap; K = 128: two), and every placement below is a stage number.
template <int NK>
struct ConvWC {
    static constexpr int STAGES_PER_GAP = 96 / (NK * 12);
    static constexpr int DMA_COL = CWC_DMA_COL, DMA_NCOL = CWC_DMA_NCOL, DMA_GAP = CWC_DMA_GAP, RES_GAP = CWC_RES_GAP, RES_LEAD = CWC_RES_LEAD;
    static constexpr int ST_GAP0 = CWC_ST_GAP0, ST_GAP1 = CWC_ST_GAP1;
    static constexpr int PAIRS_PER_COL = NK / DMA_NCOL;                          // piece pairs (one K step of the tile) a requesting column issues
    static constexpr int DMA_LAST_COL = DMA_COL + DMA_NCOL - 1, DMA_LAST_GAP = DMA_GAP + 2 * (PAIRS_PER_COL - 1);
    // memory instructions issued behind a tile's last piece: the rest of that column, then whole columns (4 loads + 4 stores each)
    static constexpr int TILE_END_WAIT = (RES_GAP > DMA_LAST_GAP ? 4 : 0) + (ST_GAP0 > DMA_LAST_GAP ? 2 : 0) + (ST_GAP1 > DMA_LAST_GAP ? 2 : 0) + (3 - DMA_LAST_COL) * 8;
    static_assert(STAGES_PER_GAP * NK * 12 == 96, "whole stages per gap");
    static_assert(DMA_NCOL == 1 || DMA_NCOL == 2, "the piece pairs of a tile over one or two columns");
    static_assert(DMA_LAST_COL <= 3 && DMA_LAST_GAP < 96 && RES_GAP + 3 < 96, "placements inside a column");
    static_assert(ST_GAP0 >= 40 && ST_GAP1 >= 65 && ST_GAP1 < 96 && ST_GAP0 <= ST_GAP1, "a half-slice is stored once its last value exists");
    static_assert(RES_LEAD >= 0 && RES_LEAD <= 2, "residual lines are requested one to three columns before their use");
    static_assert(TILE_END_WAIT <= 63, "vmcnt is six bits");
};

template <int N, class F>
__device__ __forceinline__ void cw_static_for(F&& f) {          // f(integral_constant<0>), ..., f(integral_constant<N - 1>), in order
    if constexpr (N > 0) {
        cw_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int KK, bool RELU>
__global__ __launch_bounds__(256, 1) void convw_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef ConvWK<KK> C;
    typedef ConvWC<KK / 32> W;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int K = C::K, NK = C::NK;
    constexpr unsigned OOB = 0x80000000u;

    // ---- the tiles of this workgroup (as mpx_convw.h) ----
    const int G = gridDim.x;
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int n0 = (v0 % p.n_tiles_c) * C::TC;
    const int mt0 = v0 / p.n_tiles_c, mt_step = G / p.n_tiles_c;
    const int n_mt = (p.M + C::TP - 1) / C::TP;
    const int my_tiles = mt0 < n_mt ? (n_mt - 1 - mt0) / mt_step + 1 : 0;
    if (my_tiles == 0) return;

    // ---- pixel DMA: wave w moves rows [16 w, 16 w + 16) of every stage, one 1-KiB piece per plane ----
    const int prow = lane >> 2;
    const int xrow = ((wave * 16 + prow) * K) * 2 + ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 16;
    __amdgpu_buffer_rsrc_t x_hi, x_lo;          // descriptors of the tile being FETCHED
    int x_dead = 0;
    auto set_x_desc = [&](int ti) {
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * K * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (rem < 0 ? 0 : (int)rem);
        x_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)m0 * K), 0, rec, 0x00020000);
        x_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)m0 * K), 0, rec, 0x00020000);
        x_dead = ti < my_tiles ? 0 : (int)OOB;          // past the last tile: the pieces are still issued, but touch no memory
    };
    auto dma_stage = [&](int buf, int ks) {     // this wave's two pieces of stage ks of the tile being fetched
        char* d = smem + buf * C::TILE + ks * C::STAGE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_hi, MPX_LDS_PTR(d), 16, xrow | x_dead, ks * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_lo, MPX_LDS_PTR(d + 4096), 16, xrow | x_dead, ks * 64, 0, 0);
    };

    // ---- the weights of this wave: 8 K steps x 4 cout fragments x (hi, lo), MFMA A-operand layout, from the piece-major planes ----
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    h8 wh[NK][4], wl[NK][4];
    {
        const size_t piece0 = ((size_t)(n0 >> 4) + wave * 4) * NK;
        const char* bh = (const char*)p.w_hi + piece0 * 1024 + lrow * 64 + qsw;
        const char* bl = (const char*)p.w_lo + piece0 * 1024 + lrow * 64 + qsw;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                wh[ks][a] = *(const h8*)(bh + (a * NK + ks) * 1024);
                wl[ks][a] = *(const h8*)(bl + (a * NK + ks) * 1024);
            }
    }

    // ---- pixel fragments of ONE column, double-buffered over the K steps (MFMA B operand: column = lane & 15, 16-B chunk = lane >> 4) ----
    const int b_off = lrow * 64 + qsw;
    h8 fh[2], fl[2];
    auto read_f = [&](int par, int buf, int ks, int b) {
        const char* s = smem + buf * C::TILE + ks * C::STAGE + b * 1024 + b_off;
        fh[par] = *(const h8*)s;
        fl[par] = *(const h8*)(s + 4096);
    };

    // ---- epilogue geometry (as mpx_convw.h: wave tile 64 cout x 64 pixels, a lane ends up with 8 consecutive channels of a pixel) ----
    const int erow = lane >> 4;
    const bool lo8 = (lane & 8) == 0;
    int offA[4];
    {
        const int co = n0 + wave * 64 + (2 * (lo8 ? 0 : 1) + (erow & 1)) * 16 + (erow >> 1) * 8;
        const int dead = (p.cout - 1 - co) & (int)OOB;
#pragma unroll
        for (int b = 0; b < 4; ++b) offA[b] = ((b * 16 + (lane & 7)) * p.cout + co) * 2 | dead;
    }
    const int row8 = 8 * p.cout * 2;
    u4 rh[4][2], rl[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) rh[b][k] = rl[b][k] = u4{0u, 0u, 0u, 0u};
    f4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    float one = 1.0f;                           // opaque to the optimiser (fma(x, one, y) must stay an fma for v_fma_mix_f32)
    asm volatile("" : "+v"(one));
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    // descriptors of the residual / output planes of tile ti (ti outside [0, my_tiles): zero records, every access out of range)
    auto plane_desc = [&](const half_t* base, int ti) {
        const bool live = ti >= 0 && ti < my_tiles && base != nullptr;
        const int m0 = live ? (mt0 + ti * mt_step) * C::TP : 0;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = live ? (rem > 0x7fffffffLL ? 0x7fffffff : (int)rem) : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)((base ? base : p.y_hi) + (size_t)m0 * p.cout), 0, rec, 0x00020000);
    };

    // scale / shift of the cout tile into LDS (the prologue's barrier publishes them), then into registers for the whole launch
    if (tid < 128) {
        const float* src = tid < 64 ? p.scale + n0 + tid * 4 : p.shift + n0 + (tid - 64) * 4;
        *(f4*)(smem + C::OFF_SCALE + tid * 16) = *(const f4*)src;
    }
    // ---- prologue: the first pixel tile, the residual lines of its slice 0 ----
    set_x_desc(0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) dma_stage(0, ks);
    {
        const __amdgpu_buffer_rsrc_t r_hi_rs = plane_desc(p.r_hi, 0), r_lo_rs = plane_desc(p.r_hi ? p.r_lo : nullptr, 0);
#pragma unroll
        for (int s0 = 0; s0 < W::RES_LEAD; ++s0)        // the slices whose request would have fallen into the columns before tile 0
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rh[s0][k] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, offA[s0] + k * row8, 0, 2);
                rl[s0][k] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, offA[s0] + k * row8, 0, 2);
            }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0): weights, pieces and lines are here (the builtin: hipcc's wait insertion sees it)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    f4 sc[2][2], sh[2][2];                      // scale / shift of this lane's channels in the accumulator layout
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int col = wave * 64 + (2 * q + (erow & 1)) * 16 + (erow >> 1) * 8;
        sc[q][0] = *(const f4*)(smem + C::OFF_SCALE + col * 4);
        sc[q][1] = *(const f4*)(smem + C::OFF_SCALE + col * 4 + 16);
        sh[q][0] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4);
        sh[q][1] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4 + 16);
    }
    read_f(0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);

    float ve[8], vo[8], v[8], r0 = 0.f, r1 = 0.f;
    u4 oh[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, ol[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};     // a half-slice's output chunks, until its stores are issued
#pragma unroll
    for (int j = 0; j < 8; ++j) ve[j] = vo[j] = v[j] = 0.f;

    // One column: the 96 MFMAs of pixel column B of tile ti, and in the gap behind MFMA number T the piece `T` of the epilogue slice
    // of the PREVIOUS column (SP = B - 1 of this tile; for B = 0 slice 3 of tile ti - 1) plus this column's share of the memory traffic.
    auto column = [&](auto b_tag, int ti, int buf) {
        constexpr int B = decltype(b_tag)::value;
        constexpr int SP = (B + 3) & 3, SN = (B + W::RES_LEAD) & 3;
        constexpr bool SN_NEXT_TILE = B + W::RES_LEAD > 3;
        const __amdgpu_buffer_rsrc_t y_hi_rs = plane_desc(p.y_hi, B == 0 ? ti - 1 : ti), y_lo_rs = plane_desc(p.y_lo, B == 0 ? ti - 1 : ti);
        const __amdgpu_buffer_rsrc_t r_hi_rs = plane_desc(p.r_hi, SN_NEXT_TILE ? ti + 1 : ti);
        const __amdgpu_buffer_rsrc_t r_lo_rs = plane_desc(p.r_hi ? p.r_lo : nullptr, SN_NEXT_TILE ? ti + 1 : ti);
        auto stage = [&](auto t_tag) {
            constexpr int T = decltype(t_tag)::value;
            // ---- memory traffic of this column ----
            if constexpr (T >= W::RES_GAP && T < W::RES_GAP + 4) {        // residual lines of slice SN: (k, plane) = (R >> 1, R & 1)
                constexpr int R = T - W::RES_GAP;
                if constexpr ((R & 1) == 0) rh[SN][R >> 1] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, offA[SN] + (R >> 1) * row8, 0, 2);
                else rl[SN][R >> 1] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, offA[SN] + (R >> 1) * row8, 0, 2);
            }
            if constexpr (B >= W::DMA_COL && B <= W::DMA_LAST_COL && T >= W::DMA_GAP && T <= W::DMA_LAST_GAP && ((T - W::DMA_GAP) & 1) == 0)
                dma_stage(buf ^ 1, (B - W::DMA_COL) * W::PAIRS_PER_COL + ((T - W::DMA_GAP) >> 1));
            // ---- the epilogue slice SP, one or two instructions per gap (the first MFMAs of the column cover the distance to the last
            //      MFMA that wrote acc[.][SP]: the asm statements hide that hazard from hipcc) ----
            if constexpr (T >= 4 && T < 8) {
                constexpr int j = T - 4;
                const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][SP][j]), __float_as_uint(acc[1][SP][j]), false, false);
                const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2][SP][j]), __float_as_uint(acc[3][SP][j]), false, false);
                ve[j] = __uint_as_float((unsigned)se[0]);
                ve[4 + j] = __uint_as_float((unsigned)se[1]);
                vo[j] = __uint_as_float((unsigned)so[0]);
                vo[4 + j] = __uint_as_float((unsigned)so[1]);
            }
            if constexpr (T >= 8 && T < 16) {   // acc * scale + shift, two elements per gap: ve[0..7] then vo[0..7]
                constexpr int e = 2 * (T - 8);
                if constexpr (e < 8) {
                    ve[e] = ve[e] * sc[0][e >> 2][e & 3] + sh[0][e >> 2][e & 3];
                    ve[e + 1] = ve[e + 1] * sc[0][(e + 1) >> 2][(e + 1) & 3] + sh[0][(e + 1) >> 2][(e + 1) & 3];
                } else {
                    constexpr int x = e - 8;
                    vo[x] = vo[x] * sc[1][x >> 2][x & 3] + sh[1][x >> 2][x & 3];
                    vo[x + 1] = vo[x + 1] * sc[1][(x + 1) >> 2][(x + 1) & 3] + sh[1][(x + 1) >> 2][(x + 1) & 3];
                }
            }
            auto dpp = [&](auto k_tag, auto u_tag) {     // v[2u], v[2u + 1] of half-slice k
                constexpr int k = decltype(k_tag)::value, u = decltype(u_tag)::value;
                if constexpr (k == 0) {
                    v[2 * u] = ror8(ve[2 * u], vo[2 * u], std::integral_constant<int, 0xC>{});
                    v[2 * u + 1] = ror8(ve[2 * u + 1], vo[2 * u + 1], std::integral_constant<int, 0xC>{});
                } else {
                    v[2 * u] = ror8(vo[2 * u], ve[2 * u], std::integral_constant<int, 0x3>{});
                    v[2 * u + 1] = ror8(vo[2 * u + 1], ve[2 * u + 1], std::integral_constant<int, 0x3>{});
                }
            };
            auto qop = [&](auto k_tag, auto s_tag) {     // s = 5 q + r: the five pieces of output pair q of half-slice k
                constexpr int k = decltype(k_tag)::value, s = decltype(s_tag)::value;
                constexpr int q = s / 5, r = s % 5;
                if constexpr (r == 0) r0 = v[2 * q] + cw_pair_lo(rh[SP][k][q], one, rl[SP][k][q]);
                else if constexpr (r == 1) r1 = v[2 * q + 1] + cw_pair_hi(rh[SP][k][q], one, rl[SP][k][q]);
                else if constexpr (r == 2) {
                    if (RELU) {
                        r0 = cw_relu(r0);
                        r1 = cw_relu(r1);
                    }
                } else if constexpr (r == 3) oh[k][q] = cw_pack_hi(r0, r1);
                else ol[k][q] = cw_pack_lo(oh[k][q], one, r0, r1);
            };
            auto store = [&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                __builtin_amdgcn_raw_buffer_store_b128(oh[k], y_hi_rs, offA[SP] + k * row8, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(ol[k], y_lo_rs, offA[SP] + k * row8, 0, 2);
            };
            if constexpr (T >= 16 && T < 20) dpp(std::integral_constant<int, 0>{}, std::integral_constant<int, (T >= 16 && T < 20) ? T - 16 : 0>{});
            // (no manual wait for the residual lines: hipcc's own wait insertion sees the loads, the pieces and the stores of the whole loop and places
            //  the counted vmcnt in front of the first use of each line; every memory instruction is issued on every pass -- dead ones with an
            //  out-of-range offset -- so the counts hold from the first tile on)
            if constexpr (T >= 20 && T < 40) qop(std::integral_constant<int, 0>{}, std::integral_constant<int, (T >= 20 && T < 40) ? T - 20 : 0>{});
            if constexpr (T == W::ST_GAP0) store(std::integral_constant<int, 0>{});
            if constexpr (T >= 41 && T < 45) dpp(std::integral_constant<int, 1>{}, std::integral_constant<int, (T >= 41 && T < 45) ? T - 41 : 0>{});
            if constexpr (T >= 45 && T < 65) qop(std::integral_constant<int, 1>{}, std::integral_constant<int, (T >= 45 && T < 65) ? T - 45 : 0>{});
            if constexpr (T == W::ST_GAP1) store(std::integral_constant<int, 1>{});
        };
        auto kstep = [&](auto ks_tag) {
            constexpr int ks = decltype(ks_tag)::value;
            constexpr int par = ks & 1;
            auto m = [&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                constexpr int a = i & 3, term = i >> 2;
                if constexpr (term == 0) {
                    if constexpr (ks == 0) mfma_w0(acc[a][B], wh[ks][a], fl[par]);
                    else mfma_w(acc[a][B], wh[ks][a], fl[par]);
                } else if constexpr (term == 1) mfma_w(acc[a][B], wl[ks][a], fh[par]);
                else mfma_w(acc[a][B], wh[ks][a], fh[par]);
                __builtin_amdgcn_sched_barrier(0);
                cw_static_for<W::STAGES_PER_GAP>([&](auto r_tag) { stage(std::integral_constant<int, (ks * 12 + i) * W::STAGES_PER_GAP + decltype(r_tag)::value>{}); });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i == 1) {         // the next step's fragments: the other register pair, 10 MFMAs ahead of their first use
                    if constexpr (ks + 1 < NK) read_f(par ^ 1, buf, ks + 1, B);
                    else if constexpr (B < 3) read_f(par ^ 1, buf, 0, B + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            cw_static_for<12>(m);
        };
        cw_static_for<NK>(kstep);
    };

    int buf = 0;
    for (int ti = 0; ti <= my_tiles; ++ti) {
        set_x_desc(ti + 1);                     // fetched under this tile's columns 0 and 1, into the buffer the previous tile released
        __builtin_amdgcn_sched_barrier(0);
        column(std::integral_constant<int, 0>{}, ti, buf);      // + slice 3 of tile ti - 1
        if (ti == my_tiles) break;              // (the pass behind the last tile only finishes its slice 3: its MFMAs ran on stale pixels, nothing of them is stored)
        column(std::integral_constant<int, 1>{}, ti, buf);
        column(std::integral_constant<int, 2>{}, ti, buf);
        column(std::integral_constant<int, 3>{}, ti, buf);
        // The next tile's 16 pieces were issued in columns 0 and 1; behind the last of them: column 1's residual loads (4) and stores (4),
        // then columns 2 and 3 (4 loads + 4 stores each) = 24 = TILE_END_WAIT with the default placements (the ISA test counts them:
        // tests/test_host_logic.py test_convw_k_loop_is_what_the_source_says).  Behind the barrier all four waves' pieces have landed and
        // nobody reads this tile's buffer again.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W::TILE_END_WAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
        read_f(0, buf, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the last slice's stores leave with the wave
#endif
}

}  // namespace mpx
