// mpx_convw.h -- EXPANDING 1x1 stride-1 conv + BN + residual + ReLU with K = 256 (the last conv of a layer3 bottleneck:
// 256 -> 1024) as ONE persistent 4-wave workgroup per CU whose WEIGHTS LIVE IN REGISTERS (tile id 14; f16x3 arithmetic of
// mpx_conv.h).
//
// What bounds mpx_convx.h (tile 10) on this layer class (tools/ablate_convx.sh, profiles/r04_convx_ablation.txt): the instruction
// stream alone runs 0.57 ms, every memory instruction adds its own cost on top (1.02 ms with all of them), and two thirds of the
// LDS-DMA pieces re-stream the SAME 256 KB of weights through L2 -> LDS for every pixel tile a workgroup walks -- 47 B of operand
// traffic per 1000 MACs against 16 for the 3x3 patch kernel, at the package power limit with the MFMA pipe a third busy.  A
// persistent workgroup keeps its cout tile for the whole launch, and with K = 256 the weights of 64 output channels are
// 64 x 256 x (hi + lo) = 64 KB = the 256 AGPRs of one wave.  Here:
//   * tile 256 (cout) x 64 (pixels), 4 waves (one per SIMD, 512 registers each), wave w owns channels [64 w, 64 w + 64) of the cout
//     tile for ALL pixels: its 8 x 4 x (hi, lo) weight fragments are loaded ONCE per launch, straight from the piece-major planes in
//     MFMA operand layout, and never touch the LDS;
//   * the LDS holds only pixels: two buffers of a whole 64-pixel x 256-channel tile (8 K steps x [X_hi 4 KB | X_lo 4 KB] = 64 KB);
//     a tile is requested a whole tile ahead, between the slices of the epilogue that releases its buffer -- 16 DMA pieces per wave
//     and tile instead of 48 per half-tile, a third of the L2 -> LDS bytes per MAC; the K loop has no barrier, no DMA and no counted
//     wait (one vmcnt(0) and one barrier per TILE);
//   * pixel fragments are single-buffered and re-read column by column: when the 12 MFMAs of a pixel column are issued its two
//     registers quads take the next K step's fragments (36 MFMAs of slack);
//   * epilogue from the accumulator registers as mpx_convx.h (v_permlane16_swap + DPP row_ror:8 regrouping into whole 128-B lines),
//     residual lines requested four K steps before the tile ends (ConvW::RES_STEP: earlier is slower), the 16 stores of a tile in
//     one burst behind the arithmetic.
// Per accumulator the products are summed in tile 10's order (K steps ascending; hi*lo, lo*hi, hi*hi), and the epilogue arithmetic is
// the same: results are bit-identical to tiles 7 and 10.
#pragma once
#include "mpx_conv.h"

namespace mpx {

// The MFMAs are inline assembly so that the weight operand can be CONSTRAINED to the accumulation half of the register file ("a"):
// hipcc keeps MFMA A operands in v0..v255 and parks what does not fit in AGPRs behind v_accvgpr_read copies (four copies and an
// s_nop per MFMA, measured in the first build of this kernel).  With the weights in a0..a255 everything else -- accumulators, pixel
// fragments, residual lines, epilogue temporaries -- shares v0..v255.  The compiler knows nothing about what an asm statement
// executes, so the hazards are handled here: accumulators are only read by VALU code behind mfma_drain() (the K loop's MFMAs have
// left the pipe), consecutive MFMAs never chain through the same accumulator (four apart), fragments are written by ds_read only
// (s_waitcnt lgkmcnt is inserted by the compiler, which sees the asm operands).
__device__ __forceinline__ void mfma_w(f4& d, const h8& w, const h8& x) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_w0(f4& d, const h8& w, const h8& x) {         // first product of a tile: C = 0
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_drain() {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
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

struct ConvW {
    static constexpr int TC = 256, TP = 64, NW = 4, NT = 256;
    static constexpr int K = 256, NK = K / 32;
    static constexpr int STAGE = 8192;                  // one K step of the pixel tile: [X_hi 64 rows x 64 B | X_lo 64 rows x 64 B]
    static constexpr int TILE = NK * STAGE;             // 64 KB
    static constexpr int OFF_SCALE = 2 * TILE;          // f32[256] scale, f32[256] shift of the workgroup's cout tile
    static constexpr int LDS = 2 * TILE + 2048;
    // K step behind whose first pixel column a tile's residual lines are requested.  Measured in the network at batch 2340 (two passes
    // of tools/ab_variants.sh in one call, the 23 layers 256 -> 1024): step 0: 23.9 ms, 2: 21.7, 3: 21.3, 4: 21.25, 5: 21.9, 6: 22.7 --
    // EARLIER is slower although nothing waits for the lines before the epilogue (measured, not explained by a bare copy
    // kernel's behaviour; the order and timing of a wave's memory instructions decide this layer, not their number)
    static constexpr int RES_STEP = 4;
};

template <bool RELU>
__global__ __launch_bounds__(256, 1) void convw_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef ConvW C;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int K = C::K, NK = C::NK;
    constexpr unsigned OOB = 0x80000000u;

    // ---- the tiles of this workgroup: logical ids v0, v0 + G, v0 + 2G, ... (cout tile fastest, so it is the same for all) ----
    const int G = gridDim.x;                                           // a multiple of 8 and of n_tiles_c (host)
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);    // blocks of one XCD walk neighbouring tiles
    const int n0 = (v0 % p.n_tiles_c) * C::TC;
    const int mt0 = v0 / p.n_tiles_c, mt_step = G / p.n_tiles_c;
    const int n_mt = (p.M + C::TP - 1) / C::TP;
    const int my_tiles = mt0 < n_mt ? (n_mt - 1 - mt0) / mt_step + 1 : 0;
    if (my_tiles == 0) return;

    // ---- pixel DMA: wave w moves rows [16 w, 16 w + 16) of every stage, one 1-KiB piece per plane ----------------------------
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
        x_dead = (ti < my_tiles && !(MPX_ABL_LO8 & 128)) ? 0 : (int)OOB;          // past the last tile: the pieces are still issued, but touch no memory
    };
    auto dma_stage = [&](int buf, int ks) {     // this wave's two pieces of stage ks of the tile being fetched
        char* d = smem + buf * C::TILE + ks * C::STAGE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_hi, MPX_LDS_PTR(d), 16, xrow | x_dead, ks * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_lo, MPX_LDS_PTR(d + 4096), 16, xrow | x_dead, ks * 64, 0, 0);
    };

    // ---- the weights of this wave: 8 K steps x 4 cout fragments x (hi, lo), MFMA A-operand layout (row = lane & 15, 16-B chunk =
    //      lane >> 4), read from the piece-major planes (w_packed_index: a fragment is one contiguous, swizzled 1-KiB piece) -------
    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    h8 wh[NK][4], wl[NK][4];
    {
        const size_t piece0 = ((size_t)(n0 >> 4) + wave * 4) * NK;                  // piece index of (fragment 0, K step 0)
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

    // ---- pixel fragments (MFMA B operand: column = lane & 15 of fragment b, 16-B chunk = lane >> 4) ---------------------------
    const int b_off = lrow * 64 + qsw;
    h8 bh[4], bl[4];
    auto read_b = [&](int buf, int ks, int b) {
        const char* s = smem + buf * C::TILE + ks * C::STAGE + b * 1024 + b_off;
        bh[b] = *(const h8*)s;
        bl[b] = *(const h8*)(s + 4096);
    };

    // ---- epilogue state: lane geometry of the regrouped 16-B chunks (as mpx_convx.h, wave tile 64 cout x 64 pixels) -----------
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
    f4 acc[4][4];
    u4 so_h[4][2], so_l[4][2];                  // a tile's output chunks, stored in one burst at the end of its epilogue
    float one = 1.0f;                           // opaque to the optimiser (fma(x, one, y) must stay an fma for v_fma_mix_f32)
    asm volatile("" : "+v"(one));
    auto ror8 = [](float old, float src, auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), 0x128, 0xf, MASK, false));
    };
    auto issue_epilogue_loads = [&](int ti) {   // 16 loads: the residual lines of tile ti
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = p.r_hi ? (rem > 0x7fffffffLL ? 0x7fffffff : (int)rem) : 0;          // no residual: every lane out of range
        const half_t* rbh = p.r_hi ? p.r_hi : p.y_hi;
        const half_t* rbl = p.r_hi ? p.r_lo : p.y_lo;
        const __amdgpu_buffer_rsrc_t r_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbh + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(rbl + (size_t)m0 * p.cout), 0, rec, 0x00020000);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rh[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_hi_rs, (offA[b] + k * row8) | ((MPX_ABL_LO8 & 128) ? (int)OOB : 0), 0, 2);
                if (MPX_ABL_LO8 & 1) {          // timing-only: 8 B of the lo plane per lane
                    const auto t = __builtin_amdgcn_raw_buffer_load_b64(r_lo_rs, (offA[b] + k * row8) >> 1, 0, 2);
                    rl[b][k] = u4{t[0], t[1], 0u, 0u};
                } else
                rl[b][k] = __builtin_amdgcn_raw_buffer_load_b128(r_lo_rs, (offA[b] + k * row8) | ((((MPX_ABL_LO8 & 16) && k) || (MPX_ABL_LO8 & 128)) ? (int)OOB : 0), 0, 2);
            }
    };
    auto epilogue = [&](int ti, auto&& after_slice) {               // 16 stores
        const int m0 = (mt0 + ti * mt_step) * C::TP;
        const long long rem = ((long long)p.M - m0) * p.cout * 2;
        const int rec = rem > 0x7fffffffLL ? 0x7fffffff : (int)rem;
        const __amdgpu_buffer_rsrc_t y_hi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_lo_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + (size_t)m0 * p.cout), 0, rec, 0x00020000);
        f4 sc[2][2], sh[2][2];                  // scale / shift of this lane's channels in the accumulator layout, from LDS
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = wave * 64 + (2 * q + (erow & 1)) * 16 + (erow >> 1) * 8;
            sc[q][0] = *(const f4*)(smem + C::OFF_SCALE + col * 4);
            sc[q][1] = *(const f4*)(smem + C::OFF_SCALE + col * 4 + 16);
            sh[q][0] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4);
            sh[q][1] = *(const f4*)(smem + C::OFF_SCALE + 1024 + col * 4 + 16);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (MPX_ABL_LO8 & 512) {            // timing-only: no epilogue arithmetic
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        so_h[b][k][q] = __float_as_uint(acc[k][b][q]) ^ rh[b][k][q];
                        so_l[b][k][q] = __float_as_uint(acc[2 + k][b][q]) ^ rl[b][k][q];
                    }
                after_slice(b);
                continue;
            }
            float ve[8], vo[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = acc[0][b][j], e1 = acc[1][b][j], o0 = acc[2][b][j], o1 = acc[3][b][j];
                const auto se = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                const auto so = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                ve[j] = __uint_as_float((unsigned)se[0]) * sc[0][0][j] + sh[0][0][j];
                ve[4 + j] = __uint_as_float((unsigned)se[1]) * sc[0][1][j] + sh[0][1][j];
                vo[j] = __uint_as_float((unsigned)so[0]) * sc[1][0][j] + sh[1][0][j];
                vo[4 + j] = __uint_as_float((unsigned)so[1]) * sc[1][1][j] + sh[1][1][j];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = k == 0 ? ror8(ve[j], vo[j], std::integral_constant<int, 0xC>{}) : ror8(vo[j], ve[j], std::integral_constant<int, 0x3>{});
                // + (hi + lo) of the residual line (zeros when the layer has no residual), ReLU, split into hi + lo (cw_* above)
                u4 oh, ol;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r0 = v[2 * q] + cw_pair_lo(rh[b][k][q], one, rl[b][k][q]);
                    float r1 = v[2 * q + 1] + cw_pair_hi(rh[b][k][q], one, rl[b][k][q]);
                    if (RELU) {
                        r0 = cw_relu(r0);
                        r1 = cw_relu(r1);
                    }
                    oh[q] = cw_pack_hi(r0, r1);
                    ol[q] = cw_pack_lo(oh[q], one, r0, r1);
                }
                so_h[b][k] = oh;
                so_l[b][k] = ol;
            }
            after_slice(b);
        }
        // all 16 stores in one burst behind the arithmetic (and behind the pieces of the tile after next, which the slices carry):
        // 21.3 -> 21.2 ms for the 23 layers against stores issued slice by slice; pieces BEHIND the stores: 25.2 ms (they retire in order)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                __builtin_amdgcn_raw_buffer_store_b128(so_h[b][k], y_hi_rs, (offA[b] + k * row8) | ((MPX_ABL_LO8 & 128) ? (int)OOB : 0), 0, 2);
                if (MPX_ABL_LO8 & 1) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(u2{so_l[b][k][0], so_l[b][k][1]}, y_lo_rs, (offA[b] + k * row8) >> 1, 0, 2);
                } else
                __builtin_amdgcn_raw_buffer_store_b128(so_l[b][k], y_lo_rs, (offA[b] + k * row8) | ((((MPX_ABL_LO8 & 16) && k) || (MPX_ABL_LO8 & 128)) ? (int)OOB : 0), 0, 2);
            }
    };

    // scale / shift of the cout tile into LDS (the prologue's barrier publishes them)
    if (tid < 128) {
        const float* src = tid < 64 ? p.scale + n0 + tid * 4 : p.shift + n0 + (tid - 64) * 4;
        *(f4*)(smem + C::OFF_SCALE + tid * 16) = *(const f4*)src;
    }
    // ---- prologue: the first TWO pixel tiles into the two buffers (the weight loads above are in flight next to them) -----------
    set_x_desc(0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) dma_stage(0, ks);
    set_x_desc(1);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) dma_stage(1, ks);
    // (the builtin, not an asm statement: hipcc's own wait insertion then knows that the weight loads have returned -- behind an asm
    // wait it guards the first use of every weight register in the tile loop with a vmcnt(N) that the loop's own loads run into)
    __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 4; ++b) read_b(0, 0, b);

    int buf = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int nbuf = buf ^ 1;
        set_x_desc(ti + 2);                     // requested under this tile's epilogue, into this tile's buffer
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                // the 12 MFMAs of pixel column b: hi*lo, lo*hi, hi*hi over the four cout fragments
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (ks == 0) mfma_w0(acc[a][b], wh[ks][a], bl[b]);
                    else mfma_w(acc[a][b], wh[ks][a], bl[b]);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) mfma_w(acc[a][b], wl[ks][a], bh[b]);
#pragma unroll
                for (int a = 0; a < 4; ++a) mfma_w(acc[a][b], wh[ks][a], bh[b]);
                __builtin_amdgcn_sched_barrier(0);
                // column b is free: the next K step's fragments (36 MFMAs ahead of their first use)
                if (ks + 1 < NK) {
                    read_b(buf, ks + 1, b);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ks == C::RES_STEP && b == 0) {
                    issue_epilogue_loads(ti);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // Loads, LDS-DMAs and stores retire in issue order: behind vmcnt(0) this tile's residual lines are here and this wave's pieces
        // of the next tile (requested a whole tile ago) have landed; behind the barrier all four waves' pieces have, and nobody reads
        // this tile's buffer again.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mfma_drain();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // the first fragments of the next tile travel under the epilogue (past the last tile: stale bytes, unused) ...
#pragma unroll
        for (int b = 0; b < 4; ++b) read_b(nbuf, 0, b);
        __builtin_amdgcn_sched_barrier(0);
        // ... and the tile after next is requested between its four slices, into the buffer that has just been released
        epilogue(ti, [&](int b) {
            __builtin_amdgcn_sched_barrier(0);
            dma_stage(buf, 2 * b);
            dma_stage(buf, 2 * b + 1);
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
#endif
}

}  // namespace mpx
