// mpx_btail.h -- the TAIL of a 64-channel bottleneck block in ONE launch (f16x3 arithmetic of mpx_conv.h):
//
//     t2  = relu(bn2(conv2_3x3(t1)))                              64 -> 64, stride 1, pad 1      (never leaves the registers)
//     out = relu(bn3(conv3_1x1(t2)) + identity)                   64 -> 256                      (written: the next block's identity)
//           DUAL (first block of layer1): identity = bn_d(conv_d_1x1(x)) K-concatenated as in mpx_conv.h's DUAL kernel
//     t1' = relu(bn1'(conv1'_1x1(out)))                           256 -> C1 (64, or 128 for layer2.0.conv1)
//                                                                 the NEXT block's first conv, from the tile that is still on chip
//
// Why: layer1 of the bottleneck ResNets is HBM-bound (56x56 maps; 24 % of the network's bytes for 3.5 % of its FLOPs).  Layer by
// layer a block moves, per pixel and in units of 64 channels x 4 B: conv1 r4 w1, conv2 r1 w1, conv3 r1 + r4 (identity) + w4 = 16.
// Here: t1 tile with a one-pixel halo r1.4, identity r4, out w4, t1' w1 = 10.4 -- t2 is never stored, and the 256-channel trunk is
// read ONCE per block (as the identity) instead of twice.  Everything behind conv2 is pointwise, so only the 64-channel t1 needs a
// halo; nothing is recomputed.
//
// Workgroup = 4 waves, 78 KB LDS, two workgroups per CU (one's memory- and VALU-heavy chunk epilogues overlap the other's MFMAs),
// persistent over tiles.  Tile = 8 rows x 14 columns of one image (56 = 7 x 8 = 4 x 14); a 16-lane pixel fragment is one tile row
// (lanes 14, 15 are dead: 12.5 % of the MFMA columns, nothing else).  Waves split the PIXELS (wave w = tile rows 2w, 2w+1 = two
// fragments) and own all channels of them, so a wave's conv2 accumulators (D layout: lane = pixel, registers = channels) ARE the
// B operand of its conv3 MFMAs under a permutation of K that is baked into a copy of the packed conv3 weights (mpx_api.hip, upload_tail_planes):
// K position 8g + j of a 32-wide step holds channel (j>>2)*16 + 4g + (j&3).
//
// LDS: [scale / shift vectors][t1 patch: 10 x 16 pixels x 64 channels x (hi, lo) = 40 KB; after conv2 the same bytes are four
// wave-private 8-KB staging areas][weight ring: 4 stages of 64 rows x 32 k x (hi, lo) = 8 KB].  The weight stream of a tile is
// 18 stages of conv2 (tap-major), then per 64-channel chunk of `out`: S3 stages of conv3 rows [64c, 64c+64) and S1 stages of
// conv1' columns [64c, 64c+64); stage numbers run on across tiles (the stream is the same for every tile).
// Per chunk: conv3 MFMAs -> acc * scale + shift as fp32 into the wave's staging area -> read back as (pixel, 8 channels) per lane:
// identity lines loaded and `out` lines stored as whole 128-B lines -> the same values, split into hi + lo, back into the staging
// area in operand layout -> B fragments of conv1' (natural K order).  No block barrier in the epilogues: a wave only touches its own
// pixels.
//
// A step = one stage = 24 MFMAs per wave behind ONE counted vmcnt + barrier; the weight fragments of step J+1 are read between the
// MFMAs of step J (register double buffer), stage J+4 is issued behind MFMA 1 into the slot of stage J.
// vmcnt bookkeeping: LDS-DMAs, loads and stores retire in issue order.  Per step a wave issues one stage (2 pieces), then possibly
// 8 identity loads (for the next chunk) and, behind a chunk's last conv3 step, 8 stores; BtCfg computes the immediate of every
// counted wait from that program.  The per-lane scale / shift vectors come from LDS so that no other vector-memory instruction
// exists.  At a tile boundary every wave waits for the next patch (requested behind the tile's last barrier, in front of its last 8
// stores, which may stay in flight); the partner workgroup covers.  DESIGN.md 5b has the measurements (phases, ablations, what was
// tried and dropped).
#pragma once
#include "mpx_conv.h"
#include <type_traits>

namespace mpx {

struct BtParams {
    const half_t* t_hi;      // conv2 input planes [B][H][W][64] (HEAD: the block input planes, conv1 runs on the patch)
    const half_t* t_lo;
    const half_t* w0_hi;     // HEAD only: the block's OWN conv1 (1x1, 64 -> 64), piece-major [>= 64][64]; then t_* = the block input
    const half_t* w0_lo;
    const float* sc0;
    const float* sh0;
    const half_t* w2_hi;     // conv2 weights, piece-major [>= 64][576] (mpx_pack_conv_weights)
    const half_t* w2_lo;
    const float* sc2;
    const float* sh2;
    const half_t* w3_hi;     // conv3 weights, piece-major [256][64] (DUAL: [256][128]), columns [0,64) K-permuted
    const half_t* w3_lo;
    const float* sc3;
    const float* sh3;
    const half_t* r_hi;      // identity planes [B][H][W][256]; DUAL: the block input planes [B][H][W][64]
    const half_t* r_lo;
    half_t* y_hi;            // block output planes [B][H][W][256]
    half_t* y_lo;
    const half_t* w1_hi;     // next conv1 weights, piece-major [>= C1][256]
    const half_t* w1_lo;
    const float* sc1;
    const float* sh1;
    half_t* z_hi;            // its output planes [B][H][W][C1]
    half_t* z_lo;
    int B, H, W;
    int tiles_x;             // W / 14
    int tiles_per_img;       // (H / 8) * (W / 14)
    int n_tiles;             // B * tiles_per_img
#ifdef MPX_DIAG
    unsigned long long* stamps;   // diagnostic build only (tools/probes/btail_phases.py): 8 u64 per workgroup
#endif
};

// Diagnostic build: wave 0 adds up, over the tiles of its workgroup, the s_memtime cycles of each phase.
#ifdef MPX_DIAG
#define BT_PHASE(acc)                                                       \
    do {                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
        (acc) += now_ - bt_last;                                            \
        bt_last = now_;                                                     \
    } while (0)
// finer: inside step_top -- [0] work since the last step_top, [1] counted vmcnt wait, [2] barrier, [3] stage DMA issue
#define BT_SUB(i)                                                           \
    do {                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
        bt_sub[i] += now_ - bt_sub_last;                                    \
        bt_sub_last = now_;                                                 \
    } while (0)
#else
#define BT_PHASE(acc)
#define BT_SUB(i)
#endif

constexpr int BT_TY = 8, BT_TX = 14;                    // output pixels of a tile: rows x columns
constexpr int BT_PY = BT_TY + 2;                        // patch rows; a patch row is 16 pixels (columns -1 .. 14 of the tile)
constexpr int BT_NRING = 4;
constexpr int BT_STAGE = 8192;                          // [hi: 64 rows x 64 B][lo]
constexpr int BT_BLK = BT_PY * 16 * 64;                 // one (32-channel chunk, plane) block of the patch: 160 rows x 64 B
// LDS map.  The vectors come first and the ring last so that every per-lane base address + immediate offset of the many
// unrolled ds_read / ds_write stays below the 64-KB reach of the instruction's offset field (otherwise hipcc keeps one
// address VGPR per access).
constexpr int BT_OFF_VEC = 0;                           // scale / shift vectors, <= 4 KB
constexpr int BT_OFF_PATCH = 4096;
constexpr int BT_OFF_RING = BT_OFF_PATCH + 4 * BT_BLK + 256;  // + slack: the dead lanes of the last patch row read 2 rows further
constexpr int BT_MID = 64, BT_OUT = 256;
#ifndef BT_AUX_LOAD
#define BT_AUX_LOAD 2       // cache policy bits of the identity loads / output stores (2 = nt); probe builds override them
#endif
#ifndef BT_AUX_STORE
#define BT_AUX_STORE 2
#endif
// timing-only ablations (probe builds, wrong results): BT_ABL bit 0 = patch DMAs, 1 = identity / block-input loads, 2 = stores,
// 3 = weight DMAs carry an out-of-range offset: the instruction is still issued, the memory access is not made
#ifndef BT_ABL
#define BT_ABL 0
#endif

template <bool DUAL_, int C1_, bool HEAD_ = false>
struct BtCfg {
    static constexpr bool DUAL = DUAL_;
    static constexpr int C1 = C1_;
    static constexpr bool HEAD = HEAD_;                 // the block's own conv1 (64 -> 64, 1x1) runs on the patch first: layer1.0 whole
    static_assert(!HEAD || DUAL, "the head conv1 has 64 input channels: only the first block of layer1");
    static constexpr int NH = HEAD ? 2 : 0;             // its K steps (K = 64), in front of conv2's
    static constexpr int N2 = 18;                       // conv2 steps: 9 taps x 2 chunks of 32 channels
    static constexpr int JC = NH + N2;                  // first step of the chunk phase
    static constexpr int S3 = DUAL ? 4 : 2;             // conv3 steps per output chunk (K = 64, + 64 of the downsample branch)
    static constexpr int H1 = C1 / 64;                  // 64-row blocks of conv1'
    static constexpr int S1 = 2 * H1;                   // conv1' steps per chunk
    static constexpr int SC = S3 + S1;
    static constexpr int NSTEP = NH + N2 + 4 * SC;
    static constexpr int NVEC = 2 * (BT_MID + BT_OUT + (C1 ? C1 : 4)) + (HEAD ? 2 * BT_MID : 0);    // floats: sc2 sh2 sc3 sh3 sc1 sh1 (sc0 sh0)
    static constexpr int LDS = BT_OFF_RING + BT_NRING * BT_STAGE;
    static_assert(NVEC * 4 <= BT_OFF_PATCH, "vectors overflow their LDS area");
    static_assert(C1 == 64 || C1 == 128, "conv1' has 64 or 128 output channels");
    static_assert(NSTEP % 2 == 0, "fragment double buffer: steps alternate parity across tiles");

    // ---- the vector-memory program of one tile, per wave (see the header) ----
    static constexpr int mod(int j) { return ((j % NSTEP) + NSTEP) % NSTEP; }
    static constexpr int JX = 2;                        // DUAL: step behind whose stage the 8 block-input fragment loads are issued
    static constexpr int pre(int j) {                   // loads issued right behind the stage of step j
        j = mod(j);
        if (DUAL) return (!HEAD && j == JX) ? 8 : 0;                        // HEAD reads the block input fragments from the patch
        if (j == JC - 3) return 8;                                          // identity lines of chunk 0
        for (int c = 1; c < 4; ++c)
            if (j == JC + (c - 1) * SC + S3 + (H1 == 2 ? S1 - 1 : 0)) return 8;   // ... of chunk c: in the first (C1 = 128: last,
                                                                            // the registers are short) conv1' step of chunk c-1
        return 0;
    }
    static constexpr int post(int j) {                  // stores at the end of step j
        j = mod(j);
        int n = 0;
        for (int c = 0; c < 4; ++c)
            if (j == JC + c * SC + S3 - 1) n += 8;                          // `out` lines of chunk c
        if (j == NSTEP - 1) n += 8 * H1;                                    // t1' lines
        return n;
    }
    static constexpr int all(int j) { return 2 + pre(j) + post(j); }
    // C1 = 128 is short of registers across a chunk epilogue: the last conv3 step of a chunk does not read the next step's weight
    // fragments ahead; they are read behind the epilogue instead (their stage has landed by then either way)
    static constexpr bool late_a(int j) {
        if (H1 != 2) return false;
        for (int c = 0; c < 4; ++c)
            if (j == JC + c * SC + S3 - 1) return true;
        return false;
    }
    // top of step j: the stage of step j (issued first thing in step j-3) has landed; everything behind it may be in flight
    static constexpr int wait_top(int j) { return pre(j - 3) + post(j - 3) + all(j - 2) + all(j - 1); }
};

// relu, then hi = fp16(v), lo = fp16(v - hi) for 8 values.  `one` is 1.0f in a register the optimiser cannot see through, so that
// v - (float)hi stays fma((float)hi, -one, v) and becomes ONE v_fma_mix_f32 (fp16 operand converted in the instruction) instead of a
// conversion and a subtraction: the epilogues of this kernel are VALU-bound.  Same values as split_f32 (fma(h, -1, v) = v - h exactly
// rounded once).
__device__ __forceinline__ void bt_relu_split8(const float (&v)[8], float one, h8& oh, h8& ol) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = fmaxf(v[j], 0.f);
        const half_t hi = (half_t)r;
        oh[j] = hi;
        ol[j] = (half_t)__builtin_fmaf((float)hi, -one, r);
    }
}

template <class C>
__global__ __launch_bounds__(256, 2) void btail_f16x3_kernel(const BtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    constexpr bool DUAL = C::DUAL;
    constexpr bool HEAD = C::HEAD;
    constexpr int C1 = C::C1, N2 = C::N2, NH = C::NH, JC = C::JC, S3 = C::S3, S1 = C::S1, SC = C::SC, H1 = C::H1, NSTEP = C::NSTEP;
    constexpr unsigned OOB = 0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, lg = lane >> 4;
    float one = 1.0f;
    asm volatile("" : "+s"(one));       // opaque 1.0 (bt_relu_split8)

    // tiles of this workgroup: blocks of one XCD (b % 8) walk neighbouring tiles (shared halo rows stay in that XCD's L2)
    const int G = gridDim.x;                                            // a multiple of 8 (host)
    const int v0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (v0 >= p.n_tiles) return;
    const int my_tiles = (p.n_tiles - 1 - v0) / G + 1;

    char* const ring = smem + BT_OFF_RING;
    char* const patch = smem + BT_OFF_PATCH;
    char* const stg = patch + wave * 8192;                             // wave-private staging (inside the patch bytes)
    float* const vec = (float*)(smem + BT_OFF_VEC);
    constexpr int V_SC2 = 0, V_SH2 = BT_MID, V_SC3 = 2 * BT_MID, V_SH3 = 2 * BT_MID + BT_OUT, V_SC1 = 2 * BT_MID + 2 * BT_OUT,
                  V_SH1 = V_SC1 + (C1 ? C1 : 4), V_SC0 = V_SH1 + (C1 ? C1 : 4), V_SH0 = V_SC0 + BT_MID;

    // scale / shift vectors -> LDS (published by the first barrier below)
    for (int i = tid; i < C::NVEC; i += 256) {
        float v = 0.f;
        if (i < V_SH2) v = p.sc2[i];
        else if (i < V_SC3) v = p.sh2[i - V_SH2];
        else if (i < V_SH3) v = p.sc3[i - V_SC3];
        else if (i < V_SC1) v = p.sh3[i - V_SH3];
        else if (C1 && i < V_SH1) v = p.sc1[i - V_SC1];
        else if (C1 && i < V_SH1 + C1) v = p.sh1[i - V_SH1];
        else if (HEAD && i >= V_SC0 && i < V_SH0) v = p.sc0[i - V_SC0];
        else if (HEAD && i >= V_SH0) v = p.sh0[i - V_SH0];
        vec[i] = v;
    }

    // ---- weight stream --------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t w0h = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? p.w0_hi : p.w2_hi), 0, 64 * 64 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w0l = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? p.w0_lo : p.w2_lo), 0, 64 * 64 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2_hi, 0, 64 * 576 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2l = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2_lo, 0, 64 * 576 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3_hi, 0, BT_OUT * S3 * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3l = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3_lo, 0, BT_OUT * S3 * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1_hi, 0, (C1 ? C1 : 16) * BT_OUT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1l = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1_lo, 0, (C1 ? C1 : 16) * BT_OUT * 2, 0x00020000);
    const int w_lane = (lane * 16) | ((BT_ABL & 8) ? (int)OOB : 0);
    // stage JS of the tile program into ring slot `slot`: this wave moves piece `wave` (16 rows) of each plane
    auto issue_stage = [&](auto js_tag, int slot) {
        constexpr int JS = decltype(js_tag)::value;
        char* const d = ring + slot * BT_STAGE + wave * 1024;
        if (JS < NH) {                                  // the head conv1: rows [16 wave, +16), K step JS of 2
            const int soff = (wave * 2 + JS) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w0h, MPX_LDS_PTR(d), 16, w_lane, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w0l, MPX_LDS_PTR(d + 4096), 16, w_lane, soff, 0, 0);
        } else if (JS < JC) {
            const int soff = (wave * 18 + JS - NH) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2h, MPX_LDS_PTR(d), 16, w_lane, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2l, MPX_LDS_PTR(d + 4096), 16, w_lane, soff, 0, 0);
        } else {
            constexpr int cc = (JS - JC) / SC, r = (JS - JC) % SC;
            if (r < S3) {
                const int soff = ((4 * cc + wave) * S3 + r) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3h, MPX_LDS_PTR(d), 16, w_lane, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3l, MPX_LDS_PTR(d + 4096), 16, w_lane, soff, 0, 0);
            } else {
                constexpr int r1 = r - S3, kk = r1 / (H1 ? H1 : 1), hb = r1 % (H1 ? H1 : 1);
                const int soff = ((4 * hb + wave) * 8 + 2 * cc + kk) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1h, MPX_LDS_PTR(d), 16, w_lane, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1l, MPX_LDS_PTR(d + 4096), 16, w_lane, soff, 0, 0);
            }
        }
    };

    // ---- fragment addressing ----------------------------------------------------------------------------------------------
    const int a_off = lp * 64 + ((lg ^ (((lane >> 3) & 1) << 1)) << 4);           // weight rows: mpx_conv.h's swizzle
    int pb_off[2][3];      // patch: byte offset of this lane's pixel fragment b at tap column kx (tap row ky: + ky * 1024)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int row = (2 * wave + b) * 16 + lp + kx;
            pb_off[b][kx] = row * 64 + ((lg ^ ((((lp + kx) >> 2) & 1) << 1)) << 4);
        }
    const int sb_off = lp * 64 + ((lg ^ (((lp >> 3) & 1) << 1)) << 4);             // staging, operand layout: + (b*4 + kk*2 + plane)*1024

    // ---- per-lane geometry of the line-layout passes: unit u = (pixel slot u*8 + lane/8, channels 8*(lane%8) ..) ----------------
    const int uq = lane & 7;
    int u_pix[4];          // pixel index within the image of unit u's slot (set per tile), or -1 (dead lane)

    f4 acc1[C1 ? 4 * H1 : 1][2];
    h8 t2h[2][2], t2l[2][2];       // conv3's B operand: [K step][pixel fragment]
    h8 x0h[2][2], x0l[2][2];       // DUAL: the block input under the downsample conv, B layout, natural K
    u4 idh[4], idl[4];             // identity lines of the chunk about to be finished, one per unit
    struct FragA { h8 hi[4], lo[4]; };
    struct FragB { h8 hi[2], lo[2]; };
    FragA fa[2];                   // weight fragments of the current / next step (index = step parity; NSTEP is even)
    FragB fb[2];                   // conv2: patch fragments, likewise
    FragB cb[2];                   // conv1': fragments of the chunk's two K steps, from the staging area

#ifdef MPX_DIAG
    unsigned long long bt_ph[6] = {0, 0, 0, 0, 0, 0};      // patch wait, conv2, conv3 steps, chunk epilogues, conv1' steps, t1' epilogue + end barrier
    const unsigned long long bt_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long bt_last = bt_t0;
    unsigned long long bt_sub[5] = {0, 0, 0, 0, 0}, bt_sub_last = bt_t0;
#endif
    int sbase = 0;                  // ring slot of step 0 of the current tile
    // prologue: stages 0 .. 3
    issue_stage(std::integral_constant<int, 0>{}, 0);
    issue_stage(std::integral_constant<int, 1>{}, 1);
    issue_stage(std::integral_constant<int, 2>{}, 2);
    issue_stage(std::integral_constant<int, 3>{}, 3);

    // t1 patch of this workgroup's tile `it`: 40 pieces of 16 pixels x 32 channels; this wave moves pieces wave*10 .. wave*10+9
    auto issue_patch = [&](int it) {
        const int t = v0 + it * G;
        const int n = t / p.tiles_per_img;
        const int rt = t - n * p.tiles_per_img;
        const int ty = rt / p.tiles_x, tx = rt - ty * p.tiles_x;
        const int y0 = ty * BT_TY, x0 = tx * BT_TX;
        const size_t img_pix = (size_t)n * p.H * p.W;
        const int img_bytes64 = p.H * p.W * BT_MID * 2;
        const __amdgpu_buffer_rsrc_t th = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t_hi + img_pix * BT_MID), 0, img_bytes64, 0x00020000);
        const __amdgpu_buffer_rsrc_t tl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t_lo + img_pix * BT_MID), 0, img_bytes64, 0x00020000);
        const int px = lane >> 2;
        const int ix = x0 - 1 + px;
        const int col = ix * BT_MID * 2 + (((lane & 3) ^ (((px >> 2) & 1) << 1)) << 4);
        const int col_oob = (ix | (p.W - 1 - ix)) & (int)OOB;      // ORed in AFTER the row offset is added (a small negative + row would wrap)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int idx = wave * 10 + i;
            const int py = idx >> 2, s = (idx >> 1) & 1, plane = idx & 1;
            const int iy = y0 - 1 + py;
            const int voff = (col + iy * p.W * BT_MID * 2) | col_oob | ((iy | (p.H - 1 - iy)) & (int)OOB) | (((BT_ABL & 1) && it > 0) ? (int)OOB : 0);
            char* const d = patch + (s * 2 + plane) * BT_BLK + py * 1024;
            if (plane == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(th, MPX_LDS_PTR(d), 16, voff, s * 64, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(tl, MPX_LDS_PTR(d), 16, voff, s * 64, 0, 0);
        }
    };
    issue_patch(0);

    for (int it = 0; it < my_tiles; ++it) {
        const int t = v0 + it * G;
        const int n = t / p.tiles_per_img;
        const int rt = t - n * p.tiles_per_img;
        const int ty = rt / p.tiles_x, tx = rt - ty * p.tiles_x;
        const int y0 = ty * BT_TY, x0 = tx * BT_TX;
        const size_t img_pix = (size_t)n * p.H * p.W;
        const int img_bytes64 = p.H * p.W * BT_MID * 2;
        // unit geometry, descriptors of the 256-channel planes of this image
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int slot = u * 8 + (lane >> 3);
            const int pc = slot & 15;
            u_pix[u] = pc < BT_TX ? (y0 + 2 * wave + (slot >> 4)) * p.W + x0 + pc : -1;
        }
        const int img_bytes256 = p.H * p.W * BT_OUT * 2;
        const __amdgpu_buffer_rsrc_t yh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_hi + img_pix * BT_OUT), 0, img_bytes256, 0x00020000);
        const __amdgpu_buffer_rsrc_t yl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_lo + img_pix * BT_OUT), 0, img_bytes256, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.r_hi + img_pix * (DUAL ? BT_MID : BT_OUT)), 0, DUAL ? img_bytes64 : img_bytes256, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.r_lo + img_pix * (DUAL ? BT_MID : BT_OUT)), 0, DUAL ? img_bytes64 : img_bytes256, 0x00020000);
        const __amdgpu_buffer_rsrc_t zh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.z_hi + img_pix * (C1 ? C1 : 1)), 0, p.H * p.W * (C1 ? C1 : 1) * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t zl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.z_lo + img_pix * (C1 ? C1 : 1)), 0, p.H * p.W * (C1 ? C1 : 1) * 2, 0x00020000);

        __builtin_amdgcn_sched_barrier(0);
        // the patch (and everything older) has landed; behind it only the previous tile's last 8 stores may be in flight
        if (it == 0) wait_vmcnt<0>();
        else wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        BT_PHASE(bt_ph[0]);

        // identity lines of chunk c (8 loads): unit u reads channels [64c + 8*uq, +8) of its pixel
        auto issue_identity = [&](int c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int voff = (u_pix[u] * (BT_OUT * 2) + uq * 16) | (u_pix[u] & (int)OOB) | ((BT_ABL & 2) ? (int)OOB : 0);
                idh[u] = __builtin_amdgcn_raw_buffer_load_b128(rh_rs, voff, c * 128, BT_AUX_LOAD);
                idl[u] = __builtin_amdgcn_raw_buffer_load_b128(rl_rs, voff, c * 128, BT_AUX_LOAD);
            }
        };
        auto read_a1 = [&](int slot, FragA& f, int i) {        // fragment read i = 0..7 of a stage: hi rows 0..3, lo rows 0..3
            const char* s = ring + slot * BT_STAGE + a_off + (i & 3) * 1024;
            if (i < 4) f.hi[i] = *(const h8*)s;
            else f.lo[i - 4] = *(const h8*)(s + 4096);
        };
        auto read_patch1 = [&](auto j_tag, FragB& f, int i) {  // fragment read i = 0..3 of conv2 step J: hi b0, hi b1, lo b0, lo b1
            constexpr int J = decltype(j_tag)::value;
            constexpr int tap = J >> 1, s = J & 1, ky = tap / 3, kx = tap % 3;
            const char* q = patch + (s * 2 + (i >> 1)) * BT_BLK + pb_off[i & 1][kx] + ky * 1024;
            if (i < 2) f.hi[i] = *(const h8*)q;
            else f.lo[i - 2] = *(const h8*)q;
        };
        // Step J: MFMAs on the A fragments fa[J & 1] (read during step J-1) and the given B operand, while the fragments of step J+1
        // are read (its stage has landed: the counted wait + barrier at the top), stage J+4 is issued into the slot of stage J
        // (every wave finished reading it before the barrier) and, in some steps, the identity lines of a later chunk are requested.
        auto step = [&](auto j_tag, f4 (*acc)[2], const h8 (&bh)[2], const h8 (&bl)[2]) {
            constexpr int J = decltype(j_tag)::value;
            const FragA& A = fa[J & 1];
            FragA& An = fa[(J + 1) & 1];
            __builtin_amdgcn_sched_barrier(0);
            BT_SUB(0);
            wait_vmcnt<C::wait_top(J)>();
            BT_SUB(1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            BT_SUB(2);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int a = i / 6, r = i % 6, term = r >> 1, b = r & 1;
                if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi[a], bl[b], acc[a][b], 0, 0, 0);
                else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo[a], bh[b], acc[a][b], 0, 0, 0);
                else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi[a], bh[b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if ((i & 1) == 0 && i < 16 && !C::late_a(J)) {
                    read_a1((sbase + J + 1) & 3, An, i >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((i & 1) == 0 && i >= 16 && J + 1 > NH && J + 1 < JC) {      // (the first conv2 step of HEAD reads them behind the t1 writes)
                    read_patch1(std::integral_constant<int, (J + 1 > NH && J + 1 < JC ? J + 1 - NH : 0)>{}, fb[(J + 1) & 1], (i - 16) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((i & 1) == 0 && i >= 16 && J >= JC && (J - JC) % SC == S3 + H1 - 1) {
                    // the step before conv1's second K step of this chunk: its B fragments (written by the chunk epilogue)
                    const int q = (i - 16) >> 1, b = q & 1, plane = q >> 1;
                    const h8 f = *(const h8*)(stg + (b * 4 + 2 + plane) * 1024 + sb_off);
                    if (plane == 0) cb[1].hi[b] = f;
                    else cb[1].lo[b] = f;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i == 1) {
                    issue_stage(std::integral_constant<int, (J + 4) % NSTEP>{}, (sbase + J) & 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i == 5 && C::pre(J)) {
                    if (DUAL) {
                        // block input fragments (B layout, natural K): lane (pixel lp, group lg) reads channels [32 kk + 8 lg, +8)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            const int pc = lp < BT_TX ? lp : BT_TX - 1;            // dead lanes re-read a live pixel
                            const int voff = ((y0 + 2 * wave + bb) * p.W + x0 + pc) * (BT_MID * 2) + lg * 16;
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                x0h[kk][bb] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rh_rs, voff, kk * 64, 0));
                                x0l[kk][bb] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rl_rs, voff, kk * 64, 0));
                            }
                        }
                    } else {
                        issue_identity(J < JC ? 0 : (J - JC) / SC + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // first fragments of the tile: the patch has landed; the A fragments of step 0 were read during the previous tile's last step
        if (it == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) read_a1(sbase & 3, fa[0], i);
        }
        if constexpr (HEAD) {
            // ================= the block's own conv1 on the patch (layer1.0 whole: the patch holds the block INPUT) ====================
            // The downsample branch's operand = the block input at this wave's own pixels: B fragments from the patch (tap (1,1)),
            // before conv1 overwrites it.  (The step barriers below order these reads of every wave before any write.)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    x0h[kk][b] = *(const h8*)(patch + (kk * 2 + 0) * BT_BLK + pb_off[b][1] + 1024);
                    x0l[kk][b] = *(const h8*)(patch + (kk * 2 + 1) * BT_BLK + pb_off[b][1] + 1024);
                }
            // conv1 is pointwise: t1 of every patch pixel, in place.  Patch row r (16 pixels = one fragment) belongs to wave r % 4:
            // rows w and w+4 in the regular step (two fragments), row w+8 (waves 0 and 1 only) in 12 extra MFMAs on the same weight
            // fragments, which stay in fa[J & 1] until the next step's read-ahead overwrites them.
            const int h_off = lp * 64 + ((lg ^ (((lp >> 2) & 1) << 1)) << 4);       // this lane's 16-B chunk of its pixel's row
            f4 acch[4][2], accx[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acch[a][0] = acch[a][1] = (f4){0.f, 0.f, 0.f, 0.f};
                accx[a] = (f4){0.f, 0.f, 0.f, 0.f};
            }
            auto head_step = [&](auto k_tag) {
                constexpr int kk = decltype(k_tag)::value;
                h8 bh[2], bl[2], xh, xl;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    bh[b] = *(const h8*)(patch + (kk * 2 + 0) * BT_BLK + (wave + 4 * b) * 1024 + h_off);
                    bl[b] = *(const h8*)(patch + (kk * 2 + 1) * BT_BLK + (wave + 4 * b) * 1024 + h_off);
                }
                if (wave < 2) {
                    xh = *(const h8*)(patch + (kk * 2 + 0) * BT_BLK + (wave + 8) * 1024 + h_off);
                    xl = *(const h8*)(patch + (kk * 2 + 1) * BT_BLK + (wave + 8) * 1024 + h_off);
                }
                step(std::integral_constant<int, kk>{}, acch, bh, bl);
                if (wave < 2) {
                    const FragA& A = fa[kk & 1];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        accx[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi[a], xl, accx[a], 0, 0, 0);
                        accx[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo[a], xh, accx[a], 0, 0, 0);
                        accx[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi[a], xh, accx[a], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            head_step(std::integral_constant<int, 0>{});
            head_step(std::integral_constant<int, 1>{});
            // t1 = relu(acc * scale0 + shift0), ZERO outside the image (conv2's padding), split, written over the block input: D layout
            // (lane = pixel, 4 consecutive channels per register quad) -> the 8-B piece (a & 1) * 32 + 8 g of the pixel's row of chunk a >> 1
            auto write_t1 = [&](const f4 (&v4)[4], int r) {
                const int iy = y0 - 1 + r, ix = x0 - 1 + lp;
                const float keep = ((iy | (p.H - 1 - iy) | ix | (p.W - 1 - ix)) < 0) ? 0.f : 1.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f4 sc = *(const f4*)(vec + V_SC0 + a * 16 + 4 * lg);
                    const f4 sh = *(const f4*)(vec + V_SH0 + a * 16 + 4 * lg);
                    const f4 v = v4[a] * sc + sh;
                    h4 hi4, lo4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t = fmaxf(v[j], 0.f) * keep;
                        const half_t hi = (half_t)t;
                        hi4[j] = hi;
                        lo4[j] = (half_t)__builtin_fmaf((float)hi, -one, t);
                    }
                    char* const d = patch + ((a >> 1) * 2) * BT_BLK + (r * 16 + lp) * 64 + ((((a & 1) * 2 + (lg >> 1)) ^ (((lp >> 2) & 1) << 1)) << 4) + (lg & 1) * 8;
                    *(h4*)d = hi4;
                    *(h4*)(d + BT_BLK) = lo4;
                }
            };
            {
                f4 v0[4], v1[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    v0[a] = acch[a][0];
                    v1[a] = acch[a][1];
                }
                write_t1(v0, wave);
                write_t1(v1, wave + 4);
                if (wave < 2) write_t1(accx, wave + 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // t1 of every patch row is in place
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) read_patch1(std::integral_constant<int, 0>{}, fb[NH & 1], i);

        // ================= conv2: 18 steps over the resident patch ==========================================================
        f4 acc2[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc2[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        auto conv2_step = [&](auto k_tag) {     // conv2's K step k = step NH + k of the tile
            constexpr int J = NH + decltype(k_tag)::value;
            step(std::integral_constant<int, J>{}, acc2, fb[J & 1].hi, fb[J & 1].lo);
        };
        conv2_step(std::integral_constant<int, 0>{});
        conv2_step(std::integral_constant<int, 1>{});
        conv2_step(std::integral_constant<int, 2>{});
        conv2_step(std::integral_constant<int, 3>{});
        conv2_step(std::integral_constant<int, 4>{});
        conv2_step(std::integral_constant<int, 5>{});
        conv2_step(std::integral_constant<int, 6>{});
        conv2_step(std::integral_constant<int, 7>{});
        conv2_step(std::integral_constant<int, 8>{});
        conv2_step(std::integral_constant<int, 9>{});
        conv2_step(std::integral_constant<int, 10>{});
        conv2_step(std::integral_constant<int, 11>{});
        conv2_step(std::integral_constant<int, 12>{});
        conv2_step(std::integral_constant<int, 13>{});
        conv2_step(std::integral_constant<int, 14>{});
        conv2_step(std::integral_constant<int, 15>{});
        conv2_step(std::integral_constant<int, 16>{});
        conv2_step(std::integral_constant<int, 17>{});
        BT_PHASE(bt_ph[1]);

        // t2 = relu(acc2 * scale2 + shift2), split; registers 4g..4g+3 of row fragments 2s, 2s+1 are K positions 8g..8g+7 of step s
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int a = 2 * s + half;
                const f4 sc = *(const f4*)(vec + V_SC2 + a * 16 + 4 * lg);
                const f4 sh = *(const f4*)(vec + V_SH2 + a * 16 + 4 * lg);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f4 v = acc2[a][b] * sc + sh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float r = fmaxf(v[j], 0.f);
                        const half_t hi = (half_t)r;
                        t2h[s][b][half * 4 + j] = hi;
                        t2l[s][b][half * 4 + j] = (half_t)__builtin_fmaf((float)hi, -one, r);
                    }
                }
            }

#pragma unroll
        for (int a = 0; a < 4 * H1; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc1[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

        // ================= 4 output chunks of 64 channels =======================================================================
        auto chunk = [&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            constexpr int J0 = JC + c * SC;
            f4 acc3[4][2];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc3[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
            step(std::integral_constant<int, J0>{}, acc3, t2h[0], t2l[0]);
            step(std::integral_constant<int, J0 + 1>{}, acc3, t2h[1], t2l[1]);
            if constexpr (DUAL) {
                step(std::integral_constant<int, J0 + 2>{}, acc3, x0h[0], x0l[0]);
                step(std::integral_constant<int, J0 + 3>{}, acc3, x0h[1], x0l[1]);
            }
            // ---- chunk epilogue (wave-private) ----
            __builtin_amdgcn_sched_barrier(0);
            BT_PHASE(bt_ph[2]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f4 sc = *(const f4*)(vec + V_SC3 + c * 64 + a * 16 + 4 * lg);
                const f4 sh = *(const f4*)(vec + V_SH3 + c * 64 + a * 16 + 4 * lg);
#pragma unroll
                for (int b = 0; b < 2; ++b) *(f4*)(stg + (b * 16 + lp) * 256 + (((a * 4 + lg) ^ lp) << 4)) = acc3[a][b] * sc + sh;
            }
            // one pixel fragment (16 slots = units 2b, 2b+1) at a time: its operand bytes replace its own fp32 bytes only
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f4 v0[2], v1[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int slot = (2 * b + k) * 8 + (lane >> 3);
                    v0[k] = *(const f4*)(stg + slot * 256 + (((2 * uq) ^ (slot & 15)) << 4));
                    v1[k] = *(const f4*)(stg + slot * 256 + (((2 * uq + 1) ^ (slot & 15)) << 4));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = 2 * b + k;
                    const int slot = u * 8 + (lane >> 3);
                    float v[8] = {v0[k][0], v0[k][1], v0[k][2], v0[k][3], v1[k][0], v1[k][1], v1[k][2], v1[k][3]};
                    if (!DUAL) {
                        // + identity (hi + lo): two v_fma_mix_f32 per element (the fp16 -> fp32 conversions ride in the instruction)
                        const h8 a8 = __builtin_bit_cast(h8, idh[u]);
                        const h8 c8 = __builtin_bit_cast(h8, idl[u]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf((float)c8[j], one, __builtin_fmaf((float)a8[j], one, v[j]));
                    }
                    h8 oh, ol;
                    bt_relu_split8(v, one, oh, ol);
                    const int voff = (u_pix[u] * (BT_OUT * 2) + uq * 16) | (u_pix[u] & (int)OOB) | ((BT_ABL & 4) ? (int)OOB : 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, oh), yh, voff, c * 128, BT_AUX_STORE);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, ol), yl, voff, c * 128, BT_AUX_STORE);
                    // operand layout for conv1': [fragment b][K step uq>>2][plane][16 slots][64 B], chunk uq&3 swizzled by the slot
                    char* const d = stg + (b * 4 + (uq >> 2) * 2) * 1024 + (slot & 15) * 64 + (((uq & 3) ^ (((slot >> 3) & 1) << 1)) << 4);
                    *(h8*)d = oh;
                    *(h8*)(d + 1024) = ol;
                }
            }
            if (C::late_a(J0 + S3 - 1)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) read_a1((sbase + J0 + S3) & 3, fa[(J0 + S3) & 1], i);
            }
            // B fragments of conv1's first K step of this chunk (the second one's are read during the step before it)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                cb[0].hi[b] = *(const h8*)(stg + (b * 4 + 0) * 1024 + sb_off);
                cb[0].lo[b] = *(const h8*)(stg + (b * 4 + 1) * 1024 + sb_off);
            }
            __builtin_amdgcn_sched_barrier(0);
            BT_PHASE(bt_ph[3]);
            // conv1' steps: K step kk = r1 / H1 of this chunk, 64-row block hb = r1 % H1
            step(std::integral_constant<int, J0 + S3>{}, &acc1[0], cb[0].hi, cb[0].lo);
            if constexpr (H1 == 1) {
                step(std::integral_constant<int, J0 + S3 + 1>{}, &acc1[0], cb[1].hi, cb[1].lo);
            } else {
                step(std::integral_constant<int, J0 + S3 + 1>{}, &acc1[4], cb[0].hi, cb[0].lo);
                step(std::integral_constant<int, J0 + S3 + 2>{}, &acc1[0], cb[1].hi, cb[1].lo);
                step(std::integral_constant<int, J0 + S3 + 3>{}, &acc1[4], cb[1].hi, cb[1].lo);
            }
            BT_PHASE(bt_ph[4]);
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});

        // ================= t1' = relu(acc1 * scale1 + shift1) ======================================================================
        // The last 8 stores are held back until the next tile's patch has been requested (the staging bytes it lands in are free
        // once every wave has read its last lines), so that the wait for the patch does not wait for them.
        __builtin_amdgcn_sched_barrier(0);
        h8 zoh[4], zol[4];
#pragma unroll
        for (int hb = 0; hb < H1; ++hb) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f4 sc = *(const f4*)(vec + V_SC1 + hb * 64 + a * 16 + 4 * lg);
                const f4 sh = *(const f4*)(vec + V_SH1 + hb * 64 + a * 16 + 4 * lg);
#pragma unroll
                for (int b = 0; b < 2; ++b) *(f4*)(stg + (b * 16 + lp) * 256 + (((a * 4 + lg) ^ lp) << 4)) = acc1[hb * 4 + a][b] * sc + sh;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int slot = u * 8 + (lane >> 3);
                const f4 w0 = *(const f4*)(stg + slot * 256 + (((2 * uq) ^ (slot & 15)) << 4));
                const f4 w1 = *(const f4*)(stg + slot * 256 + (((2 * uq + 1) ^ (slot & 15)) << 4));
                const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                bt_relu_split8(v, one, zoh[u], zol[u]);
                if (hb + 1 < H1) {
                    const int voff = (u_pix[u] * (C1 * 2) + uq * 16) | (u_pix[u] & (int)OOB) | ((BT_ABL & 4) ? (int)OOB : 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, zoh[u]), zh, voff, hb * 128, BT_AUX_STORE);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, zol[u]), zl, voff, hb * 128, BT_AUX_STORE);
                }
            }
        }
        // every wave is done with its staging area (= the patch bytes) before the next tile's patch lands
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BT_SUB(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        BT_SUB(2);
        if (it + 1 < my_tiles) issue_patch(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        BT_SUB(4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int voff = (u_pix[u] * (C1 * 2) + uq * 16) | (u_pix[u] & (int)OOB) | ((BT_ABL & 4) ? (int)OOB : 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, zoh[u]), zh, voff, (H1 - 1) * 128, BT_AUX_STORE);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, zol[u]), zl, voff, (H1 - 1) * 128, BT_AUX_STORE);
        }
        __builtin_amdgcn_sched_barrier(0);
        BT_SUB(3);
        sbase = (sbase + NSTEP) & 3;
        BT_PHASE(bt_ph[5]);
    }
    wait_vmcnt<0>();
#ifdef MPX_DIAG
    if (p.stamps && tid == 0) {
        unsigned long long* o_ = p.stamps + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o_[i] = bt_ph[i];
        o_[6] = __builtin_amdgcn_s_memtime() - bt_t0;
        o_[7] = (unsigned long long)my_tiles;
        unsigned long long* q_ = p.stamps + (size_t)(gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) q_[i] = bt_sub[i];
    }
#endif
#endif
}

typedef BtCfg<false, 64> BtResC64;      // layer1.1 / layer1.2 tails: identity = the trunk, next conv1 256 -> 64
typedef BtCfg<false, 128> BtResC128;    // layer1's last block: next conv1 = layer2.0.conv1, 256 -> 128
typedef BtCfg<true, 64> BtDualC64;      // layer1.0: downsample branch K-concatenated, next conv1 256 -> 64
typedef BtCfg<true, 64, true> BtHeadC64; // layer1.0 whole: its own conv1 runs on the patch of the block input first

// The wait immediates, checked against the program written out by hand for the identity tail (steps 16 .. 23: the last conv2 steps,
// conv3 of chunk 0 with its 8 identity loads requested in step 15 and its 8 stores behind step 19, conv1' of chunk 0 with the next
// identity loads in step 20): stage DMAs count 2, everything younger than the awaited stage may be in flight.
static_assert(BtResC64::NSTEP == 34 && BtResC128::NSTEP == 42 && BtDualC64::NSTEP == 42, "steps per tile");
static_assert(BtResC64::wait_top(14) == 4 && BtResC64::wait_top(16) == 12 && BtResC64::wait_top(18) == 12 && BtResC64::wait_top(19) == 4,
              "conv2 -> chunk 0");
static_assert(BtResC64::wait_top(20) == 12 && BtResC64::wait_top(21) == 20 && BtResC64::wait_top(22) == 20 && BtResC64::wait_top(23) == 12,
              "chunk pattern: conv1' steps behind the stores, the next chunk's conv3 steps behind the identity loads");
static_assert(BtResC64::wait_top(0) == 20 && BtDualC64::wait_top(4) == 12 && BtDualC64::wait_top(6) == 4, "tile start; block-input loads of the DUAL tail in step 2");

}  // namespace mpx
