// mpx_conv.h -- K1/K2: conv (implicit GEMM) + BatchNorm scale/shift + residual + ReLU, gfx950.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],   k = (ky, kx, ci), ci fastest
// on v_mfma_f32_16x16x32_f16 as three products  W_hi*X_lo + W_lo*X_hi + W_hi*X_hi  (split-fp16
// operands, fp32 accumulation).  A = weights (MFMA rows = cout), B = pixels (MFMA cols), so an
// accumulator register quad is 4 consecutive channels of one pixel.
//
// Workgroup = NWR x NWC waves (cout x pixel), tile TC(cout) x TP(pixel), K advances 32 per step.
// LDS: a ring of NSW stages [W_hi TCx64B][W_lo TCx64B] and a ring of NSX stages [X_hi TPx64B][X_lo TPx64B],
// filled by buffer_load_dwordx4 ... lds in 1-KiB pieces (16 rows x 64 B, lane-linear destination) and
// consumed behind a COUNTED s_waitcnt vmcnt + one raw s_barrier per K step; fragments are double-buffered
// in registers (read step k+1 while the MFMAs of step k run), so the younger DMA stages stay in flight
// under the MFMAs (cdna_hip_programming.md 5, "Pipelining across barriers").
// Rows are 64 B, which makes a ds_read_b128 fragment read 2-way bank conflicted; the 16-B chunk index
// is XORed with ((row>>3)&1)<<1 on the DMA's SOURCE address and on the fragment read (rule 21).
// Epilogue: accumulators -> fp32 tile in LDS (XOR-swizzled 16-B chunks) -> each thread owns 8
// consecutive channels of one pixel: residual planes are read and output planes written as whole
// 128-B lines (16 B per lane, 16 or 8 lanes per pixel row).
//
// Variants (ConvCfg): 128x256 with 8 waves and 3-deep rings (one workgroup per CU; fewest bytes per FLOP) and
// 128x128 with 4 waves, a 2-deep W ring (L2 hits) and a 3-deep X ring (HBM / Infinity Cache latency), 80 KB:
// two workgroups per CU, so the HBM-bound epilogue of one overlaps the MFMA-bound K loop of the other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mpx {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MPX_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MPX_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void split_f32(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}

struct ConvParams {
    const half_t* x_hi;      // input planes, NHWC (pix_stride elements per pixel)
    const half_t* x_lo;
    const half_t* w_hi;      // packed weights [cout_pad][ktot]
    const half_t* w_lo;
    const float* scale;      // [cout_pad]
    const float* shift;
    const half_t* r_hi;      // residual planes [M][cout] or null
    const half_t* r_lo;
    half_t* y_hi;            // output planes [M][cout]
    half_t* y_lo;
    float* y_f32;            // fp32 output [M][cout] (fc) or null
    int hin, win;            // input spatial extent the bounds check uses
    int pix_stride;          // fp16 elements between adjacent input pixels
    int ho, wo;
    int kh, kw, stride, pad;
    int k_per_tap;           // K contributed by one (ky,kx) tap (= cin; 32 for the stem)
    int ktot;                // kh*kw*k_per_tap
    int cout;                // real output channels (store bound and row pitch of y/r); multiple of 8
    int M;                   // B*ho*wo output pixels
    int n_tiles_c;           // cout_pad / TC
    int relu;
    int patch_rows;          // mpx_conv3p.h only: allocated rows of an input patch (multiple of 16)
    // DUAL kernels only (a block's last conv with its downsample branch K-concatenated, mpx_api.hip build_fused):
    // K steps [0, k1/32) read x_* as above; K steps [k1/32, ktot/32) are a 1x1 stride-`stride2` conv over x2_*.
    const half_t* x2_hi;
    const half_t* x2_lo;
    int hin2, win2, pix_stride2, stride2;
    int k1;
    int n_tiles;             // persistent kernels (tools/probes/experimental/mpx_convp.h): n_tiles_p * n_tiles_c, walked by a fixed grid
#ifdef MPX_DIAG
    unsigned long long* stamps;   // diagnostic build only (tools/probes/conv_timeline.py): 8 u64 per workgroup
#endif
};

// Packed weight planes are PIECE-major: a plane [cout_pad][K] is stored as [cout_pad/16][K/32][16 rows][32 halfs], i.e. every
// 1-KiB LDS-DMA piece (16 output channels x one 32-wide K step) is contiguous in memory and already in the order of its LDS image
// (16-B chunk c of row r sits at chunk position c ^ (((r>>3)&1)<<1), the ring's bank swizzle): a lane's source address is
// piece base + lane*16, a piece is 8 whole 128-B lines instead of 16 half lines scattered over 16 rows (the texture addresser is
// the busiest unit of the 1x1 kernels).  Element index of weight (row, k) within a plane:
__host__ __device__ inline size_t w_packed_index(int row, int k, int K) {
    const int r = row & 15, c = (k >> 3) & 3;
    return ((((size_t)(row >> 4) * (size_t)(K >> 5) + (size_t)(k >> 5)) * 16 + r) * 4 + (size_t)(c ^ (((r >> 3) & 1) << 1))) * 8 + (size_t)(k & 7);
}

// Diagnostic build (-DMPX_DIAG, never the product library): wave 0 of every workgroup keeps s_memtime stamps of its
// phases in SGPRs and writes them, with the CU it ran on, to p.stamps at the end.
#ifdef MPX_DIAG
#define MPX_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPX_STAMP_WRITE(p, t_start, t_pro, t_kend, t_epi)                                                        \
    if ((p).stamps && threadIdx.x == 0) {                                                                         \
        unsigned long long* o_ = (p).stamps + (size_t)blockIdx.x * 8;                                             \
        o_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                                   \
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   /* HW_ID | XCC_ID */    \
        o_[1] = t_start; o_[2] = t_pro; o_[3] = t_kend; o_[4] = t_epi; o_[5] = __builtin_amdgcn_s_memtime();        \
        o_[6] = __builtin_amdgcn_s_memrealtime();                                                                  \
    }
#else
#define MPX_STAMP(var)
#define MPX_STAMP_WRITE(p, a, b, c, d)
#endif

template <int TC_, int TP_, int NWR_, int NWC_, int NSW_, int NSX_, int MINB_ = 2>
struct ConvCfg {
    static constexpr int TC = TC_, TP = TP_, NWR = NWR_, NWC = NWC_, NSW = NSW_, NSX = NSX_;   // ring depths of W and X
    static constexpr int MINB = MINB_;              // workgroups per CU the register budget is sized for
    static constexpr int NW = NWR * NWC, NT = 64 * NW;
    static constexpr int CF = TC / NWR / 16;        // 16-row cout fragments per wave
    static constexpr int PF = TP / NWC / 16;        // 16-col pixel fragments per wave
    static constexpr int WSTAGE = TC * 128, XSTAGE = TP * 128;      // [hi rows x 64 B][lo rows x 64 B]
    static constexpr int XBASE = NSW * WSTAGE;
    static constexpr int WP = TC / 16, XP = TP / 16;            // 1-KiB pieces per plane
    static constexpr bool HALF_W = (2 * WP == NW);              // waves [0,NW/2) move W_hi, the rest W_lo
    static constexpr int WJ = HALF_W ? 1 : WP / NW;             // W pieces per wave per plane
    static constexpr int XJ = XP / NW;                          // X pieces per wave per plane
    static constexpr int LW = HALF_W ? 1 : 2 * WJ, LX = 2 * XJ; // DMA instructions per wave per W / X stage
    static constexpr int RING = NSW * WSTAGE + NSX * XSTAGE;
    static constexpr int EPI = TP * TC * 4;
    static constexpr int LDS = RING > EPI ? RING : EPI;
    // Per step a wave issues W(ks+NSW) and then X(ks+NSX).  At the top of step ks+1 stage ks+2 must have landed
    // (its fragments are read during that step): everything issued after W(ks+2) / X(ks+2) may stay in flight.
    static constexpr int WAIT_W = LX + (NSW - 2) * (LW + LX), WAIT_X = (NSX - 2) * (LW + LX);
    static constexpr int WAIT_STEP = WAIT_W < WAIT_X ? WAIT_W : WAIT_X;
    static constexpr int WAIT_PROLOGUE = (NSW - 1) * LW + (NSX - 1) * LX;
    static constexpr int DMA_FIRST = 1;                         // MFMA index after which the first DMA group is issued
    static_assert(HALF_W || WP % NW == 0, "W pieces must divide over the waves");
    static_assert(XP % NW == 0, "X pieces must divide over the waves");
    static_assert((TC / NWR) % 16 == 0 && (TP / NWC) % 16 == 0 && NSW >= 2 && NSX >= NSW, "tile shape");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // lgkmcnt(0): this wave's fragment reads of the previous step have returned before it enters the
    // barrier that frees their ring slot for the next DMA
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// POOL (the stem of the ImageNet ResNets with its 3x3 stride-2 pad-1 max pool in the same launch): a tile is not 256 consecutive
// output pixels but the 15 x 17 conv outputs (255 of the 256 slots) under a 7 x 8 block of POOLED pixels of one image -- pooled rows
// 7*ty .. 7*ty+6 need conv rows 14*ty-1 .. 14*ty+13, pooled columns 8*tx .. 8*tx+7 conv columns 16*tx-1 .. 16*tx+15 -- so neighbouring
// tiles recompute one conv row / column (15/14 x 17/16 = 1.14 x the MFMAs; the 56 x 56 pooled map is 8 x 7 tiles exactly) and the
// 4-byte conv output (3.2 MB per image written, then read again by the pool) never exists.  The K loop is
// unchanged; the epilogue takes the max over the fp32 tile in LDS.  Bit-identical to conv -> pool: the same accumulation per
// conv pixel, and rounding to hi + lo is monotonic, so max-then-round = round-then-max.
constexpr int POOL_PY = 7, POOL_PX = 8;                                    // pooled pixels per tile: rows, columns
constexpr int POOL_CY = 2 * POOL_PY + 1, POOL_CX = 2 * POOL_PX + 1;        // conv pixels under them

template <class C, bool DUAL = false, bool POOL = false>
__global__ __launch_bounds__(C::NT, C::MINB) void conv_f16x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource builtins are device-only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TC = C::TC, TP = C::TP, NSW = C::NSW, NSX = C::NSX, NW = C::NW, NT = C::NT;
    constexpr int CF = C::CF, PF = C::PF, WJ = C::WJ, XJ = C::XJ, WSTAGE = C::WSTAGE, XSTAGE = C::XSTAGE, XBASE = C::XBASE;
    constexpr int OFF_WHI = 0, OFF_WLO = TC * 64, OFF_XHI = 0, OFF_XLO = TP * 64;     // within a W stage / an X stage

    MPX_STAMP(t_start);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::NWC, wc = wave % C::NWC;

    // XCD-aware bijective remap: blocks that share an XCD (b % 8) walk a contiguous range of
    // logical tiles, cout tiles fastest, so the X tile of one pixel range stays in that XCD's L2.
    int L;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    const int mt = L / p.n_tiles_c;
    const int nt = L - mt * p.n_tiles_c;
    const int n0 = nt * TC;
    // POOL: tile mt = (image, ty, tx) over the pooled map; m0 = first pixel of the image
    const int pool_tny = POOL ? p.ho / 2 / POOL_PY : 1, pool_tnx = POOL ? p.wo / 2 / POOL_PX : 1;
    const int pool_n = mt / (pool_tny * pool_tnx);
    const int pool_ty = (mt - pool_n * pool_tny * pool_tnx) / pool_tnx;
    const int pool_tx = mt - pool_n * pool_tny * pool_tnx - pool_ty * pool_tnx;
    const int m0 = POOL ? pool_n * p.ho * p.wo : mt * TP;

    // ---- DMA addressing: buffer_load ... lds with wave-uniform descriptors -------------------------------
    // X descriptors start at the first image this tile touches, so a lane's byte offset is small (a few
    // images); W descriptors start at the tile's first cout row and the K step is the SGPR soffset.  Per step a
    // lane spends ~5 VALU per X piece (tap bounds test, one add, one select to the out-of-range offset that
    // makes the buffer unit return zeros) and none per W piece.
    const int nk = p.ktot >> 5;
    const int prow = lane >> 2;                                      // row within a 16-row piece
    const int src_q = ((lane & 3) ^ (((prow >> 3) & 1) << 1)) * 8;   // swizzled source chunk, in elements
    const int howo = p.ho * p.wo;
    const int n_first = m0 / howo;
    const int img_elems = p.hin * p.win * p.pix_stride;
    constexpr unsigned OOB = 0x80000000u;                            // >= num_records of every descriptor
    __amdgpu_buffer_rsrc_t x_rs_hi, x_rs_lo, w_rs_hi, w_rs_lo;
    {
        const int n_img = p.M / howo;
        const size_t rem = (size_t)(n_img - n_first) * img_elems * 2;
        const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
        x_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_hi + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
        x_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x_lo + (size_t)n_first * img_elems), 0, nrec, 0x00020000);
        const int wrec = TC * p.ktot * 2;
        w_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_hi + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
        w_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w_lo + (size_t)n0 * p.ktot), 0, wrec, 0x00020000);
    }
    int x_off0[XJ], x_iy0[XJ], x_ix0[XJ];
#pragma unroll
    for (int i = 0; i < XJ; ++i) {
        if (POOL) {
            // slot s of the tile = conv pixel (14*ty - 1 + s / 17, 16*tx - 1 + s % 17); slot 255 and pixels outside the map are dead
            const int sl = (i * NW + wave) * 16 + prow;
            const int sy = sl / POOL_CX, sx = sl - sy * POOL_CX;
            const int oy = 2 * POOL_PY * pool_ty - 1 + sy, ox = 2 * POOL_PX * pool_tx - 1 + sx;
            const bool ok = sl < POOL_CY * POOL_CX && oy >= 0 && oy < p.ho && ox >= 0 && ox < p.wo && m0 < p.M;
            x_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);
            x_ix0[i] = ox * p.stride - p.pad;
            x_off0[i] = (x_iy0[i] * p.win + x_ix0[i]) * p.pix_stride * 2 + src_q * 2;
            continue;
        }
        const int m = m0 + (i * NW + wave) * 16 + prow;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        x_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
        x_ix0[i] = ox * p.stride - p.pad;
        x_off0[i] = (((n - n_first) * p.hin + x_iy0[i]) * p.win + x_ix0[i]) * p.pix_stride * 2 + src_q * 2;
    }
    // DUAL: second pixel operand (the block input under the downsample conv): descriptor pair of its own and one
    // byte offset per X piece (1x1, no padding: only m >= M is out of range)
    __amdgpu_buffer_rsrc_t x2_rs_hi = x_rs_hi, x2_rs_lo = x_rs_lo;
    int x2_off0[DUAL ? XJ : 1];
    if (DUAL) {
        const int n_img = p.M / howo;
        const int img2 = p.hin2 * p.win2 * p.pix_stride2;
        const size_t rem = (size_t)(n_img - n_first) * img2 * 2;
        const int nrec = rem > 0x7fffffffu ? 0x7fffffff : (int)rem;
        x2_rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_hi + (size_t)n_first * img2), 0, nrec, 0x00020000);
        x2_rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_lo + (size_t)n_first * img2), 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < XJ; ++i) {
            const int m = m0 + (i * NW + wave) * 16 + prow;
            const int n = m / howo;
            const int rem2 = m - n * howo;
            const int oy = rem2 / p.wo;
            const int ox = rem2 - oy * p.wo;
            const int off = (((n - n_first) * p.hin2 + oy * p.stride2) * p.win2 + ox * p.stride2) * p.pix_stride2 * 2 + src_q * 2;
            x2_off0[i] = off | ((p.M - 1 - m) & (int)OOB);
        }
    }
    // W pieces of this wave: piece (j*NW + wave) of each plane; HALF_W: one piece of one plane.  The planes are piece-major
    // (w_packed_index): a lane reads byte lane*16 of the piece, and the piece's position -- (piece * K/32 + ks) KiB -- is
    // wave-uniform, i.e. the SGPR soffset: no VGPR per piece.
    const int w_lane = lane * 16;
    int w_piece[WJ];                 // wave-uniform byte offset of piece j at K step 0
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int piece = C::HALF_W ? (wave % (NW / 2)) : (j * NW + wave);
        w_piece[j] = piece * 16 * p.ktot * 2;
    }

    // `live` = false (a step past the end of K) keeps the ring and the vmcnt bookkeeping in shape: the X pieces
    // read out of range (zeros), the W pieces read whatever follows the row (never used).
    auto stage_w = [&](int buf, int ks) {
        char* sb = smem + buf * WSTAGE;
        const int soff = ks * 1024;
        // a step past the end of K only keeps the vmcnt bookkeeping in shape: an out-of-range offset makes the
        // buffer unit return zeros without a memory access, so the drain in front of the epilogue is short
        const int dead = ks < nk ? 0 : (int)OOB;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (C::HALF_W) {
                if (wave < NW / 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + wave * 1024), 16, w_lane | dead, soff + w_piece[j], 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + wave * 1024), 16, w_lane | dead, soff + w_piece[j], 0, 0);
            } else {
                const int d = (j * NW + wave) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_hi, MPX_LDS_PTR(sb + OFF_WHI + d), 16, w_lane | dead, soff + w_piece[j], 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs_lo, MPX_LDS_PTR(sb + OFF_WLO + d), 16, w_lane | dead, soff + w_piece[j], 0, 0);
            }
        }
    };
    auto stage_x = [&](int i, int buf, int ky, int kx, int c0, bool live) {
        char* sb = smem + XBASE + buf * XSTAGE;
        const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
        const int delta = ((ky * p.win + kx) * p.pix_stride + c0) * 2;      // wave-uniform
        // Out-of-range taps (and steps past the end of K) set the offset's sign bit, which is >= num_records, so
        // the buffer unit returns zeros.  Pure bit arithmetic on purpose: hipcc lowers a per-lane `ok ? a : b` to an
        // s_and_saveexec region, and an MFMA that the scheduler drops into such a region would run with EXEC
        // partly off and leave those lanes' results unwritten.
        const int voff = (x_off0[i] + delta) | ((iy | (p.hin - 1 - iy) | ix | (p.win - 1 - ix)) & (int)OOB) | (live ? 0 : (int)OOB);
        const int d = (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_hi, MPX_LDS_PTR(sb + OFF_XHI + d), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs_lo, MPX_LDS_PTR(sb + OFF_XLO + d), 16, voff, 0, 0, 0);
    };
    auto stage_x2 = [&](int i, int buf, int cb, bool live) {       // DUAL: channels [cb, cb+32) of the second operand
        char* sb = smem + XBASE + buf * XSTAGE;
        const int voff = (x2_off0[DUAL ? i : 0] + cb * 2) | (live ? 0 : (int)OOB);
        const int d = (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rs_hi, MPX_LDS_PTR(sb + OFF_XHI + d), 16, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rs_lo, MPX_LDS_PTR(sb + OFF_XLO + d), 16, voff, 0, 0, 0);
    };
    f4 acc[CF][PF];
#pragma unroll
    for (int a = 0; a < CF; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    int ky = 0, kx = 0, c0 = 0;     // coordinates of the NEXT step to stage
    int cb = 0;                     // DUAL: channel offset of the next stage within the second operand
    auto advance = [&]() {      // branch-free (the K step must stay one basic block for the scheduler)
        c0 += 32;
        const bool wc0 = (c0 == p.k_per_tap);
        c0 = wc0 ? 0 : c0;
        kx += wc0 ? 1 : 0;
        const bool wkx = (kx == p.kw);
        kx = wkx ? 0 : kx;
        ky += wkx ? 1 : 0;
    };

    const int lrow = lane & 15;
    const int qsw = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16;
    const int a_off = (wr * (TC / C::NWR) + lrow) * 64 + qsw;
    const int b_off = (wc * (TP / C::NWC) + lrow) * 64 + qsw;

    struct Frags {
        h8 a_hi[CF], a_lo[CF], b_hi[PF], b_lo[PF];
    };
    constexpr int NF = 2 * (CF + PF);                   // fragment reads per step
    constexpr int NM = 3 * CF * PF;                     // MFMAs per step
    auto load_frag = [&](int wslot, int xslot, Frags& f, int j) {   // fragment j: a_hi[..] a_lo[..] b_hi[..] b_lo[..]
        const char* sw = smem + wslot * WSTAGE;
        const char* sx = smem + XBASE + xslot * XSTAGE;
        if (j < CF) f.a_hi[j] = *(const h8*)(sw + OFF_WHI + a_off + j * 1024);
        else if (j < 2 * CF) f.a_lo[j - CF] = *(const h8*)(sw + OFF_WLO + a_off + (j - CF) * 1024);
        else if (j < 2 * CF + PF) f.b_hi[j - 2 * CF] = *(const h8*)(sx + OFF_XHI + b_off + (j - 2 * CF) * 1024);
        else f.b_lo[j - 2 * CF - PF] = *(const h8*)(sx + OFF_XLO + b_off + (j - 2 * CF - PF) * 1024);
    };
    auto load_frags = [&](int wslot, int xslot, Frags& f) {
#pragma unroll
        for (int j = 0; j < NF; ++j) load_frag(wslot, xslot, f, j);
    };
    auto mfma_one = [&](const Frags& f, int i) {        // flat order (a, term, b): neighbours hit different accumulators
        const int a = i / (3 * PF), r = i % (3 * PF), term = r / PF, b = r % PF;
        if (term == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
        else if (term == 1) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
    };
    auto mfma_row = [&](const Frags& f, int a) {        // PF*3 MFMAs: one 16-channel row of fragments
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_lo[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_lo[a], f.b_hi[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a_hi[a], f.b_hi[b], acc[a][b], 0, 0, 0);
        }
    };
    auto mfma_all = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < CF; ++a) mfma_row(f, a);
    };

    // Pipeline: the fragment registers are one more stage.  While the MFMAs of step ks run from registers, the
    // fragments of step ks+1 are read from the rings and the DMAs W(ks+2 .. ks+NSW), X(ks+2 .. ks+NSX) are in
    // flight.  W and X have rings of their own: X usually comes from HBM / Infinity Cache and gets the deeper one.
    // prologue: W(0..NSW-1), X(0..NSX-1) issued stage by stage, W before X (dummies past the end of K)
#pragma unroll
    for (int s = 0; s < NSX; ++s) {
        if (s < NSW) stage_w(s, s);
#pragma unroll
        for (int i = 0; i < XJ; ++i) stage_x(i, s, ky, kx, c0, s < nk);
        advance();
    }
    wait_vmcnt<C::WAIT_PROLOGUE>();
    __builtin_amdgcn_s_barrier();
    MPX_STAMP(t_pro);
    Frags fa, fb;
    load_frags(0, 0, fa);

    int wslot = 0, xslot = 0;       // ring slots of step ks
    // A step that has a successor: frags of ks are in `cur`; leaves frags of ks+1 in `nxt`.  Branch-free, and
    // hand-scheduled: every instruction sits between sched_barrier(0) fences, because the waves of a workgroup
    // leave the barrier in lockstep and an in-order wave that meets a burst (16 fragment reads from 8 waves at
    // once, or a clump of DMA address arithmetic) cannot issue MFMAs behind it.  The order per step is
    //   MFMA 0 (hipcc guards its operands with an lgkmcnt(0) that finds nothing outstanding here),
    //   then one fragment read of step ks+1 after every second MFMA,
    //   with the DMA groups W(ks+NSW), X(ks+NSX) piece by piece after MFMAs 1, 5, 9, ... (early, so that they have
    //   the rest of the step to land).
    auto full_step = [&](auto seg_tag, int ks, const Frags& cur, Frags& nxt) {
        constexpr bool SEGB = decltype(seg_tag)::value;     // this step stages from the second operand (DUAL)
        // own pieces of stage ks+1 landed (younger ones stay in flight), and -- lgkmcnt(0) -- this wave's reads of
        // the slots of step ks returned; the barrier then frees them for W(ks+NSW) and X(ks+NSX)
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<C::WAIT_STEP>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nw = (wslot + 1 == NSW) ? 0 : wslot + 1;
        const int nx = (xslot + 1 == NSX) ? 0 : xslot + 1;
        const bool live = ks + NSX < nk;
        constexpr int G = 1 + XJ;                       // DMA groups: W, X piece 0, X piece 1, ...
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if ((i & 1) == 0 && i / 2 < NF) {
                load_frag(nw, nx, nxt, i / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (i == C::DMA_FIRST + 4 * g) {      // early in the step: the pieces need the rest of it to land
                    if (g == 0) stage_w(wslot, ks + NSW);
                    else if (SEGB) stage_x2(g - 1, xslot, cb, live);
                    else stage_x(g - 1, xslot, ky, kx, c0, live);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (SEGB) cb += 32;
        else advance();
        wslot = nw;
        xslot = nx;
    };
    typedef std::integral_constant<bool, false> SegA;
    typedef std::integral_constant<bool, DUAL> SegLast;     // the operand the last K steps stage from
    int ks = 0;
    if (DUAL) {
        // steps whose DMA stage (ks + NSX) still belongs to the first operand; the host guarantees that k1/32 - NSX is
        // even and >= 0, so the fa / fb ping-pong stays in phase
        const int n_a = (p.k1 >> 5) - NSX;
        for (; ks < n_a; ks += 2) {
            full_step(SegA{}, ks, fa, fb);
            full_step(SegA{}, ks + 1, fb, fa);
        }
    }
    for (; ks + 2 < nk; ks += 2) {
        full_step(SegLast{}, ks, fa, fb);
        full_step(SegLast{}, ks + 1, fb, fa);
    }
    if (ks + 2 == nk) {
        full_step(SegLast{}, ks, fa, fb);
        mfma_all(fb);
    } else {
        mfma_all(fa);
    }
    wait_vmcnt<0>();                // the trailing dummy DMAs must land before the epilogue reuses the LDS
    MPX_STAMP(t_kend);

    // ---- epilogue ----
    // Phase 0: prefetch the residual rows this thread will own in phase 2 (whole 16-B chunks).
    constexpr int GPP = TC / 8;                 // threads per pixel row (8 channels each)
    constexpr int PPI = NT / GPP;               // pixels per phase-2 iteration
    constexpr int ITERS = TP / PPI;
    constexpr int RP = TC * 4;                  // fp32 tile row pitch in bytes
    const int g = tid % GPP;
    const int prow2 = tid / GPP;
    const int co8 = n0 + g * 8;
    const bool co_ok = co8 < p.cout;
    h8 rh[ITERS], rl[ITERS];
    if (p.r_hi) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int pix = m0 + it * PPI + prow2;
            if (co_ok && pix < p.M) {
                const size_t o = (size_t)pix * p.cout + co8;
                rh[it] = __builtin_nontemporal_load((const h8*)(p.r_hi + o));    // streamed once: keep X/W in L2
                rl[it] = __builtin_nontemporal_load((const h8*)(p.r_lo + o));
            }
        }
    }
    __syncthreads();                            // all fragment reads of the last stage are done
    // Phase 1: acc*scale+shift -> fp32 tile [pixel][cout] in LDS; D row = cout (lane>>4)*4+reg, col = pixel.
#pragma unroll
    for (int a = 0; a < CF; ++a) {
        const int col = wr * (TC / C::NWR) + a * 16 + (lane >> 4) * 4;     // cout within the tile
        const f4 sc = *(const f4*)(p.scale + n0 + col);
        const f4 sh = *(const f4*)(p.shift + n0 + col);
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            const int pl = wc * (TP / C::NWC) + b * 16 + lrow;              // pixel within the tile
            const f4 v = acc[a][b] * sc + sh;
            *(f4*)(smem + pl * RP + (((col >> 2) ^ (pl & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    MPX_STAMP(t_epi);
    if (POOL) {
        // Phase 2, pooled: one thread = 8 consecutive channels of one POOLED pixel = max over its 3 x 3 conv slots (the slots
        // outside the map -- conv row / column -1 of the first tiles -- do not take part: max-pool padding)
        constexpr int NPOOL = POOL_PY * POOL_PX;
        const int po = p.ho / 2;
#pragma unroll
        for (int it = 0; it < (NPOOL + PPI - 1) / PPI; ++it) {
            const int q = it * PPI + prow2;
            if (!(co_ok && q < NPOOL && m0 < p.M)) continue;
            const int py = q / POOL_PX, px = q - py * POOL_PX;
            float best[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                if (2 * POOL_PY * pool_ty - 1 + 2 * py + dy < 0) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (2 * POOL_PX * pool_tx - 1 + 2 * px + dx < 0) continue;
                    const int sl = (2 * py + dy) * POOL_CX + 2 * px + dx;
                    const f4 v0 = *(const f4*)(smem + sl * RP + (((2 * g) ^ (sl & 7)) << 4));
                    const f4 v1 = *(const f4*)(smem + sl * RP + (((2 * g + 1) ^ (sl & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        best[j] = fmaxf(best[j], v0[j]);
                        best[4 + j] = fmaxf(best[4 + j], v1[j]);
                    }
                }
            }
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // what conv -> pool stores: the conv value rounded to hi + lo, and THAT number split again (on a tie the second split
                // can pick the other neighbour: same value, other hi / lo pair); rounding is monotonic, so max commutes with it
                half_t hi, lo;
                split_f32(p.relu ? fmaxf(best[j], 0.f) : best[j], hi, lo);
                split_f32((float)hi + (float)lo, hi, lo);
                oh[j] = hi;
                ol[j] = lo;
            }
            const size_t o = (((size_t)pool_n * po + POOL_PY * pool_ty + py) * po + POOL_PX * pool_tx + px) * p.cout + co8;
            __builtin_nontemporal_store(oh, (h8*)(p.y_hi + o));
            __builtin_nontemporal_store(ol, (h8*)(p.y_lo + o));
        }
        MPX_STAMP_WRITE(p, t_start, t_pro, t_kend, t_epi);
        return;
    }
    // Phase 2: one thread = 8 consecutive channels of one pixel.
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int pl = it * PPI + prow2;
        const int pix = m0 + pl;
        if (!(co_ok && pix < p.M)) continue;
        const f4 v0 = *(const f4*)(smem + pl * RP + (((2 * g) ^ (pl & 7)) << 4));
        const f4 v1 = *(const f4*)(smem + pl * RP + (((2 * g + 1) ^ (pl & 7)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.r_hi) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rh[it][j] + (float)rl[it][j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const size_t o = (size_t)pix * p.cout + co8;
        if (p.y_f32) {
            *(f4*)(p.y_f32 + o) = (f4){v[0], v[1], v[2], v[3]};
            *(f4*)(p.y_f32 + o + 4) = (f4){v[4], v[5], v[6], v[7]};
        } else {
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t hi, lo;
                split_f32(v[j], hi, lo);
                oh[j] = hi;
                ol[j] = lo;
            }
            __builtin_nontemporal_store(oh, (h8*)(p.y_hi + o));
            __builtin_nontemporal_store(ol, (h8*)(p.y_lo + o));
        }
    }
    MPX_STAMP_WRITE(p, t_start, t_pro, t_kend, t_epi);
#endif
}

// Tile variants selectable per layer (mpx_set_conv_tile).
typedef ConvCfg<128, 256, 2, 4, 3, 3> ConvTile0;   // 8 waves, 144 KB LDS, 1 workgroup / CU
typedef ConvCfg<64, 256, 1, 4, 2, 2> ConvTile1;    // cout <= 64: 4 waves side by side, each 64 cout x 64 pixels; 80 KB LDS, 2 workgroups / CU
typedef ConvCfg<128, 128, 2, 2, 2, 2> ConvTile2;   // 4 waves, 2-deep rings, 64 KB LDS, 2 workgroups / CU
typedef ConvCfg<64, 192, 1, 4, 2, 2> ConvTile4;    // cout <= 64: 64 KB LDS, 2 workgroups / CU
#ifdef MPX_EXPERIMENTAL                            // probe builds only (no layer class's default): ids 3 and 5
typedef ConvCfg<128, 128, 2, 2, 2, 3> ConvTile3;   // as tile 2 with a 3-deep X ring (80 KB): measured 5-11 % slower (2 x 80 KB no longer co-reside)
typedef ConvCfg<64, 128, 2, 2, 2, 2, 3> ConvTile5; // cout <= 64: 48 KB LDS, 3 workgroups / CU: ties tile 2 at best, 3-20 % slower elsewhere
#endif
// id 6 = the 3x3 patch kernel (mpx_conv3p.h), 9 = mpx_conv256.h, 10 = mpx_convx.h
// id 7: the 128x128 tile cut into 8 waves of 32 cout x 64 pixels: 124 VGPRs, so two workgroups = 16 waves per CU (4 per
// SIMD).  Each wave issues half the LDS-DMA pieces and half the MFMAs of a tile-2 wave; with four waves per SIMD one
// wave's DMA issue and barrier waits are covered by three others.
typedef ConvCfg<128, 128, 4, 2, 2, 2, 4> ConvTile7;
// (tile 7 with a 3-deep X ring, 80 KB, was measured in round 2: two workgroups still co-reside and the K loop's share of a
// workgroup's life falls from 47 % to 40 %, but the layer time does not move: DESIGN.md 5)

}  // namespace mpx
