// mpx_api.hip -- C-ABI (include/mpx.h) + host-side engine: topology, weight packing, workspace,
// launch sequencing.  Compiled for gfx950 only:  hipcc --offload-arch=gfx950 -shared -fPIC.
#include "../../include/mpx.h"
#include "mpx_kernels.h"
#include "mpx_conv3p.h"
#include "mpx_conv3pp.h"
#include "mpx_conv256.h"
#include "mpx_conv256p.h"
#include "mpx_convx.h"
#include "mpx_convw.h"
#include "mpx_btail.h"
#include "mpx_stemtab.h"
#ifdef MPX_EXPERIMENTAL
// Kernels that were measured in the network and did not become any layer class's default: tile ids 3 and 5 (other
// shapes of the generic kernel), 8 (persistent kernel with a register epilogue) and 11 (pixel-stationary expanding 1x1 kernel).
// They live under tools/probes/experimental/ and are compiled into probe builds only (tools/probes/build_experimental.sh).
#include "../../tools/probes/experimental/mpx_convp.h"
#include "../../tools/probes/experimental/mpx_convs.h"
#endif

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <initializer_list>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

using namespace mpx;

namespace {

constexpr int kActBufs = 4;
constexpr int kStemK = 7;
constexpr int kProfilePairs = 4096;
constexpr size_t kActElemsPerImage = 112 * 112 * 64;   // ImageNet: largest activation (stem output, = 56*56*256)
constexpr int kSmallCPad = 32;                         // small nets: channels are stored padded to a multiple of 32

enum OpKind { OP_CONV = 0, OP_MAXPOOL = 1, OP_AVGPOOL = 2, OP_HEAD = 3, OP_AVGPAD = 4, OP_BTAIL = 5 };
enum Buf { BUF_INPUT = -1, BUF_POOL = -2, BUF_NONE = -3, BUF_STEM = -4 };   // BUF_STEM: the pooled stem output (planes of its own)

struct ConvLayer {
    mpx_conv_desc d;
    half_t* w_hi = nullptr;
    half_t* w_lo = nullptr;
    float* scale = nullptr;
    float* shift = nullptr;
    bool loaded = false;
    bool is_fc = false;
    bool is_stem = false;
    bool has_bias = false;  // the reference module is nn.Conv2d(bias=True) (small nets): state_dict has <name>.bias
    int cin_pad = 0;        // channels per pixel of the input planes (= cin except for the small nets, which pad to 32)
    int cout_store = 0;     // channels per pixel of the output planes (row pitch and store bound)
    int tile = 0;           // ConvTile<n> variant (mpx_set_conv_tile)
    // downsample fusion (bottleneck blocks): the block's last 1x1 conv ("main") and its downsample 1x1 conv ("ds")
    // run as ONE launch over the K-concatenation [W3 * s3/s | Wds * sds/s] (build_fused)
    int fuse_partner = -1;  // main -> ds layer index, ds -> main layer index
    bool fuse_main = false;
    half_t* fw_hi = nullptr;    // main only: fused planes [cout_pad][k1 + k2]
    half_t* fw_lo = nullptr;
    float* fscale = nullptr;
    float* fshift = nullptr;
    bool fused_loaded = false;
    std::vector<float> hw, hgamma, hbeta, hmean, hvar;   // host copies of a fusable layer's tensors (until both are in)
    float heps = 0.f;
};

struct Op {
    int kind;
    int conv;       // layer index for OP_CONV
    int in, out, res;
    int hin, c;     // pools
    int in2;        // fused main conv: buffer of the block input (the downsample branch's operand), else BUF_NONE
    int z = BUF_NONE;   // OP_BTAIL: buffer of the next block's conv1 output; `conv` = index into mpx_engine::tails,
                        // in = t1, res = block input / identity, out = block output
};

// A 64-channel bottleneck block whose tail runs as ONE launch (mpx_btail.h): conv2 -> conv3 (+ identity or K-concatenated
// downsample branch) -> the next block's conv1.  Weights: conv2's and next1's packed planes as they are; conv3's (or the fused
// conv3 | downsample planes') with columns [0,64) K-permuted into planes of its own.
struct TailBlock {
    int c1 = -1, c2 = -1, c3 = -1, ds = -1, next1 = -1;     // c1: the block's own conv1 (runs inside the launch when the block has a downsample branch)
    half_t* w3p_hi = nullptr;
    half_t* w3p_lo = nullptr;
    bool ready = false;
};

struct ProfRec {
    hipEvent_t t0, t1;
    int kind;
    int conv;
};

}  // namespace

struct mpx_engine {
    int arch = 0, max_batch = 0, device = 0;
    int num_cus = 256;
    // geometry of the network family: ImageNet ResNets (224x224x3, padded NHWC4 staging for the 7x7 stem, 1000 classes) or
    // the reference's two small networks (28x28x1 / 32x32x3, staging [B][H][W][32] read by a generic 3x3 conv, 10 classes)
    int img = MPX_IMG, in_ch = 3, ncls = MPX_NUM_CLASSES, logit_pitch = MPX_NUM_CLASSES;
    bool small = false;
    size_t act_elems_per_image = 0;
    float* k0_scratch = nullptr;    // small nets: f32[2 + 4096 + max_batch]: image min, max-min, per-superpixel max, per-mask max
    bool bottleneck = false;
    int feat = 0;
    std::vector<ConvLayer> convs;
    std::vector<Op> ops;
    std::vector<Op> ops_bt;         // the same network with layer1's block tails as OP_BTAIL (empty if the arch has none)
    std::vector<TailBlock> tails;
    int ops_bt_x = 0;               // buffer of the last block's output in ops_bt
    bool fuse_bt = true;            // mpx_forward uses ops_bt when every tail is ready (mpx_set_fusion bit 1)
    char* arena = nullptr;
    size_t arena_bytes = 0;
    char* tab_arena = nullptr;      // the stem table of one image (CSR, heavy-pixel list, bit planes): allocated by the FIRST mpx_stem_table_build
    size_t tab_arena_bytes = 0;
    half_t* in_hi = nullptr;
    half_t* in_lo = nullptr;
    half_t* act_hi[kActBufs] = {};
    half_t* act_lo[kActBufs] = {};
    half_t* pool_hi = nullptr;
    half_t* pool_lo = nullptr;
    float* logits = nullptr;
    float* seg_scratch = nullptr;   // f32[4096] per-superpixel counts (K5)
    // the stem by superposition (mpx_stemtab.h; ImageNet ResNets): pooled stem output planes [max_batch][56][56][64] of their own (written by
    // the stem + pool launch or by mpx_stem_table_apply), ONE image's table, the bit planes of a staging call's mask rows
    half_t* stem_hi = nullptr;
    half_t* stem_lo = nullptr;
    float* stem_w32 = nullptr;      // [147][64] fp32 stem weights, tap-major
    float* stem_s32 = nullptr;      // [64] gamma / sqrt(var + eps)
    float* stem_t32 = nullptr;      // [64] beta - mean * scale
    int* tab_cnt = nullptr;
    int* tab_off = nullptr;
    int* tab_lab = nullptr;
    int* tab_heavy = nullptr;       // [56 * 56 + 1]: the heavy pooled pixels of the table in place, last element = their number
    float* tab_vec = nullptr;
    unsigned* tab_bits = nullptr;   // [4096][ceil(max_batch / 32) + 1]
    size_t tab_sizes[4] = {0, 0, 0, 0};   // bytes of one int vector / lab / vec / bits (mpx_create), taken from tab_arena on first use
    int tab_S = -1;                 // S of the table in place (-1: none)
    bool stem_w_loaded = false;
    std::vector<uint8_t> slot_src;  // per input slot: 0 = never staged, 1 = K0 (input staging), 2 = stem table (pooled planes)
    std::string err;
    bool in_forward = false;
    unsigned last_kernels = 0;      // bit t: the last conv call launched the kernel of tile id t (mpx_last_conv_kernels)
    bool fuse_ds = true;    // mpx_forward runs a block's last conv and its downsample conv as one launch (mpx_set_fusion)
    bool fuse_pool = true;  // ... and the ImageNet stem conv with its 3x3 stride-2 max pool (mpx_stem_conv_maxpool)
    bool prof_on = false;
    std::vector<ProfRec> prof_pool;
    int prof_used = 0;
    hipStream_t prof_stream = nullptr;
#ifdef MPX_DIAG
    unsigned long long* stamps = nullptr;
#endif
};

namespace {

int fail(mpx_engine* h, int code, const char* fmt, ...) {
    if (h) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        h->err = buf;
    }
    return code;
}

#define MPX_HIP(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((h), (int)e_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                        __FILE__, __LINE__);                                                     \
    } while (0)

// every entry point selects the engine's device; mpx_forward does it ONCE and runs its ~110 launches without the call
#define MPX_SET_DEVICE(h)                                   \
    do {                                                    \
        if (!(h)->in_forward) MPX_HIP((h), hipSetDevice((h)->device)); \
    } while (0)

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void set_name(char* dst, const std::string& s) {
    std::snprintf(dst, 48, "%s", s.c_str());
}

int default_tile(const mpx_conv_desc& d);

// torchvision ResNet topology (models/resnet.py, un-vendored): conv list and op list.
int build_topology_small(mpx_engine* h);

int build_topology(mpx_engine* h) {
    if (h->arch == MPX_ARCH_MNIST_NET || (h->arch > MPX_ARCH_CIFAR_RESNET && h->arch < MPX_ARCH_CIFAR_RESNET + 1000))
        return build_topology_small(h);
    h->act_elems_per_image = kActElemsPerImage;
    int depths[4];
    switch (h->arch) {
        case 18: depths[0] = 2; depths[1] = 2; depths[2] = 2; depths[3] = 2; h->bottleneck = false; break;
        case 34: depths[0] = 3; depths[1] = 4; depths[2] = 6; depths[3] = 3; h->bottleneck = false; break;
        case 50: depths[0] = 3; depths[1] = 4; depths[2] = 6; depths[3] = 3; h->bottleneck = true; break;
        case 101: depths[0] = 3; depths[1] = 4; depths[2] = 23; depths[3] = 3; h->bottleneck = true; break;
        case 152: depths[0] = 3; depths[1] = 8; depths[2] = 36; depths[3] = 3; h->bottleneck = true; break;
        default: return MPX_E_ARG;
    }
    const int exp = h->bottleneck ? 4 : 1;
    auto add_conv = [&](const std::string& name, const std::string& bn, int cin, int cout, int k, int stride,
                        int pad, int hin, int relu, int residual) {
        ConvLayer L;
        std::memset(&L.d, 0, sizeof L.d);
        set_name(L.d.name, name);
        set_name(L.d.bn_name, bn);
        L.d.cin = cin; L.d.cout = cout; L.d.ksize = k; L.d.stride = stride; L.d.pad = pad;
        L.d.hin = hin; L.d.hout = (hin + 2 * pad - k) / stride + 1;
        L.d.relu = relu; L.d.residual = residual;
        L.is_stem = (cin == 3);
        L.cin_pad = cin;
        L.cout_store = cout;
        L.d.k_packed = L.is_stem ? kStemK * 32 : k * k * cin;
        L.d.cout_pad = (int)round_up(cout, 128);
        L.tile = default_tile(L.d);
        h->convs.push_back(L);
        return (int)h->convs.size() - 1;
    };
    auto add_op = [&](int kind, int conv, int in, int out, int res, int hin, int c, int in2 = BUF_NONE) {
        Op o{kind, conv, in, out, res, hin, c, in2};
        h->ops.push_back(o);
    };
    auto pick = [&](std::initializer_list<int> busy) {
        for (int b = 0; b < kActBufs; ++b) {
            bool used = false;
            for (int u : busy) used |= (u == b);
            if (!used) return b;
        }
        return -100;
    };

    int c = add_conv("conv1", "bn1", 3, 64, 7, 2, 3, 224, 1, 0);
    add_op(OP_CONV, c, BUF_INPUT, 0, BUF_NONE, 0, 0);
    add_op(OP_MAXPOOL, -1, 0, BUF_STEM, BUF_NONE, 112, 64);      // the pooled planes have a buffer of their own: nothing overwrites them
    int X = BUF_STEM, cin = 64, hcur = 56;
    const int widths[4] = {64, 128, 256, 512};
    struct BlockRec { int c1, c2, c3, ds, hin; };
    std::vector<BlockRec> blocks;       // bottleneck blocks in forward order (the block-tail plan below is built from them)
    for (int s = 0; s < 4; ++s) {
        const int w = widths[s];
        for (int b = 0; b < depths[s]; ++b) {
            const int stride = (b == 0 && s > 0) ? 2 : 1;
            const std::string p = "layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
            const bool ds = (b == 0) && (stride != 1 || cin != w * exp);
            const int hout = hcur / stride;
            if (!h->bottleneck) {
                const int T1 = pick({X});
                c = add_conv(p + "conv1", p + "bn1", cin, w, 3, stride, 1, hcur, 1, 0);
                add_op(OP_CONV, c, X, T1, BUF_NONE, 0, 0);
                int res = X, T2 = -100;
                if (ds) {
                    T2 = pick({X, T1});
                    c = add_conv(p + "downsample.0", p + "downsample.1", cin, w * exp, 1, stride, 0, hcur, 0, 0);
                    add_op(OP_CONV, c, X, T2, BUF_NONE, 0, 0);
                    res = T2;
                }
                const int O = pick({X, T1, T2});
                c = add_conv(p + "conv2", p + "bn2", w, w, 3, 1, 1, hout, 1, 1);
                add_op(OP_CONV, c, T1, O, res, 0, 0);
                X = O;
            } else {
                BlockRec R{-1, -1, -1, -1, hcur};
                const int T1 = pick({X});
                R.c1 = c = add_conv(p + "conv1", p + "bn1", cin, w, 1, 1, 0, hcur, 1, 0);
                add_op(OP_CONV, c, X, T1, BUF_NONE, 0, 0);
                const int T2 = pick({X, T1});
                R.c2 = c = add_conv(p + "conv2", p + "bn2", w, w, 3, stride, 1, hcur, 1, 0);
                add_op(OP_CONV, c, T1, T2, BUF_NONE, 0, 0);
                int res = X, dsc = -1;
                if (ds) {
                    const int T3 = pick({X, T1, T2});
                    R.ds = dsc = add_conv(p + "downsample.0", p + "downsample.1", cin, w * exp, 1, stride, 0, hcur, 0, 0);
                    add_op(OP_CONV, dsc, X, T3, BUF_NONE, 0, 0);
                    res = T3;
                }
                R.c3 = c = add_conv(p + "conv3", p + "bn3", w, w * exp, 1, 1, 0, hout, 1, 1);
                add_op(OP_CONV, c, T2, T1, res, 0, 0, ds ? X : BUF_NONE);   // T1 is dead after conv2
                if (ds) {
                    h->convs[c].fuse_partner = dsc; h->convs[c].fuse_main = true;
                    h->convs[dsc].fuse_partner = c;
                }
                blocks.push_back(R);
                X = T1;
            }
            cin = w * exp;
            hcur = hout;
        }
    }
    // Block-tail plan (mpx_btail.h): a block whose conv2 is 64 -> 64 3x3 stride 1 on a 56x56 map, whose conv3 is 64 -> 256 and
    // whose successor starts with a 1x1 stride-1 conv 256 -> 64 / 128 (layer1 of ResNet-50 / 101 / 152) runs conv2, conv3 (+ the
    // identity or the K-concatenated downsample branch) and the successor's conv1 as ONE launch.  Four distinct buffers are live
    // across such a launch (block input, t1, block output, next t1): other workgroups still read t1's halo while this one writes.
    {
        auto tail_ok = [&](size_t k) {
            if (k + 1 >= blocks.size()) return false;
            const mpx_conv_desc& d2 = h->convs[blocks[k].c2].d;
            const mpx_conv_desc& d3 = h->convs[blocks[k].c3].d;
            const mpx_conv_desc& n1 = h->convs[blocks[k + 1].c1].d;
            if (!(d2.cin == BT_MID && d2.cout == BT_MID && d2.ksize == 3 && d2.stride == 1 && d2.hin == 56)) return false;
            if (!(d3.cin == BT_MID && d3.cout == BT_OUT)) return false;
            if (blocks[k].ds >= 0) {
                const mpx_conv_desc& dd = h->convs[blocks[k].ds].d;
                if (!(dd.cin == BT_MID && dd.stride == 1 && dd.ksize == 1)) return false;
            }
            return n1.cin == BT_OUT && n1.ksize == 1 && n1.stride == 1 && n1.hin == 56 && (n1.cout == 64 || n1.cout == 128) &&
                   !(blocks[k].ds >= 0 && n1.cout != 64);
        };
        bool any = false;
        for (size_t k = 0; k < blocks.size(); ++k) any |= tail_ok(k);
        if (any) {
            std::vector<Op>& out = h->ops_bt;
            out.push_back(h->ops[0]);       // stem conv, max pool
            out.push_back(h->ops[1]);
            int Xb = BUF_STEM, T1c = -100;         // T1c: buffer of a t1 that the previous tail launch has already written
            for (size_t k = 0; k < blocks.size(); ++k) {
                const BlockRec& R = blocks[k];
                int T1 = T1c;
                // a tail with a downsample branch (layer1.0) runs its block's own conv1 too, on the patch of the block input: no conv1
                // launch, no t1 tensor (in = BUF_NONE)
#ifdef BT_NO_HEAD       // probe builds (tools/ab_lib.sh): layer1.0.conv1 stays a launch of its own
                const bool whole = false;
#else
                const bool whole = tail_ok(k) && R.ds >= 0 && T1 < 0;
#endif
                if (T1 < 0 && !whole) {
                    T1 = pick({Xb});
                    out.push_back(Op{OP_CONV, R.c1, Xb, T1, BUF_NONE, 0, 0, BUF_NONE});
                }
                T1c = -100;
                if (tail_ok(k)) {
                    TailBlock tb;
                    tb.c1 = R.c1; tb.c2 = R.c2; tb.c3 = R.c3; tb.ds = R.ds; tb.next1 = blocks[k + 1].c1;
                    h->tails.push_back(tb);
                    if (whole) T1 = BUF_NONE;
                    const int O = pick({Xb, T1}), Z = pick({Xb, T1, O});
                    Op o{OP_BTAIL, (int)h->tails.size() - 1, T1, O, Xb, 0, 0, BUF_NONE};
                    o.z = Z;
                    out.push_back(o);
                    Xb = O;
                    T1c = Z;
                    continue;
                }
                const int T2 = pick({Xb, T1});
                out.push_back(Op{OP_CONV, R.c2, T1, T2, BUF_NONE, 0, 0, BUF_NONE});
                int res = Xb;
                if (R.ds >= 0) {
                    const int T3 = pick({Xb, T1, T2});
                    out.push_back(Op{OP_CONV, R.ds, Xb, T3, BUF_NONE, 0, 0, BUF_NONE});
                    res = T3;
                }
                out.push_back(Op{OP_CONV, R.c3, T2, T1, res, 0, 0, R.ds >= 0 ? Xb : BUF_NONE});
                Xb = T1;
            }
            h->ops_bt_x = Xb;
        }
    }
    h->feat = cin;
    add_op(OP_AVGPOOL, -1, X, BUF_POOL, BUF_NONE, hcur, cin);
    c = add_conv("fc", "", cin, MPX_NUM_CLASSES, 1, 1, 0, 1, 0, 0);
    h->convs[c].is_fc = true;
    add_op(OP_CONV, c, BUF_POOL, BUF_NONE, BUF_NONE, 0, 0);
    add_op(OP_HEAD, -1, BUF_NONE, BUF_NONE, BUF_NONE, 0, 0);
    if (!h->ops_bt.empty()) {
        h->ops_bt.push_back(Op{OP_AVGPOOL, -1, h->ops_bt_x, BUF_POOL, BUF_NONE, hcur, cin, BUF_NONE});
        h->ops_bt.push_back(h->ops[h->ops.size() - 2]);
        h->ops_bt.push_back(h->ops[h->ops.size() - 1]);
    }
    return 0;
}

// The reference's small networks.  Channels are stored padded to 32 (zero weights / zero scale and
// shift on the padding, so padded channels stay exactly 0 through ReLU and residual adds).
//   MPX_ARCH_MNIST_NET: Classification_Net, generate_gp_training_data_mnist.py:86-105
//       conv1..5 = Conv2d(3x3, pad 1, BIAS) + BN + ReLU (strides 1,1,2,1,2; 1->32->32->64->64->128), conv6 = Conv2d(128,128,3,pad 1)
//       with bias and nothing else, mean over HxW, fc1 128->10.
//   MPX_ARCH_CIFAR_RESNET + depth: ResNetCifar(depth = 6n+2), models/resnet.py:77-146: conv3x3(3->16)+BN+ReLU, three stages
//       of n BasicBlocks (16, 32, 64 planes; the first block of stages 2 and 3 has stride 2 and DownsampleB = AvgPool2d(2) +
//       zero channels on the identity, :64-74), AvgPool2d(8), fc 64->10.
int build_topology_small(mpx_engine* h) {
    h->small = true;
    h->ncls = 10;
    h->logit_pitch = 16;
    auto add_conv = [&](const std::string& name, const std::string& bn, int cin, int cout, int k, int stride, int pad,
                        int hin, int relu, int residual, bool bias, bool fc) {
        ConvLayer L;
        std::memset(&L.d, 0, sizeof L.d);
        set_name(L.d.name, name);
        set_name(L.d.bn_name, bn);
        L.d.cin = cin; L.d.cout = cout; L.d.ksize = k; L.d.stride = stride; L.d.pad = pad;
        L.d.hin = hin; L.d.hout = (hin + 2 * pad - k) / stride + 1;
        L.d.relu = relu; L.d.residual = residual;
        L.has_bias = bias;
        L.is_fc = fc;
        L.cin_pad = (int)round_up(cin, kSmallCPad);
        L.cout_store = fc ? h->logit_pitch : (int)round_up(cout, kSmallCPad);
        L.d.k_packed = k * k * L.cin_pad;
        L.d.cout_pad = (int)round_up(cout, 128);
        L.tile = default_tile(L.d);
        h->convs.push_back(L);
        return (int)h->convs.size() - 1;
    };
    auto add_op = [&](int kind, int conv, int in, int out, int res, int hin, int c) {
        Op o{kind, conv, in, out, res, hin, c, BUF_NONE};
        h->ops.push_back(o);
    };
    int feat = 0, hlast = 0, X = 0;
    size_t act = 0;
    auto track = [&](int hout, int c) { act = std::max(act, (size_t)hout * hout * round_up(c, kSmallCPad)); };
    if (h->arch == MPX_ARCH_MNIST_NET) {
        h->img = 28; h->in_ch = 1;
        const int cfg[5][3] = {{1, 32, 1}, {32, 32, 1}, {32, 64, 2}, {64, 64, 1}, {64, 128, 2}};
        int hcur = 28, in = BUF_INPUT;
        for (int i = 0; i < 5; ++i) {
            const std::string n = "conv" + std::to_string(i + 1);
            const int c = add_conv(n + ".0", n + ".1", cfg[i][0], cfg[i][1], 3, cfg[i][2], 1, hcur, 1, 0, true, false);
            const int out = (in == 0) ? 1 : 0;
            add_op(OP_CONV, c, in, out, BUF_NONE, 0, 0);
            in = out;
            hcur = h->convs[c].d.hout;
            track(hcur, cfg[i][1]);
        }
        const int c6 = add_conv("conv6", "", 128, 128, 3, 1, 1, hcur, 0, 0, true, false);     // bias only: packed like fc (gamma = NULL)
        X = (in == 0) ? 1 : 0;
        add_op(OP_CONV, c6, in, X, BUF_NONE, 0, 0);
        feat = 128; hlast = hcur;
        h->feat = feat;
        add_op(OP_AVGPOOL, -1, X, BUF_POOL, BUF_NONE, hlast, feat);
        const int cf = add_conv("fc1", "", feat, 10, 1, 1, 0, 1, 0, 0, true, true);
        add_op(OP_CONV, cf, BUF_POOL, BUF_NONE, BUF_NONE, 0, 0);
    } else {
        const int depth = h->arch - MPX_ARCH_CIFAR_RESNET;
        if (depth < 8 || (depth - 2) % 6 != 0) return MPX_E_ARG;
        const int n = (depth - 2) / 6;
        h->img = 32; h->in_ch = 3;
        int c = add_conv("conv1", "bn1", 3, 16, 3, 1, 1, 32, 1, 0, false, false);
        add_op(OP_CONV, c, BUF_INPUT, 0, BUF_NONE, 0, 0);
        track(32, 16);
        X = 0;
        int inplanes = 16, hcur = 32;
        const int planes_of[3] = {16, 32, 64};
        for (int sidx = 0; sidx < 3; ++sidx) {
            const int planes = planes_of[sidx];
            for (int b = 0; b < n; ++b) {
                const int stride = (sidx > 0 && b == 0) ? 2 : 1;
                const std::string p = "layer" + std::to_string(sidx + 1) + "." + std::to_string(b) + ".";
                int busy[3] = {X, -1, -1};
                auto pick = [&]() {
                    for (int q = 0; q < kActBufs; ++q)
                        if (q != busy[0] && q != busy[1] && q != busy[2]) return q;
                    return -100;
                };
                const int T1 = pick();
                busy[1] = T1;
                c = add_conv(p + "conv1", p + "bn1", inplanes, planes, 3, stride, 1, hcur, 1, 0, false, false);
                add_op(OP_CONV, c, X, T1, BUF_NONE, 0, 0);
                const int hout = h->convs[c].d.hout;
                int res = X;
                if (stride != 1 || inplanes != planes) {            // DownsampleB on the identity
                    const int T2 = pick();
                    busy[2] = T2;
                    add_op(OP_AVGPAD, -1, X, T2, BUF_NONE, hcur, (int)round_up(inplanes, kSmallCPad) * 65536 + (int)round_up(planes, kSmallCPad));
                    res = T2;
                }
                const int O = pick();
                c = add_conv(p + "conv2", p + "bn2", planes, planes, 3, 1, 1, hout, 1, 1, false, false);
                add_op(OP_CONV, c, T1, O, res, 0, 0);
                X = O;
                inplanes = planes;
                hcur = hout;
                track(hcur, planes);
            }
        }
        feat = 64; hlast = hcur;
        h->feat = feat;
        add_op(OP_AVGPOOL, -1, X, BUF_POOL, BUF_NONE, hlast, feat);
        const int cf = add_conv("fc", "", feat, 10, 1, 1, 0, 1, 0, 0, true, true);
        add_op(OP_CONV, cf, BUF_POOL, BUF_NONE, BUF_NONE, 0, 0);
    }
    add_op(OP_HEAD, -1, BUF_NONE, BUF_NONE, BUF_NONE, 0, 0);
    h->act_elems_per_image = act;
    return 0;
}

uint16_t half_bits(half_t v) {
    uint16_t u;
    std::memcpy(&u, &v, 2);
    return u;
}

hipStream_t as_stream(void* s) { return (hipStream_t)s; }

struct ProfScope {
    mpx_engine* h;
    hipStream_t st;
    ProfRec* rec = nullptr;
    ProfScope(mpx_engine* h_, hipStream_t st_, int kind, int conv) : h(h_), st(st_) {
        if (h->prof_on && h->prof_used < (int)h->prof_pool.size()) {
            rec = &h->prof_pool[h->prof_used++];
            rec->kind = kind;
            rec->conv = conv;
            h->prof_stream = st;
            (void)hipEventRecord(rec->t0, st);
        }
    }
    ~ProfScope() {
        if (rec) (void)hipEventRecord(rec->t1, st);
    }
};

// tile id (mpx_set_conv_tile) of a shape of the generic kernel
template <class Cfg>
constexpr int tile_id_of() {
    return std::is_same<Cfg, ConvTile0>::value ? 0 : std::is_same<Cfg, ConvTile1>::value ? 1 : std::is_same<Cfg, ConvTile2>::value ? 2 :
           std::is_same<Cfg, ConvTile4>::value ? 4 : std::is_same<Cfg, ConvTile7>::value ? 7 : 31;       // 31: a probe build's shape
}

template <class Cfg, bool DUAL = false>
int launch_conv_tile(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = (p.cout + Cfg::TC - 1) / Cfg::TC;     // weights are padded to cout_pad >= n_tiles_c * TC rows
    if (p.n_tiles_c * Cfg::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const int n_tiles_p = (p.M + Cfg::TP - 1) / Cfg::TP;
    const long long nblocks = (long long)n_tiles_p * p.n_tiles_c;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    if (DUAL && (((p.k1 >> 5) - Cfg::NSX) < 0 || (((p.k1 >> 5) - Cfg::NSX) & 1)))
        return fail(h, MPX_E_INTERNAL, "dual conv: k1/32 - ring depth must be even and >= 0");
    hipLaunchKernelGGL((conv_f16x3_kernel<Cfg, DUAL>), dim3((unsigned)nblocks), dim3(Cfg::NT), Cfg::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << tile_id_of<Cfg>();
    return 0;
}

#ifdef MPX_EXPERIMENTAL
// Persistent kernel (mpx_convp.h, tile id 8 = the 128x128 4-wave tile): a fixed grid of
// MINB-per-CU workgroups walks all tiles.
template <class Cfg, bool DUAL = false>
int launch_convp_tile(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = (p.cout + Cfg::TC - 1) / Cfg::TC;
    if (p.n_tiles_c * Cfg::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const int n_tiles_p = (p.M + Cfg::TP - 1) / Cfg::TP;
    const long long n_tiles = (long long)n_tiles_p * p.n_tiles_c;
    if (n_tiles <= 0 || n_tiles > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    if (DUAL && (((p.k1 >> 5) - Cfg::NSX) < 0 || (((p.k1 >> 5) - Cfg::NSX) & 1)))
        return fail(h, MPX_E_INTERNAL, "dual conv: k1/32 - ring depth must be even and >= 0");
    p.n_tiles = (int)n_tiles;
    const int wg_per_cu = (160 * 1024) / Cfg::RING < Cfg::MINB * 256 / Cfg::NT ? (160 * 1024) / Cfg::RING : Cfg::MINB * 256 / Cfg::NT;
    const long long resident = (long long)h->num_cus * (wg_per_cu > 0 ? wg_per_cu : 1);     // a multiple of 8 (256 CUs)
    const unsigned grid = (unsigned)(n_tiles < resident ? n_tiles : resident);
    hipLaunchKernelGGL((convp_f16x3_kernel<Cfg, DUAL>), dim3(grid), dim3(Cfg::NT), Cfg::RING, st, p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

#endif

// 256x256 tile (mpx_conv256.h, tile id 9): 1x1 stride-1 layers whose cout is a multiple of 256 and whose K is a multiple of 64
bool conv256_eligible(const ConvLayer& L) {
    return !L.is_fc && !L.is_stem && L.d.ksize == 1 && L.d.stride == 1 && L.d.pad == 0 && L.cout_store % 256 == 0 &&
           L.cin_pad % 64 == 0 && L.cin_pad == L.d.cin;
}

// persistent expanding-1x1 kernel (mpx_convx.h, tile id 10): as tile 9 plus K >= 128; a launch with fewer tiles than one grid
// unit (8 or n_tiles_c workgroups) runs on the 128x128 kernel instead
bool convx_eligible(const ConvLayer& L) { return conv256_eligible(L) && L.cin_pad >= 128; }

int launch_convx(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = p.cout / ConvX::TC;
    if (p.n_tiles_c * ConvX::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const long long n_tiles_p = ((long long)p.M + ConvX::TP - 1) / ConvX::TP;
    const long long total = n_tiles_p * p.n_tiles_c;
    if (total <= 0 || total > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    // one workgroup per CU; the grid must be a multiple of 8 (XCD classes) and of n_tiles_c (a workgroup keeps its cout tile)
    long long grid = h->num_cus;
    if (total < grid) grid = total;
    const int unit = 8 > p.n_tiles_c ? (8 % p.n_tiles_c == 0 ? 8 : 8 * p.n_tiles_c) : (p.n_tiles_c % 8 == 0 ? p.n_tiles_c : 8 * p.n_tiles_c);
    grid = grid / unit * unit;
    // less than one round of tiles (small batches: one image's window table, a BO round): the persistent walk would serialise the
    // tiles that do not fill a grid unit, and half-empty CUs gain more from tiles a quarter of the size -- the 128x128 kernel, which
    // sums in the same order (bit-identical).  tools/layer_profile.py, conv ms per forward: batch 29 8.24 -> 3.92, 118 10.28 -> 6.78,
    // 211 13.50 -> 11.62 with this rule on the three persistent kernels
    if (grid <= 0 || total < h->num_cus) return launch_conv_tile<ConvTile7>(h, p, cout_pad, st);
    hipLaunchKernelGGL(convx_f16x3_kernel, dim3((unsigned)grid), dim3(ConvX::NT), ConvX::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 10;
    return 0;
}

// weights-in-registers expanding-1x1 kernel (mpx_convw.h, tile id 14): tile 10's layers with K = 256 exactly (a wave keeps the
// 64 x 256 x (hi + lo) weights of its channels in 256 VGPRs); under one round of tiles the 128x128 kernel, as launch_convx
bool convw_eligible(const ConvLayer& L) { return conv256_eligible(L) && L.cin_pad == ConvW::K; }

int launch_convw(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    if (p.ktot != ConvW::K) return fail(h, MPX_E_INTERNAL, "convw: K = %d (the kernel holds K = %d in registers)", p.ktot, ConvW::K);
    // The kernel serves what every layer it is the default for asks of it and what the parity tests cover: the residual add + ReLU of a block's
    // last conv.  A call through mpx_conv_bn_act WITHOUT a residual or (no such layer exists in the networks) without ReLU takes tile 10's
    // kernel, which sums in the same order: only <K, RELU = true> is instantiated (the <false> and null-residual paths had no test).
    if (!p.relu || !p.r_hi) return launch_convx(h, p, cout_pad, st);
    p.n_tiles_c = p.cout / ConvW::TC;
    if (p.n_tiles_c * ConvW::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const long long n_tiles_p = ((long long)p.M + ConvW::TP - 1) / ConvW::TP;
    const long long total = n_tiles_p * p.n_tiles_c;
    if (total <= 0 || total > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    long long grid = h->num_cus;
    if (total < grid) grid = total;
    const int unit = 8 > p.n_tiles_c ? (8 % p.n_tiles_c == 0 ? 8 : 8 * p.n_tiles_c) : (p.n_tiles_c % 8 == 0 ? p.n_tiles_c : 8 * p.n_tiles_c);
    grid = grid / unit * unit;
    if (grid <= 0 || total < 2LL * h->num_cus) return launch_conv_tile<ConvTile7>(h, p, cout_pad, st);
    // (the kernel is a template over K; K = 128 -- 128 -> 512 on 28 x 28 maps -- was built and measured in round 5: parity green, bit-equal to tile 10, and a
    // tie with it, 7.26 against 7.19 ms for the four layers: two stages of the epilogue slice per MFMA gap do not hide any more.  Only K = 256 is instantiated.)
    hipLaunchKernelGGL((convw_f16x3_kernel<ConvW::K, true>), dim3((unsigned)grid), dim3(ConvW::NT), ConvW::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 14;
    return 0;
}

#ifdef MPX_EXPERIMENTAL
// X-stationary expanding-1x1 kernel (mpx_convs.h, tile id 11): 1x1 stride-1 layers with cout % 256 == 0 and K = 128 or 256
bool convs_eligible(const ConvLayer& L) { return conv256_eligible(L) && (L.cin_pad == 128 || L.cin_pad == 256); }

int launch_convs(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = p.cout / ConvS::TC;
    if (p.n_tiles_c * ConvS::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const long long n_pt = ((long long)p.M + ConvS::TP - 1) / ConvS::TP;
    if (n_pt <= 0 || n_pt > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    const long long grid = n_pt < h->num_cus ? n_pt : h->num_cus;         // one persistent workgroup per CU
    hipLaunchKernelGGL(convs_f16x3_kernel, dim3((unsigned)grid), dim3(ConvS::NT), ConvS::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

#endif

int launch_conv256(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = p.cout / Conv256::TC;
    if (p.n_tiles_c * Conv256::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const int n_tiles_p = (p.M + Conv256::TP - 1) / Conv256::TP;
    const long long nblocks = (long long)n_tiles_p * p.n_tiles_c;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    hipLaunchKernelGGL(conv256_f16x3_kernel, dim3((unsigned)nblocks), dim3(Conv256::NT), Conv256::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 9;
    return 0;
}

// persistent 256x256 kernel (mpx_conv256p.h, tile id 13): tile 9's layers without a residual operand; a launch with fewer tiles than
// one grid unit (8 or n_tiles_c workgroups) runs on the 256x256 kernel itself (same bits)
bool conv256p_eligible(const ConvLayer& L) { return conv256_eligible(L) && !L.d.residual; }

int launch_conv256p(mpx_engine* h, ConvParams& p, int cout_pad, hipStream_t st) {
    p.n_tiles_c = p.cout / Conv256P::TC;
    if (p.n_tiles_c * Conv256P::TC > cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    const long long n_tiles_p = ((long long)p.M + Conv256P::TP - 1) / Conv256P::TP;
    const long long total = n_tiles_p * p.n_tiles_c;
    if (total <= 0 || total > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    long long grid = h->num_cus;
    if (total < grid) grid = total;
    const int unit = 8 > p.n_tiles_c ? (8 % p.n_tiles_c == 0 ? 8 : 8 * p.n_tiles_c) : (p.n_tiles_c % 8 == 0 ? p.n_tiles_c : 8 * p.n_tiles_c);
    grid = grid / unit * unit;
    if (p.r_hi) return launch_conv256(h, p, cout_pad, st);
    if (grid <= 0 || total < h->num_cus) return launch_conv_tile<ConvTile2>(h, p, cout_pad, st);      // under one round: smaller tiles (launch_convx)
    hipLaunchKernelGGL(conv256p_f16x3_kernel, dim3((unsigned)grid), dim3(Conv256P::NT), Conv256P::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 13;
    return 0;
}

// Rows of the largest input patch any TP-pixel tile of an HxH map needs (mpx_conv3p.h), rounded up to 16.
int patch_rows_needed(int H, int TP) {
    const int W = H, PW = W + 2, PIMG = (H + 2) * PW, howo = H * W;
    auto pb = [&](long long m) { const long long n = m / howo, rem = m % howo; return n * PIMG + (rem / W) * PW + rem % W; };
    long long worst = 0;
    for (long long m0 = 0; m0 < (long long)TP * howo; m0 += TP) {      // the pattern repeats after lcm(TP, H*W) pixels
        const long long r = pb(m0 + TP - 1) - pb(m0) + 2 * PW + 3;
        worst = r > worst ? r : worst;
    }
    return (int)((worst + 15) / 16 * 16);
}

constexpr int kLdsLimit = 160 * 1024;

// 3x3 stride-1 layers whose patch fits the LDS: the patch kernel (tile id 6)
template <class PC>
bool patch_fits(const mpx_conv_desc& d) {
    if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.cin % 64 != 0 || d.hin != d.hout) return false;
    const int rows = patch_rows_needed(d.hin, PC::TP);
    return rows <= PC::MAX_PATCH_ROWS && PC::lds_bytes(rows) <= kLdsLimit;
}

bool patch_eligible(const mpx_conv_desc& d) {
    return d.cout <= 64 ? patch_fits<PatchTile1>(d) : (patch_fits<PatchTile0>(d) || patch_fits<PatchTile2>(d));
}

template <class PC>
int launch_conv_patch(mpx_engine* h, ConvParams& p, const mpx_conv_desc& d, hipStream_t st) {
    p.n_tiles_c = (p.cout + PC::TC - 1) / PC::TC;
    if (p.n_tiles_c * PC::TC > d.cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    p.patch_rows = patch_rows_needed(d.hin, PC::TP);
    const int lds = PC::lds_bytes(p.patch_rows);
    const int n_tiles_p = (p.M + PC::TP - 1) / PC::TP;
    const long long nblocks = (long long)n_tiles_p * p.n_tiles_c;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    hipLaunchKernelGGL(conv3x3p_f16x3_kernel<PC>, dim3((unsigned)nblocks), dim3(PC::NT), lds, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 6;
    return 0;
}

// persistent patch kernel (mpx_conv3pp.h, tile id 12): the patch kernel's layers with cout >= 128 and no residual whose LDS image leaves 1 KB for the
// cout tile's scale / shift vectors; a launch with fewer tiles than one grid unit runs on the patch kernel itself (same bits)
template <class PC>
bool patchp_fits(const mpx_conv_desc& d) {
    return d.cout >= 128 && !d.residual && patch_fits<PC>(d) && PatchPersistent<PC>::lds_bytes(patch_rows_needed(d.hin, PC::TP)) <= kLdsLimit;
}

bool patchp_eligible(const mpx_conv_desc& d) { return patchp_fits<PatchTile0>(d) || patchp_fits<PatchTile2>(d); }

template <class PC>
int launch_conv_patchp(mpx_engine* h, ConvParams& p, const mpx_conv_desc& d, hipStream_t st) {
    p.n_tiles_c = (p.cout + PC::TC - 1) / PC::TC;
    if (p.n_tiles_c * PC::TC > d.cout_pad) return fail(h, MPX_E_ARG, "conv tile exceeds the packed weight rows");
    p.patch_rows = patch_rows_needed(d.hin, PC::TP);
    const int lds = PatchPersistent<PC>::lds_bytes(p.patch_rows);
    const long long n_tiles_p = ((long long)p.M + PC::TP - 1) / PC::TP;
    const long long total = n_tiles_p * p.n_tiles_c;
    if (total <= 0 || total > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    // one workgroup per CU; the grid is a multiple of 8 (XCD classes) and of n_tiles_c (a workgroup keeps its cout tile)
    long long grid = h->num_cus;
    if (total < grid) grid = total;
    const int unit = 8 > p.n_tiles_c ? (8 % p.n_tiles_c == 0 ? 8 : 8 * p.n_tiles_c) : (p.n_tiles_c % 8 == 0 ? p.n_tiles_c : 8 * p.n_tiles_c);
    grid = grid / unit * unit;
    // (the kernel divides pixel and padded-pixel indices with a float reciprocal: exact below 2^23)
    const long long padded = (long long)(p.M / (d.hin * d.hin) + 2) * (d.hin + 2) * (d.hin + 2);
    if (grid <= 0 || total < h->num_cus || p.r_hi || padded >= (1 << 23)) return launch_conv_patch<PC>(h, p, d, st);     // under one round: a workgroup per tile
    hipLaunchKernelGGL(conv3x3pp_f16x3_kernel<PC>, dim3((unsigned)grid), dim3(PC::NT), lds, st, p);
    MPX_HIP(h, hipGetLastError());
    h->last_kernels |= 1u << 12;
    return 0;
}

// Default variant per layer, from tools/tile_sweep.sh and in-network tools/layer_profile.py runs on MI355X at
// batch 2048: cout <= 64 layers take a 64-row tile (no zero-padded MFMA rows); of the rest, the 128x128 tile
// with two workgroups per CU wins on every 1x1 shape by 5-15 % (one workgroup's HBM-bound epilogue overlaps the
// other's K loop) and wide 3x3 layers keep the 128x256 tile (fewest L2 bytes per FLOP; 7 % faster in the
// network although the isolated layer bench prefers 128x128); the stride-1 ones whose input patch fits the LDS run
// the patch kernel (mpx_conv3p.h, tile id 6).  Every tile stays selectable.
int default_tile(const mpx_conv_desc& d) {
    if (d.cout <= 64) return d.ksize >= 3 ? 1 : 4;       // (the patch kernel is 10 % slower than tile 1 on 64->64 in the network)
    // ... as one persistent workgroup per CU (mpx_conv3pp.h) where the layer has no residual operand: in the network (batch 2340,
    // A/B in one call) 128->128 on 28x28 3.89 -> 3.67 ms, 256->256 on 14x14 24.65 -> 24.49 ms per batch; 7x7 maps lose 1 % (2.29 -> 2.32)
#ifndef MPX_PROBE_NO_PERSISTENT_PATCH   // A/B builds only (tools/ab_lib.sh)
    if (patchp_eligible(d) && d.hout >= 14) return 12;
#endif
    if (patch_eligible(d)) return 6;                      // 3x3 stride 1 on 28x28 / 14x14 / 7x7 maps: -10..18 % against tile 0
    if (d.ksize == 3) return 0;
    // expanding 1x1 layers (short K, long epilogue): four waves per SIMD cover the epilogue better, -2..4 % in the
    // network; the reducing ones (long K) gain nothing from it
    // ... and on 28x28 / 14x14 maps with K >= 128 the persistent pipelined kernel (mpx_convx.h) is 2-4 % faster still in the
    // network (256->1024: 21.5 -> 21.0 ms, 128->512: 6.5 -> 6.2); on 7x7 maps and K = 64 it ties tile 7
    // ... and with K = 256 exactly (256 -> 1024, a fifth of ResNet-101's conv time) the kernel that keeps the weights of a wave in its
    // 256 AGPRs and only pixels in the LDS (mpx_convw.h): in the network at batch 2340, A/B in one call, 24.3 -> 21.3 ms for the 23 layers
#ifndef MPX_PROBE_NO_CONVW               // A/B builds only (tools/ab_lib.sh)
    if (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.cout > d.cin && d.cout % 256 == 0 && d.cin == ConvW::K && d.hout >= 14) return 14;
#endif
    // (round 4: on 7x7 maps too -- 512 -> 2048 x3 3.53 -> 3.38 ms per batch in the network, A/B in one call; round 2 had measured a tie)
    if (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.cout > d.cin && d.cout % 256 == 0 && d.cin % 64 == 0 && d.cin >= 128) return 10;
    if (d.stride == 1 && d.cout > d.cin) return 7;
    // reducing / square 1x1 stride-1 layers with cout % 256 == 0: the 256x256 tile (mpx_conv256.h) halves the operand bytes
    // per MAC and runs +10..18 % (1024->256: 365 vs 322 TFLOP/s, 1024->512: 411 vs 346); expanding layers lose on it (one
    // workgroup per CU: nothing covers the long epilogue)
    // ... as one persistent workgroup per CU with a register epilogue (mpx_conv256p.h) where the layer has no residual operand (all of
    // them in the torchvision ResNets): in the network at batch 2340, A/B in one call, 1024->256 13.73 -> 13.12 ms, 512->256 1.47 -> 1.32,
    // 1024->512 1.18 -> 1.11, 2048->512 1.11 -> 1.10 per batch
#ifdef MPX_PROBE_NO_PERSISTENT_256      // A/B builds only (tools/ab_lib.sh)
    if (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.cout % 256 == 0 && d.cin % 64 == 0) return 9;
#endif
    if (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.cout % 256 == 0 && d.cin % 64 == 0) return d.residual ? 9 : 13;
    return 2;
}

// one launch of layer L's conv with the given kernel variant
int dispatch_conv(mpx_engine* h, const ConvLayer& L, ConvParams& p, int tile, hipStream_t st) {
    if (tile == 12)
        return patchp_fits<PatchTile0>(L.d) ? launch_conv_patchp<PatchTile0>(h, p, L.d, st) : launch_conv_patchp<PatchTile2>(h, p, L.d, st);
    if (tile == 6) {
        if (L.d.cout <= 64) return launch_conv_patch<PatchTile1>(h, p, L.d, st);
        return patch_fits<PatchTile0>(L.d) ? launch_conv_patch<PatchTile0>(h, p, L.d, st) : launch_conv_patch<PatchTile2>(h, p, L.d, st);
    }
    switch (tile) {
        case 10: return launch_convx(h, p, L.d.cout_pad, st);
        case 14: return launch_convw(h, p, L.d.cout_pad, st);
        case 9: return launch_conv256(h, p, L.d.cout_pad, st);
        case 13: return launch_conv256p(h, p, L.d.cout_pad, st);
        case 0: return launch_conv_tile<ConvTile0>(h, p, L.d.cout_pad, st);
        case 1: return launch_conv_tile<ConvTile1>(h, p, L.d.cout_pad, st);
        case 2: return launch_conv_tile<ConvTile2>(h, p, L.d.cout_pad, st);
        case 7: return launch_conv_tile<ConvTile7>(h, p, L.d.cout_pad, st);
        case 4: return launch_conv_tile<ConvTile4>(h, p, L.d.cout_pad, st);
#ifdef MPX_EXPERIMENTAL
        case 11: return launch_convs(h, p, L.d.cout_pad, st);
        case 8: return launch_convp_tile<ConvTile2>(h, p, L.d.cout_pad, st);
        case 3: return launch_conv_tile<ConvTile3>(h, p, L.d.cout_pad, st);
        case 5: return launch_conv_tile<ConvTile5>(h, p, L.d.cout_pad, st);
#endif
        default: return fail(h, MPX_E_INTERNAL, "conv: no kernel for tile id %d", tile);
    }
}

int launch_conv(mpx_engine* h, int i, const half_t* in_hi, const half_t* in_lo, const half_t* r_hi,
                const half_t* r_lo, half_t* y_hi, half_t* y_lo, float* y_f32, int B, hipStream_t st) {
    const ConvLayer& L = h->convs[i];
    if (!L.loaded) return fail(h, MPX_E_STATE, "layer %d (%s) has no weights", i, L.d.name);
    ConvParams p;
    std::memset(&p, 0, sizeof p);
    p.w_hi = L.w_hi; p.w_lo = L.w_lo; p.scale = L.scale; p.shift = L.shift;
    p.r_hi = r_hi; p.r_lo = r_lo; p.y_hi = y_hi; p.y_lo = y_lo; p.y_f32 = y_f32;
    p.cout = L.cout_store;
    p.relu = L.d.relu;
    p.ktot = L.d.k_packed;
    if (L.is_stem) {
        // padded NHWC4 staging: a (ky) tap is one 64-B run of 8 pixels x 4 channels, no bounds to check
        p.x_hi = h->in_hi; p.x_lo = h->in_lo;
        p.hin = MPX_IMG_PAD; p.win = MPX_IMG_PAD; p.pix_stride = 4;
        p.kh = kStemK; p.kw = 1; p.stride = 2; p.pad = 0; p.k_per_tap = 32;
    } else {
        p.x_hi = in_hi; p.x_lo = in_lo;
        p.hin = L.d.hin; p.win = L.d.hin; p.pix_stride = L.cin_pad;
        p.kh = L.d.ksize; p.kw = L.d.ksize; p.stride = L.d.stride; p.pad = L.d.pad; p.k_per_tap = L.cin_pad;
    }
    p.ho = L.d.hout; p.wo = L.d.hout;
    const long long M = (long long)B * p.ho * p.wo;
    if (M > 0x7fffffffLL || (long long)B * p.hin * p.win > 0x7fffffffLL)
        return fail(h, MPX_E_ARG, "batch too large for 32-bit pixel indices");
    p.M = (int)M;
#ifdef MPX_DIAG
    p.stamps = h->stamps;
#endif
    ProfScope ps(h, st, OP_CONV, i);
    h->last_kernels = 0;
    // The 256x256 kernel keeps ONE workgroup per CU, so a launch runs in "rounds" of num_cus tiles and a small remainder would
    // keep most CUs idle for a whole tile time (14x14 maps at batch 2048: 1568 tiles = 6.125 rounds).  Then the images of
    // the whole rounds go to that kernel and the last few images (images are independent: an image range is a pointer
    // offset) to the 128x128 kernel, whose tiles are a quarter of the size and which sums in the same order (bit-identical
    // scores wherever a mask sits in the batch): 1024->256 12.9 -> 12.3 ms, 2048->512 1.15 -> 1.01 ms per batch.  (The same
    // split of the 3x3 patch kernel's and tile 0's last round was measured and gains nothing: their remainder lives as long
    // as one of their tiles whatever its tile size)
    int big_tc = 0, big_tp = 0;
    if (L.tile == 9 || L.tile == 13) { big_tc = Conv256::TC; big_tp = Conv256::TP; }
    if (big_tc && !L.is_stem) {
        const long long tiles_c = (p.cout + big_tc - 1) / big_tc, tiles_p = (M + big_tp - 1) / big_tp;
        const long long total = tiles_p * tiles_c, rounds = total / h->num_cus, rest = total - rounds * h->num_cus;
        const long long howo = (long long)p.ho * p.wo;
        const long long n_a = rounds * h->num_cus / tiles_c * big_tp / howo;           // images that fill the whole rounds
        if (rounds >= 1 && rest > 0 && 2 * rest <= h->num_cus && n_a >= 1 && n_a < B) {
            ConvParams q = p;
            p.M = (int)(n_a * howo);
            int rc = dispatch_conv(h, L, p, L.tile, st);
            if (rc) return rc;
            const size_t xo = (size_t)n_a * p.hin * p.win * p.pix_stride, yo = (size_t)n_a * howo * p.cout;
            q.M = (int)(M - n_a * howo);
            q.x_hi += xo; q.x_lo += xo;
            q.y_hi += yo; q.y_lo += yo;
            if (q.r_hi) { q.r_hi += yo; q.r_lo += yo; }
            return dispatch_conv(h, L, q, 2, st);
        }
    }
    return dispatch_conv(h, L, p, L.tile, st);
}

// A block's last 1x1 conv with its downsample branch K-concatenated: out = relu(s * (W3' . t2 + Wds' . x) + shift)
// (build_fused).  in = t2 planes [B][ho][wo][k1], x2 = block input planes [B][hin2][hin2][k2].
int launch_conv_fused(mpx_engine* h, int i, const half_t* in_hi, const half_t* in_lo, const half_t* x2_hi,
                      const half_t* x2_lo, half_t* y_hi, half_t* y_lo, int B, hipStream_t st) {
    const ConvLayer& L = h->convs[i];
    if (!L.fuse_main || !L.fused_loaded) return fail(h, MPX_E_STATE, "layer %d (%s) has no fused weights", i, L.d.name);
    const ConvLayer& D = h->convs[L.fuse_partner];
    ConvParams p;
    std::memset(&p, 0, sizeof p);
    p.w_hi = L.fw_hi; p.w_lo = L.fw_lo; p.scale = L.fscale; p.shift = L.fshift;
    p.y_hi = y_hi; p.y_lo = y_lo;
    p.cout = L.d.cout; p.relu = 1;
    p.k1 = L.d.cin; p.ktot = L.d.cin + D.d.cin;
    p.x_hi = in_hi; p.x_lo = in_lo;
    p.hin = L.d.hin; p.win = L.d.hin; p.pix_stride = L.d.cin;
    p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.k_per_tap = L.d.cin;
    p.x2_hi = x2_hi; p.x2_lo = x2_lo;
    p.hin2 = D.d.hin; p.win2 = D.d.hin; p.pix_stride2 = D.d.cin; p.stride2 = D.d.stride;
    p.ho = L.d.hout; p.wo = L.d.hout;
    const long long M = (long long)B * p.ho * p.wo;
    if (M > 0x7fffffffLL || (long long)B * p.hin2 * p.win2 > 0x7fffffffLL)
        return fail(h, MPX_E_ARG, "batch too large for 32-bit pixel indices");
    p.M = (int)M;
#ifdef MPX_DIAG
    p.stamps = h->stamps;
#endif
    ProfScope ps(h, st, OP_CONV, i);
    if (L.tile == 2) return launch_conv_tile<ConvTile2, true>(h, p, L.d.cout_pad, st);
#ifdef MPX_EXPERIMENTAL
    if (L.tile == 8) return launch_convp_tile<ConvTile2, true>(h, p, L.d.cout_pad, st);
#endif
    return launch_conv_tile<ConvTile7, true>(h, p, L.d.cout_pad, st);       // (tiles 7, 10 and 14: the persistent kernels have no dual-operand form)
}

// The ImageNet stem (7x7 stride-2 conv + BN + ReLU) with its 3x3 stride-2 pad-1 max pool in ONE launch (mpx_conv.h, POOL): writes the
// pooled planes [B][56][56][64]; the 112x112 conv output is never stored.  Needs the 64-row tile 1 (the stem's default).
bool stem_pool_eligible(const mpx_engine* h) {
    if (h->small || h->convs.empty()) return false;
    const ConvLayer& L = h->convs[0];
    return L.is_stem && L.d.cout == 64 && L.d.hout == 112 && (L.d.hout / 2) % POOL_PY == 0 && (L.d.hout / 2) % POOL_PX == 0 && L.d.relu;
}

int launch_stem_pool(mpx_engine* h, half_t* y_hi, half_t* y_lo, int B, hipStream_t st) {
    const ConvLayer& L = h->convs[0];
    if (!stem_pool_eligible(h)) return fail(h, MPX_E_STATE, "stem + max pool: this architecture has no 7x7 stem with a max pool");
    if (!L.loaded) return fail(h, MPX_E_STATE, "layer 0 (%s) has no weights", L.d.name);
    ConvParams p;
    std::memset(&p, 0, sizeof p);
    p.w_hi = L.w_hi; p.w_lo = L.w_lo; p.scale = L.scale; p.shift = L.shift;
    p.y_hi = y_hi; p.y_lo = y_lo;
    p.cout = L.cout_store; p.relu = L.d.relu; p.ktot = L.d.k_packed;
    p.x_hi = h->in_hi; p.x_lo = h->in_lo;
    p.hin = MPX_IMG_PAD; p.win = MPX_IMG_PAD; p.pix_stride = 4;
    p.kh = kStemK; p.kw = 1; p.stride = 2; p.pad = 0; p.k_per_tap = 32;
    p.ho = L.d.hout; p.wo = L.d.hout;
    const long long M = (long long)B * p.ho * p.wo;
    if (M > 0x7fffffffLL) return fail(h, MPX_E_ARG, "batch too large for 32-bit pixel indices");
    p.M = (int)M;
    p.n_tiles_c = 1;
#ifdef MPX_DIAG
    p.stamps = h->stamps;
#endif
    const long long nblocks = (long long)B * (p.ho / 2 / POOL_PY) * (p.wo / 2 / POOL_PX);
    if (nblocks > 0x7fffffffLL) return fail(h, MPX_E_ARG, "conv grid out of range");
    ProfScope ps(h, st, OP_CONV, 0);
    hipLaunchKernelGGL((conv_f16x3_kernel<ConvTile1, false, true>), dim3((unsigned)nblocks), dim3(ConvTile1::NT), ConvTile1::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

// K permutation of a block tail's conv3 operand (mpx_btail.h): K position 8g + j of a 32-wide step holds channel (j>>2)*16 + 4g + (j&3),
// because that is where the conv2 accumulators of a lane group g sit (D layout of v_mfma_f32_16x16x32_f16: registers 4g .. 4g+3 of
// two stacked 16-row fragments).
constexpr int bt_perm_channel(int kpos) {
    const int s = kpos >> 5, q = kpos & 31, g = q >> 3, j = q & 7;
    return 32 * s + (j >> 2) * 16 + 4 * g + (j & 3);
}
constexpr bool bt_perm_is_bijection() {        // every channel of [0,64) exactly once, each 32-wide K step onto its own 32 channels
    bool seen[BT_MID] = {};
    for (int k = 0; k < BT_MID; ++k) {
        const int c = bt_perm_channel(k);
        if (c < 0 || c >= BT_MID || seen[c] || (c >> 5) != (k >> 5)) return false;
        seen[c] = true;
    }
    return true;
}
static_assert(bt_perm_is_bijection(), "K permutation of the block tail's conv3 operand");
static_assert(bt_perm_channel(0) == 0 && bt_perm_channel(3) == 3 && bt_perm_channel(4) == 16 && bt_perm_channel(8) == 4 && bt_perm_channel(36) == 48,
              "K position 8g + j <-> channel (j>>2)*16 + 4g + (j&3): registers 4g..4g+3 of two stacked 16-row accumulator fragments");

// Upload the K-permuted copy of packed planes `hi`/`lo` ([rows >= 256][K], piece-major, host) into the tail's own planes:
// columns [0,64) permuted, columns [64,K) (the downsample branch of a DUAL tail) as they are.
int upload_tail_planes(mpx_engine* h, TailBlock& tb, const std::vector<uint16_t>& hi, const std::vector<uint16_t>& lo, int K) {
    std::vector<uint16_t> ph((size_t)BT_OUT * K), pl((size_t)BT_OUT * K);
    for (int row = 0; row < BT_OUT; ++row)
        for (int k = 0; k < K; ++k) {
            const int src = k < BT_MID ? bt_perm_channel(k) : k;
            ph[w_packed_index(row, k, K)] = hi[w_packed_index(row, src, K)];
            pl[w_packed_index(row, k, K)] = lo[w_packed_index(row, src, K)];
        }
    MPX_HIP(h, hipMemcpy(tb.w3p_hi, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(tb.w3p_lo, pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
    tb.ready = true;
    return 0;
}

TailBlock* tail_of_conv3(mpx_engine* h, int c3) {
    for (TailBlock& tb : h->tails)
        if (tb.c3 == c3) return &tb;
    return nullptr;
}

bool tails_ready(const mpx_engine* h) {
    if (h->tails.empty()) return false;
    for (const TailBlock& tb : h->tails)
        if (!tb.ready || !h->convs[tb.c2].loaded || !h->convs[tb.next1].loaded || !h->convs[tb.c1].loaded) return false;
    return true;
}

// One block tail (mpx_btail.h): t1 planes [B][56][56][64], x = identity planes [B][56][56][256] (first block of layer1: the block
// input [B][56][56][64]), y = block output [B][56][56][256], z = the next block's conv1 output [B][56][56][64 | 128].
int launch_btail(mpx_engine* h, int ti, const half_t* t_hi, const half_t* t_lo, const half_t* x_hi, const half_t* x_lo,
                 half_t* y_hi, half_t* y_lo, half_t* z_hi, half_t* z_lo, int B, hipStream_t st) {
    const TailBlock& tb = h->tails[ti];
    const ConvLayer& L2 = h->convs[tb.c2];
    const ConvLayer& L3 = h->convs[tb.c3];
    const ConvLayer& N1 = h->convs[tb.next1];
    if (!tb.ready || !L2.loaded || !N1.loaded) return fail(h, MPX_E_STATE, "block tail %s: weights missing", L2.d.name);
    const bool dual = tb.ds >= 0;
    const bool head = t_hi == nullptr;          // the block's own conv1 runs inside the launch, on the patch of the block input
    if (head && !dual) return fail(h, MPX_E_ARG, "block tail %s: only a block with a downsample branch (64-channel input) can run its own conv1 in the launch", L2.d.name);
    BtParams p;
    std::memset(&p, 0, sizeof p);
    p.t_hi = head ? x_hi : t_hi; p.t_lo = head ? x_lo : t_lo;
    if (head) {
        const ConvLayer& L1 = h->convs[tb.c1];
        if (!L1.loaded) return fail(h, MPX_E_STATE, "block tail %s: weights of %s missing", L2.d.name, L1.d.name);
        if (!(L1.d.cin == BT_MID && L1.d.cout == BT_MID && L1.d.ksize == 1 && L1.d.stride == 1)) return fail(h, MPX_E_INTERNAL, "block tail: %s is not a 64 -> 64 1x1 conv", L1.d.name);
        p.w0_hi = L1.w_hi; p.w0_lo = L1.w_lo; p.sc0 = L1.scale; p.sh0 = L1.shift;
    }
    p.w2_hi = L2.w_hi; p.w2_lo = L2.w_lo; p.sc2 = L2.scale; p.sh2 = L2.shift;
    p.w3_hi = tb.w3p_hi; p.w3_lo = tb.w3p_lo;
    p.sc3 = dual ? L3.fscale : L3.scale; p.sh3 = dual ? L3.fshift : L3.shift;
    p.r_hi = x_hi; p.r_lo = x_lo; p.y_hi = y_hi; p.y_lo = y_lo;
    p.w1_hi = N1.w_hi; p.w1_lo = N1.w_lo; p.sc1 = N1.scale; p.sh1 = N1.shift;
    p.z_hi = z_hi; p.z_lo = z_lo;
    p.B = B; p.H = L2.d.hin; p.W = L2.d.hin;
    if (p.H % BT_TY || p.W % BT_TX) return fail(h, MPX_E_INTERNAL, "block tail: %dx%d map is not a whole number of 8x14 tiles", p.H, p.W);
    p.tiles_x = p.W / BT_TX;
    p.tiles_per_img = (p.H / BT_TY) * p.tiles_x;
    const long long n_tiles = (long long)B * p.tiles_per_img;
    if (n_tiles <= 0 || n_tiles > 0x7fffffffLL) return fail(h, MPX_E_ARG, "block tail: batch out of range");
    p.n_tiles = (int)n_tiles;
#ifdef MPX_DIAG
    p.stamps = h->stamps;
#endif
    const long long resident = 2LL * h->num_cus;                       // two 80-KB workgroups per CU
    const unsigned grid = (unsigned)std::min<long long>(resident / 8 * 8, (n_tiles + 7) / 8 * 8);
    ProfScope ps(h, st, OP_CONV, tb.c2);
    if (head) hipLaunchKernelGGL(btail_f16x3_kernel<BtHeadC64>, dim3(grid), dim3(256), BtHeadC64::LDS, st, p);
    else if (dual) hipLaunchKernelGGL(btail_f16x3_kernel<BtDualC64>, dim3(grid), dim3(256), BtDualC64::LDS, st, p);
    else if (N1.d.cout == 128) hipLaunchKernelGGL(btail_f16x3_kernel<BtResC128>, dim3(grid), dim3(256), BtResC128::LDS, st, p);
    else hipLaunchKernelGGL(btail_f16x3_kernel<BtResC64>, dim3(grid), dim3(256), BtResC64::LDS, st, p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

// Fused planes of a (main, ds) pair from the host copies of both layers.  With s3 = g3/sqrt(v3+eps) and sd likewise,
//   bn3(W3.t2) + bnd(Wd.x) = s * ((W3 * s3/s) . t2 + (Wd * sd/s) . x) + shift3 + shiftd,   s = max(|s3|, |sd|) per channel
// (both ratios are <= 1 in magnitude, so a vanishing gamma on either branch is harmless).  The scaled rows are rounded
// to fp32 once and then split into hi + lo like any other weight (22 bits); per-row power-of-two normalisation as
// mpx_pack_conv_weights.
int build_fused(mpx_engine* h, int main) {
    ConvLayer& L = h->convs[main];
    ConvLayer& D = h->convs[L.fuse_partner];
    const int cout = L.d.cout, k1 = L.d.cin, k2 = D.d.cin, K = k1 + k2, rows = L.d.cout_pad;
    std::vector<uint16_t> hi((size_t)rows * K, 0), lo((size_t)rows * K, 0);
    std::vector<float> sc(rows, 0.f), sh(rows, 0.f), row(K);
    for (int co = 0; co < cout; ++co) {
        const double s3 = (double)L.hgamma[co] / std::sqrt((double)L.hvar[co] + (double)L.heps);
        const double sd = (double)D.hgamma[co] / std::sqrt((double)D.hvar[co] + (double)D.heps);
        double s = std::fmax(std::fabs(s3), std::fabs(sd));
        if (!(s > 0.0) || !std::isfinite(s)) s = 1.0;
        const double r3 = s3 / s, rd = sd / s;
        float mx = 0.f;
        for (int k = 0; k < k1; ++k) { row[k] = (float)((double)L.hw[(size_t)co * k1 + k] * r3); mx = std::fmax(mx, std::fabs(row[k])); }
        for (int k = 0; k < k2; ++k) { row[k1 + k] = (float)((double)D.hw[(size_t)co * k2 + k] * rd); mx = std::fmax(mx, std::fabs(row[k1 + k])); }
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) {
            int ex;
            std::frexp(mx, &ex);
            e = 10 - ex;
        }
        for (int k = 0; k < K; ++k) {
            const float sv = std::ldexp(row[k], e);
            const half_t vh = (half_t)sv;
            const half_t vl = (half_t)(sv - (float)vh);
            hi[w_packed_index(co, k, K)] = half_bits(vh);
            lo[w_packed_index(co, k, K)] = half_bits(vl);
        }
        sc[co] = (float)std::ldexp(s, -e);
        sh[co] = (float)(((double)L.hbeta[co] - (double)L.hmean[co] * s3) + ((double)D.hbeta[co] - (double)D.hmean[co] * sd));
    }
    MPX_HIP(h, hipMemcpy(L.fw_hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.fw_lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.fscale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.fshift, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    L.fused_loaded = true;
    // (the host copies of both layers stay: they are small, and reloading ONE layer of the pair later must rebuild these planes)
    if (TailBlock* tb = tail_of_conv3(h, main)) return upload_tail_planes(h, *tb, hi, lo, K);
    return 0;
}

}  // namespace

// =============================================================================================
extern "C" {

int mpx_pack_conv_weights(const mpx_conv_desc* d, const float* w, const float* conv_bias, const float* gamma,
                          const float* beta, const float* mean, const float* var, float eps, uint16_t* w_hi,
                          uint16_t* w_lo, float* scale, float* shift) {
    if (!d || !w || !w_hi || !w_lo || !scale || !shift) return MPX_E_ARG;
    const int cin = d->cin, cout = d->cout, k = d->ksize, K = d->k_packed;
    const bool stem = (cin == 3 && k == kStemK);
    if (k <= 0 || cin <= 0 || cout <= 0 || cout > d->cout_pad || d->cout_pad % 16 != 0) return MPX_E_ARG;
    // K = k*k*cin_pad: the input planes may carry more channels per pixel than the layer reads (small nets pad to 32)
    const int cin_pad = stem ? 0 : K / (k * k);
    if (stem ? (K != kStemK * 32) : (K != k * k * cin_pad || cin_pad < cin || K % 32 != 0)) return MPX_E_ARG;
    std::memset(w_hi, 0, (size_t)d->cout_pad * K * 2);
    std::memset(w_lo, 0, (size_t)d->cout_pad * K * 2);
    for (int co = 0; co < d->cout_pad; ++co) {
        scale[co] = 0.f;
        shift[co] = 0.f;
    }
    for (int co = 0; co < cout; ++co) {
        const float* wc = w + (size_t)co * cin * k * k;
        float mx = 0.f;
        for (int j = 0; j < cin * k * k; ++j) mx = std::fmax(mx, std::fabs(wc[j]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) {
            int ex;
            std::frexp(mx, &ex);          // mx = f * 2^ex, f in [0.5,1)  ->  mx*2^(10-ex) in [512,1024)
            e = 10 - ex;
        }
        auto put = [&](int kk, float v) {
            const float sv = std::ldexp(v, e);
            const half_t hi = (half_t)sv;
            const half_t lo = (half_t)(sv - (float)hi);
            const size_t at = w_packed_index(co, kk, K);
            w_hi[at] = half_bits(hi);
            w_lo[at] = half_bits(lo);
        };
        if (stem) {
            for (int ky = 0; ky < k; ++ky)
                for (int px = 0; px < k; ++px)
                    for (int c = 0; c < 3; ++c) put(ky * 32 + px * 4 + c, wc[((size_t)c * k + ky) * k + px]);
        } else {
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    for (int ci = 0; ci < cin; ++ci)
                        put((ky * k + kx) * cin_pad + ci, wc[((size_t)ci * k + ky) * k + kx]);
        }
        // y = bn(conv(x) + conv_bias) = s * acc + (beta + (conv_bias - mean) * s); without BatchNorm: y = acc + beta (+ conv_bias)
        double s = 1.0, t = 0.0;
        const double cb = conv_bias ? (double)conv_bias[co] : 0.0;
        if (gamma) {
            s = (double)gamma[co] / std::sqrt((double)var[co] + (double)eps);
            t = (double)beta[co] + (cb - (double)mean[co]) * s;
        } else {
            t = (beta ? (double)beta[co] : 0.0) + cb;
        }
        scale[co] = (float)std::ldexp(s, -e);
        shift[co] = (float)t;
    }
    return 0;
}

int mpx_create(int arch_id, int max_batch, int device, mpx_engine** out) {
    if (!out || max_batch <= 0) return MPX_E_ARG;
    *out = nullptr;
    mpx_engine* h = new (std::nothrow) mpx_engine();
    if (!h) return MPX_E_NOMEM;
    h->arch = arch_id; h->max_batch = max_batch; h->device = device;
    int rc = build_topology(h);
    if (rc) { delete h; return rc; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete h; return (int)e; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cus = cus;
    }

    // one arena: scratch | input planes | activation planes | pooled | logits | weights
    const size_t in_elems = h->small ? (size_t)h->img * h->img * kSmallCPad : (size_t)MPX_IMG_PAD * MPX_IMG_PAD * 4;
    const size_t in_plane = round_up((size_t)max_batch * in_elems * 2 + 256, 256);
    const size_t act_plane = round_up((size_t)max_batch * h->act_elems_per_image * 2, 256);
    const size_t pool_plane = round_up((size_t)max_batch * round_up(h->feat, kSmallCPad) * 2, 256);
    const size_t logit_bytes = round_up((size_t)max_batch * h->logit_pitch * 4, 256);
    const size_t k0_bytes = h->small ? round_up((size_t)(2 + 4096 + max_batch) * sizeof(float), 256) : 0;
    size_t wbytes = 0;
    for (const ConvLayer& L : h->convs) {
        wbytes += 2 * round_up((size_t)L.d.cout_pad * L.d.k_packed * 2, 256) + 2 * round_up((size_t)L.d.cout_pad * 4, 256);
        if (L.fuse_main)
            wbytes += 2 * round_up((size_t)L.d.cout_pad * (L.d.cin + h->convs[L.fuse_partner].d.cin) * 2, 256) + 2 * round_up((size_t)L.d.cout_pad * 4, 256);
    }
    for (const TailBlock& tb : h->tails) wbytes += 2 * round_up((size_t)BT_OUT * (tb.ds >= 0 ? 2 : 1) * BT_MID * 2, 256);
    const size_t scratch_bytes = 4096 * sizeof(float);
    // the stem by superposition (ImageNet ResNets): pooled stem planes, fp32 stem weights + BatchNorm vectors, one image's table (worst case:
    // 49 entries per conv output pixel), the bit planes of one staging call
    const bool stemtab = stem_pool_eligible(h);
    const int tab_nmb = (max_batch + 31) / 32 + 1;
    const size_t stem_plane = stemtab ? round_up((size_t)max_batch * ST_POOLED * ST_POOLED * ST_C * 2, 256) : 0;
    const size_t stem_w_bytes = stemtab ? round_up((size_t)ST_TAPS * 3 * ST_C * 4, 256) + 2 * 256 : 0;
    const size_t tab_int_bytes = stemtab ? round_up((size_t)(ST_NPIX + 1) * 4, 256) : 0;
    const size_t tab_lab_bytes = stemtab ? round_up((size_t)ST_MAX_ENTRIES * 4, 256) : 0;
    const size_t tab_vec_bytes = stemtab ? round_up((size_t)ST_MAX_ENTRIES * ST_C * 4, 256) : 0;
    const size_t tab_bits_bytes = stemtab ? round_up((size_t)4096 * tab_nmb * 4, 256) : 0;
    // (the table itself -- 3 * tab_int + tab_lab + tab_vec + tab_bits, 160 MB -- is allocated by the first mpx_stem_table_build: an engine that
    // only ever stages through K0, stem = "conv", never pays for it)
    const size_t stemtab_bytes = 2 * stem_plane + stem_w_bytes;
    h->tab_sizes[0] = tab_int_bytes; h->tab_sizes[1] = tab_lab_bytes; h->tab_sizes[2] = tab_vec_bytes; h->tab_sizes[3] = tab_bits_bytes;
    const size_t total = scratch_bytes + 2 * in_plane + 2 * kActBufs * act_plane + 2 * pool_plane + logit_bytes + k0_bytes + wbytes + stemtab_bytes;
    e = hipMalloc((void**)&h->arena, total);
    if (e != hipSuccess) { delete h; return (int)e; }
    h->arena_bytes = total;
    char* cur = h->arena;
    auto take = [&](size_t n) { char* r = cur; cur += n; return r; };
    h->seg_scratch = (float*)take(scratch_bytes);
    h->in_hi = (half_t*)take(in_plane);
    h->in_lo = (half_t*)take(in_plane);
    for (int b = 0; b < kActBufs; ++b) {
        h->act_hi[b] = (half_t*)take(act_plane);
        h->act_lo[b] = (half_t*)take(act_plane);
    }
    h->pool_hi = (half_t*)take(pool_plane);
    h->pool_lo = (half_t*)take(pool_plane);
    h->logits = (float*)take(logit_bytes);
    if (k0_bytes) h->k0_scratch = (float*)take(k0_bytes);
    if (stemtab) {
        h->stem_hi = (half_t*)take(stem_plane);
        h->stem_lo = (half_t*)take(stem_plane);
        h->stem_w32 = (float*)take(stem_w_bytes - 512);
        h->stem_s32 = (float*)take(256);
        h->stem_t32 = (float*)take(256);
    }
    h->slot_src.assign((size_t)max_batch, 0);
    for (ConvLayer& L : h->convs) {
        const size_t wb = round_up((size_t)L.d.cout_pad * L.d.k_packed * 2, 256);
        const size_t sb = round_up((size_t)L.d.cout_pad * 4, 256);
        L.w_hi = (half_t*)take(wb);
        L.w_lo = (half_t*)take(wb);
        L.scale = (float*)take(sb);
        L.shift = (float*)take(sb);
        if (L.fuse_main) {
            const size_t fb = round_up((size_t)L.d.cout_pad * (L.d.cin + h->convs[L.fuse_partner].d.cin) * 2, 256);
            L.fw_hi = (half_t*)take(fb);
            L.fw_lo = (half_t*)take(fb);
            L.fscale = (float*)take(sb);
            L.fshift = (float*)take(sb);
        }
    }
    for (TailBlock& tb : h->tails) {
        const size_t pb = round_up((size_t)BT_OUT * (tb.ds >= 0 ? 2 : 1) * BT_MID * 2, 256);
        tb.w3p_hi = (half_t*)take(pb);
        tb.w3p_lo = (half_t*)take(pb);
    }
    // the never-written borders (ImageNet) / padding channels (small nets) of the input staging must be zero, and so must
    // the pooled planes' padding channels
    e = hipMemset(h->arena, 0, scratch_bytes + 2 * in_plane);
    if (e == hipSuccess) e = hipMemset(h->pool_hi, 0, 2 * pool_plane);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile0>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile0::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile1>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile1::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile1::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile2>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile2::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile7>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile7::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile2::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile7::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv256_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Conv256::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)convx_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ConvX::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)convw_f16x3_kernel<ConvW::K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvW::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv256p_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Conv256P::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile4>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile4::LDS);
#ifdef MPX_EXPERIMENTAL
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile3>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile3::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_f16x3_kernel<ConvTile5>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile5::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)convp_f16x3_kernel<ConvTile2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile2::RING);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)convp_f16x3_kernel<ConvTile2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvTile2::RING);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)convs_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ConvS::LDS);
#endif
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)btail_f16x3_kernel<BtResC64>, hipFuncAttributeMaxDynamicSharedMemorySize, BtResC64::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)btail_f16x3_kernel<BtResC128>, hipFuncAttributeMaxDynamicSharedMemorySize, BtResC128::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)btail_f16x3_kernel<BtDualC64>, hipFuncAttributeMaxDynamicSharedMemorySize, BtDualC64::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)btail_f16x3_kernel<BtHeadC64>, hipFuncAttributeMaxDynamicSharedMemorySize, BtHeadC64::LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3p_f16x3_kernel<PatchTile0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3p_f16x3_kernel<PatchTile1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3p_f16x3_kernel<PatchTile2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3pp_f16x3_kernel<PatchTile0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3pp_f16x3_kernel<PatchTile2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return (int)e; }
    *out = h;
    return 0;
}

int mpx_destroy(mpx_engine* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    for (ProfRec& r : h->prof_pool) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    if (h->arena) (void)hipFree(h->arena);
    if (h->tab_arena) (void)hipFree(h->tab_arena);
    delete h;
    return 0;
}

const char* mpx_last_error(const mpx_engine* h) { return h ? h->err.c_str() : "null engine"; }
int mpx_max_batch(const mpx_engine* h) { return h ? h->max_batch : MPX_E_ARG; }
int mpx_num_cus(const mpx_engine* h) { return h ? h->num_cus : MPX_E_ARG; }
int mpx_last_conv_kernels(const mpx_engine* h) { return h ? (int)h->last_kernels : MPX_E_ARG; }
size_t mpx_workspace_bytes(const mpx_engine* h) { return h ? h->arena_bytes + h->tab_arena_bytes : 0; }
int mpx_num_convs(const mpx_engine* h) { return h ? (int)h->convs.size() : MPX_E_ARG; }

int mpx_conv_info(const mpx_engine* h, int i, mpx_conv_desc* out) {
    if (!h || !out || i < 0 || i >= (int)h->convs.size()) return MPX_E_ARG;
    *out = h->convs[i].d;
    return 0;
}

int mpx_set_conv_weights(mpx_engine* h, int i, const float* w, const float* conv_bias, const float* gamma, const float* beta,
                         const float* mean, const float* var, float eps) {
    if (!h) return MPX_E_ARG;
    if (i < 0 || i >= (int)h->convs.size() || !w) return fail(h, MPX_E_ARG, "set_conv_weights: bad layer index or null weight");
    ConvLayer& L = h->convs[i];
    const bool has_bn = L.d.bn_name[0] != 0;
    if (has_bn && (!gamma || !beta || !mean || !var)) return fail(h, MPX_E_ARG, "set_conv_weights: BatchNorm tensors missing for %s", L.d.name);
    if (!has_bn) gamma = nullptr;
    if (L.has_bias && has_bn && !conv_bias) return fail(h, MPX_E_ARG, "set_conv_weights: %s is a conv with bias; conv_bias missing", L.d.name);
    const size_t n = (size_t)L.d.cout_pad * L.d.k_packed;
    std::vector<uint16_t> hi(n), lo(n);
    std::vector<float> sc(L.d.cout_pad), sh(L.d.cout_pad);
    int rc = mpx_pack_conv_weights(&L.d, w, conv_bias, gamma, beta, mean, var, eps, hi.data(), lo.data(), sc.data(), sh.data());
    if (rc) return fail(h, rc, "pack failed for %s", L.d.name);
    MPX_SET_DEVICE(h);
    MPX_HIP(h, hipMemcpy(L.w_hi, hi.data(), n * 2, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.w_lo, lo.data(), n * 2, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    MPX_HIP(h, hipMemcpy(L.shift, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    L.loaded = true;
    if (L.is_stem && h->stem_w32) {
        // the stem by superposition works from the fp32 tensors themselves: weights tap-major [(ky * 7 + kx) * 3 + ch][cout], the
        // BatchNorm as y = acc * s + t with s = gamma / sqrt(var + eps), t = beta - mean * s (rounded to fp32 once)
        std::vector<float> wt((size_t)ST_TAPS * 3 * ST_C), bs(ST_C), bt(ST_C);
        for (int co = 0; co < ST_C; ++co) {
            for (int ch = 0; ch < 3; ++ch)
                for (int t = 0; t < ST_TAPS; ++t) wt[((size_t)t * 3 + ch) * ST_C + co] = w[((size_t)co * 3 + ch) * ST_TAPS + t];
            const double sd = (double)gamma[co] / std::sqrt((double)var[co] + (double)eps);
            bs[co] = (float)sd;
            bt[co] = (float)((double)beta[co] - (double)mean[co] * sd);
        }
        MPX_HIP(h, hipMemcpy(h->stem_w32, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
        MPX_HIP(h, hipMemcpy(h->stem_s32, bs.data(), bs.size() * 4, hipMemcpyHostToDevice));
        MPX_HIP(h, hipMemcpy(h->stem_t32, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
        h->stem_w_loaded = true;
        h->tab_S = -1;                  // a table built from the old weights is stale
    }
    if (TailBlock* tb = tail_of_conv3(h, i)) {
        if (tb->ds < 0) {               // identity tail: the permuted copy of this layer's own planes (scale / shift are shared)
            rc = upload_tail_planes(h, *tb, hi, lo, L.d.k_packed);
            if (rc) return rc;
        } else {
            tb->ready = false;          // rebuilt by build_fused below once both layers of the pair are in
        }
    }
    if (L.fuse_partner >= 0) {
        const size_t nw = (size_t)L.d.cout * L.d.cin;       // both layers of a pair are 1x1
        L.hw.assign(w, w + nw);
        L.hgamma.assign(gamma, gamma + L.d.cout); L.hbeta.assign(beta, beta + L.d.cout);
        L.hmean.assign(mean, mean + L.d.cout); L.hvar.assign(var, var + L.d.cout);
        L.heps = eps;
        ConvLayer& P = h->convs[L.fuse_partner];
        const int main = L.fuse_main ? i : L.fuse_partner;
        h->convs[main].fused_loaded = false;
        if (!P.hw.empty()) return build_fused(h, main);
    }
    return 0;
}

int mpx_set_conv_tile(mpx_engine* h, int i, int tile) {
    if (!h) return MPX_E_ARG;
    if (i < 0 || i >= (int)h->convs.size()) return fail(h, MPX_E_ARG, "set_conv_tile: bad layer index");
    ConvLayer& L = h->convs[i];
    if (tile < 0) tile = default_tile(L.d);
    // product ids = what default_tile can return: 0, 1, 2, 4, 6, 7, 9, 10, 12, 13, 14
    bool known = tile == 0 || tile == 1 || tile == 2 || tile == 4 || tile == 6 || tile == 7 || tile == 9 || tile == 10 || tile == 12 || tile == 13 || tile == 14;
#ifdef MPX_EXPERIMENTAL
    known = known || tile == 3 || tile == 5 || tile == 8 || tile == 11;
    if (tile == 11 && !convs_eligible(L))
        return fail(h, MPX_E_ARG, "set_conv_tile: the pixel-stationary expanding-1x1 kernel (11) runs 1x1 stride-1 layers with cout %% 256 == 0 and cin = 256 (%s is not one)", L.d.name);
    if (tile == 8 && (L.is_fc || L.is_stem || L.d.cout < 128))
        return fail(h, MPX_E_ARG, "set_conv_tile: the persistent kernel (8) runs conv layers with cout >= 128 (%s is not one)", L.d.name);
#endif
    if (!known) return fail(h, MPX_E_ARG, "set_conv_tile: unknown tile %d (product ids: 0, 1, 2, 4, 6, 7, 9, 10, 12, 13, 14)", tile);
    if (tile == 13 && !conv256p_eligible(L))
        return fail(h, MPX_E_ARG, "set_conv_tile: the persistent 256x256 kernel (13) runs 1x1 stride-1 layers with cout %% 256 == 0, cin %% 64 == 0 and no residual operand (%s is not one)", L.d.name);
    if (tile == 12 && !patchp_eligible(L.d))
        return fail(h, MPX_E_ARG, "set_conv_tile: the persistent patch kernel (12) needs a 3x3 stride-1 layer with cout >= 128 and no residual operand whose input patch fits the LDS (%s is not one)", L.d.name);
    if (tile == 14 && !convw_eligible(L))
        return fail(h, MPX_E_ARG, "set_conv_tile: the weights-in-registers expanding-1x1 kernel (14) runs 1x1 stride-1 layers with cout %% 256 == 0 and cin = 256 (%s is not one)", L.d.name);
    if (tile == 10 && !convx_eligible(L))
        return fail(h, MPX_E_ARG, "set_conv_tile: the persistent expanding-1x1 kernel (10) runs 1x1 stride-1 layers with cout %% 256 == 0 and cin %% 64 == 0, cin >= 128 (%s is not one)", L.d.name);
    if (tile == 9 && !conv256_eligible(L))
        return fail(h, MPX_E_ARG, "set_conv_tile: the 256x256 kernel (9) runs 1x1 stride-1 layers with cout %% 256 == 0 and cin %% 64 == 0 (%s is not one)", L.d.name);
    if (tile == 6 && !patch_eligible(L.d))
        return fail(h, MPX_E_ARG, "set_conv_tile: the patch kernel (6) needs a 3x3 stride-1 layer whose input patch fits the LDS (%s does not)", L.d.name);
    L.tile = tile;
    return 0;
}

int mpx_get_conv_tile(const mpx_engine* h, int i) {
    if (!h || i < 0 || i >= (int)h->convs.size()) return MPX_E_ARG;
    return h->convs[i].tile;
}

int mpx_weights_complete(const mpx_engine* h) {
    if (!h) return MPX_E_ARG;
    for (const ConvLayer& L : h->convs)
        if (!L.loaded) return 0;
    return 1;
}

int mpx_mask_apply_normalize(mpx_engine* h, const uint8_t* img_u8_hwc, const float* img_f32_chw, const int32_t* seg,
                             const uint8_t* onoff, int M, int S, const float mean[3], const float std[3], int slot0,
                             float* out_f32_nchw, void* stream) {
    if (!h) return MPX_E_ARG;
    if (h->small) return fail(h, MPX_E_STATE, "mask_apply_normalize: this engine runs one of the small networks; use mpx_mask_apply_minmax");
    if ((img_u8_hwc == nullptr) == (img_f32_chw == nullptr))
        return fail(h, MPX_E_ARG, "mask_apply_normalize: exactly one of img_u8_hwc / img_f32_chw must be given");
    if (!seg || !onoff || M <= 0 || S <= 0) return fail(h, MPX_E_ARG, "mask_apply_normalize: null input or empty M/S");
    if (img_u8_hwc && (!mean || !std)) return fail(h, MPX_E_ARG, "mask_apply_normalize: mean/std required for u8 input");
    if (slot0 < 0 || slot0 + M > h->max_batch) return fail(h, MPX_E_STATE, "mask_apply_normalize: slots [%d,%d) exceed max_batch %d", slot0, slot0 + M, h->max_batch);
    const size_t lds = round_up((size_t)K0_MT * S, 16);
    if (lds > 64 * 1024) return fail(h, MPX_E_ARG, "mask_apply_normalize: S=%d too large (max %d)", S, 64 * 1024 / K0_MT);
    MaskParams p;
    std::memset(&p, 0, sizeof p);
    p.img_u8 = img_u8_hwc; p.img_f32 = img_f32_chw; p.seg = seg; p.onoff = onoff;
    p.out_hi = h->in_hi; p.out_lo = h->in_lo; p.out_f32 = out_f32_nchw;
    for (int c = 0; c < 3; ++c) {
        p.mean[c] = mean ? mean[c] : 0.f;
        p.std[c] = std ? std[c] : 1.f;
    }
    p.M = M; p.S = S; p.slot0 = slot0;
    p.size = MPX_IMG; p.pad_size = MPX_IMG_PAD; p.border = 3;
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 1, -1);
    dim3 grid((MPX_IMG * MPX_IMG + 255) / 256, (M + K0_MT - 1) / K0_MT);
    hipLaunchKernelGGL(mask_apply_normalize_kernel, grid, dim3(256), lds, st, p);
    MPX_HIP(h, hipGetLastError());
    std::fill(h->slot_src.begin() + slot0, h->slot_src.begin() + slot0 + M, (uint8_t)1);
    return 0;
}

int mpx_stem_table_build(mpx_engine* h, const uint8_t* img_u8_hwc, const float* img_f32_chw, const int32_t* seg, int S,
                         const float mean[3], const float std[3], void* stream) {
    if (!h) return MPX_E_ARG;
    if (!h->stem_w32) return fail(h, MPX_E_STATE, "stem_table_build: this architecture has no 7x7 stem with a max pool");
    if ((img_u8_hwc == nullptr) == (img_f32_chw == nullptr))
        return fail(h, MPX_E_ARG, "stem_table_build: exactly one of img_u8_hwc / img_f32_chw must be given");
    if (!seg || S <= 0 || S > 4096) return fail(h, MPX_E_ARG, "stem_table_build: null label map or S outside [1, 4096]");
    if (img_u8_hwc && (!mean || !std)) return fail(h, MPX_E_ARG, "stem_table_build: mean/std required for u8 input");
    if (!h->stem_w_loaded) return fail(h, MPX_E_STATE, "stem_table_build: layer 0 (%s) has no weights", h->convs[0].d.name);
    MPX_SET_DEVICE(h);
    if (!h->tab_arena) {            // first table of this engine: its 160 MB are allocated now, once (the only allocation behind the boundary after mpx_create)
        const size_t total = 3 * h->tab_sizes[0] + h->tab_sizes[1] + h->tab_sizes[2] + h->tab_sizes[3];
        MPX_HIP(h, hipMalloc((void**)&h->tab_arena, total));
        h->tab_arena_bytes = total;
        char* cur = h->tab_arena;
        auto take = [&](size_t n) { char* r = cur; cur += n; return r; };
        h->tab_cnt = (int*)take(h->tab_sizes[0]);
        h->tab_off = (int*)take(h->tab_sizes[0]);
        h->tab_heavy = (int*)take(h->tab_sizes[0]);
        h->tab_lab = (int*)take(h->tab_sizes[1]);
        h->tab_vec = (float*)take(h->tab_sizes[2]);
        h->tab_bits = (unsigned*)take(h->tab_sizes[3]);
    }
    StemTabParams p;
    std::memset(&p, 0, sizeof p);
    p.img_u8 = img_u8_hwc; p.img_f32 = img_f32_chw; p.seg = seg; p.S = S;
    for (int c = 0; c < 3; ++c) {
        p.mean[c] = mean ? mean[c] : 0.f;
        p.std[c] = std ? std[c] : 1.f;
    }
    p.w = h->stem_w32; p.cnt = h->tab_cnt; p.off = h->tab_off; p.lab = h->tab_lab; p.vec = h->tab_vec;
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 1, -1);
    hipLaunchKernelGGL(stemtab_count_kernel, dim3(ST_NPIX / 4), dim3(256), 0, st, p);
    MPX_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(stemtab_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)h->tab_cnt, h->tab_off);
    MPX_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(stemtab_fill_kernel, dim3(ST_NPIX / 4), dim3(256), 0, st, p);
    MPX_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(stemtab_heavy_kernel, dim3(1), dim3(1024), 0, st, (const int*)h->tab_off, h->tab_heavy, h->tab_heavy + ST_POOLED * ST_POOLED);
    MPX_HIP(h, hipGetLastError());
    h->tab_S = S;
    return 0;
}

int mpx_stem_table_apply(mpx_engine* h, const uint8_t* onoff, int M, int S, int slot0, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!h->stem_w32) return fail(h, MPX_E_STATE, "stem_table_apply: this architecture has no 7x7 stem with a max pool");
    if (!onoff || M <= 0) return fail(h, MPX_E_ARG, "stem_table_apply: null mask rows or empty M");
    if (h->tab_S < 0) return fail(h, MPX_E_STATE, "stem_table_apply: no table in place (mpx_stem_table_build first; loading layer 0 again discards it)");
    if (S != h->tab_S) return fail(h, MPX_E_ARG, "stem_table_apply: S=%d, the table was built with S=%d", S, h->tab_S);
    if (slot0 < 0 || slot0 + M > h->max_batch) return fail(h, MPX_E_STATE, "stem_table_apply: slots [%d,%d) exceed max_batch %d", slot0, slot0 + M, h->max_batch);
    const int nmb = (M + 31) / 32;
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    {
        ProfScope ps(h, st, 1, -1);
        hipLaunchKernelGGL(onoff_bitplanes_kernel, dim3((S * nmb + 255) / 256), dim3(256), 0, st, onoff, M, S, nmb, h->tab_bits);
        MPX_HIP(h, hipGetLastError());
    }
    StemApplyParams p;
    std::memset(&p, 0, sizeof p);
    p.off = h->tab_off; p.lab = h->tab_lab; p.vec = h->tab_vec; p.bits = h->tab_bits; p.s = h->stem_s32; p.t = h->stem_t32;
    p.out_hi = h->stem_hi; p.out_lo = h->stem_lo; p.nmb = nmb; p.M = M; p.slot0 = slot0;
    p.heavy_list = h->tab_heavy; p.heavy_count = h->tab_heavy + ST_POOLED * ST_POOLED;
    {
        ProfScope ps(h, st, OP_CONV, 0);        // the stem's work: booked on layer 0 like the stem + pool launch it replaces
        // the pooled pixels with many superpixels under one window first (their workgroups share a pixel's masks), then every pixel
        hipLaunchKernelGGL(stem_apply_heavy_kernel, dim3(h->num_cus, nmb), dim3(256), 0, st, p);
        MPX_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(stem_apply_kernel, dim3(ST_POOLED * ST_POOLED / 4, nmb), dim3(256), 0, st, p);
        MPX_HIP(h, hipGetLastError());
    }
    std::fill(h->slot_src.begin() + slot0, h->slot_src.begin() + slot0 + M, (uint8_t)2);
    return 0;
}

int mpx_mask_apply_minmax(mpx_engine* h, const float* img_f32_chw, const int32_t* seg, const uint8_t* removed, int M, int S,
                          int slot0, float* out_f32_nchw, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!h->small) return fail(h, MPX_E_STATE, "mask_apply_minmax: only the small networks (MNIST net, CIFAR ResNet) use this mask convention");
    if (!img_f32_chw || !seg || !removed || M <= 0 || S <= 0 || S > 4096) return fail(h, MPX_E_ARG, "mask_apply_minmax: null input, empty M/S or S > 4096");
    if (slot0 < 0 || slot0 + M > h->max_batch) return fail(h, MPX_E_STATE, "mask_apply_minmax: slots [%d,%d) exceed max_batch %d", slot0, slot0 + M, h->max_batch);
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 1, -1);
    const int hw = h->img * h->img;
    float* stats = h->k0_scratch;
    float* mask_max = h->k0_scratch + 2 + 4096;
    hipLaunchKernelGGL(smallnet_image_stats_kernel, dim3(1), dim3(256), 0, st, img_f32_chw, seg, h->in_ch, hw, S, stats);
    MPX_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(smallnet_mask_max_kernel, dim3((M + 255) / 256), dim3(256), 0, st, removed, M, S, (const float*)stats, mask_max);
    MPX_HIP(h, hipGetLastError());
    SmallMaskParams p;
    std::memset(&p, 0, sizeof p);
    p.img = img_f32_chw; p.seg = seg; p.removed = removed; p.stats = stats; p.mask_max = mask_max;
    p.out_hi = h->in_hi; p.out_lo = h->in_lo; p.out_f32 = out_f32_nchw;
    p.C = h->in_ch; p.hw = hw; p.M = M; p.S = S; p.slot0 = slot0;
    hipLaunchKernelGGL(smallnet_mask_apply_kernel, dim3((hw + 255) / 256, M), dim3(256), 0, st, p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_avgpool2_pad(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int B, int hin,
                     int cin_p, int cout_p, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!in_hi || !in_lo || !out_hi || !out_lo || B <= 0 || hin <= 0 || (hin & 1) || cin_p <= 0 || (cin_p & 7) || cout_p < cin_p || (cout_p & 7))
        return fail(h, MPX_E_ARG, "avgpool2_pad: bad arguments (hin even, channel counts multiples of 8, cout_p >= cin_p)");
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 2, -1);
    const size_t total = (size_t)B * (hin / 2) * (hin / 2) * (cout_p / 8);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(avgpool2_pad_kernel, dim3(grid), dim3(256), 0, st, (const half_t*)in_hi, (const half_t*)in_lo,
                       (half_t*)out_hi, (half_t*)out_lo, B, hin, cin_p, cout_p);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_geometry(const mpx_engine* h, int* image_size, int* in_channels, int* num_classes, int* logit_pitch) {
    if (!h) return MPX_E_ARG;
    if (image_size) *image_size = h->img;
    if (in_channels) *in_channels = h->in_ch;
    if (num_classes) *num_classes = h->ncls;
    if (logit_pitch) *logit_pitch = h->logit_pitch;
    return 0;
}

int mpx_conv_bn_act(mpx_engine* h, int i, const void* in_hi, const void* in_lo, const void* res_hi, const void* res_lo,
                    void* out_hi, void* out_lo, float* out_f32, int B, void* stream) {
    if (!h) return MPX_E_ARG;
    if (i < 0 || i >= (int)h->convs.size() || B <= 0) return fail(h, MPX_E_ARG, "conv_bn_act: bad layer index or batch");
    const ConvLayer& L = h->convs[i];
    if (L.is_stem) {
        if (in_hi || in_lo) return fail(h, MPX_E_ARG, "conv_bn_act: layer 0 reads the engine input staging; pass NULL inputs");
        if (B > h->max_batch) return fail(h, MPX_E_STATE, "conv_bn_act: B > max_batch");
    } else if (h->small && i == 0 && !in_hi && !in_lo) {
        in_hi = h->in_hi;               // the small networks' first conv reads the engine's [B][H][W][32] staging
        in_lo = h->in_lo;
        if (B > h->max_batch) return fail(h, MPX_E_STATE, "conv_bn_act: B > max_batch");
    } else if (!in_hi || !in_lo) {
        return fail(h, MPX_E_ARG, "conv_bn_act: null input planes");
    }
    if ((res_hi == nullptr) != (res_lo == nullptr)) return fail(h, MPX_E_ARG, "conv_bn_act: residual planes must come in pairs");
    if (L.is_fc ? (out_f32 == nullptr) : (out_f32 != nullptr || !out_hi || !out_lo))
        return fail(h, MPX_E_ARG, "conv_bn_act: fc writes out_f32, every other layer writes out_hi/out_lo");
    MPX_SET_DEVICE(h);
    return launch_conv(h, i, (const half_t*)in_hi, (const half_t*)in_lo, (const half_t*)res_hi, (const half_t*)res_lo,
                       (half_t*)out_hi, (half_t*)out_lo, out_f32, B, as_stream(stream));
}

int mpx_maxpool3x3s2(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int B, int hin,
                     int c, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!in_hi || !in_lo || !out_hi || !out_lo || B <= 0 || hin <= 0 || (hin & 1) || c <= 0 || (c & 7))
        return fail(h, MPX_E_ARG, "maxpool3x3s2: bad arguments (hin even, c multiple of 8)");
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 2, -1);
    const size_t total = (size_t)B * (hin / 2) * (hin / 2) * (c / 8);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, st, (const half_t*)in_hi, (const half_t*)in_lo,
                       (half_t*)out_hi, (half_t*)out_lo, B, hin, c);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_global_avgpool(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int B, int hw,
                       int c, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!in_hi || !in_lo || !out_hi || !out_lo || B <= 0 || hw <= 0 || c <= 0 || (c & 7))
        return fail(h, MPX_E_ARG, "global_avgpool: bad arguments (c multiple of 8)");
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 2, -1);
    const int total = B * (c / 8);
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const half_t*)in_hi,
                       (const half_t*)in_lo, (half_t*)out_hi, (half_t*)out_lo, B, hw, c);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_head_softmax_gather(mpx_engine* h, const float* logits, const int32_t* label, float* score, int32_t* pred,
                            int B, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!logits || !label || !score || !pred || B <= 0) return fail(h, MPX_E_ARG, "head_softmax_gather: null pointer or empty batch");
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    ProfScope ps(h, st, 3, -1);
    hipLaunchKernelGGL(head_softmax_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, label, score, pred, B,
                       h->ncls, h->logit_pitch);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_forward(mpx_engine* h, const int32_t* label, float* score, int32_t* pred, float* logits_out, int B,
                void* stream) {
    if (!h) return MPX_E_ARG;
    if (!label || !score || !pred || B <= 0) return fail(h, MPX_E_ARG, "forward: null pointer or empty batch");
    if (B > h->max_batch) return fail(h, MPX_E_STATE, "forward: B=%d exceeds max_batch=%d", B, h->max_batch);
    if (mpx_weights_complete(h) != 1) return fail(h, MPX_E_STATE, "forward: weights not loaded for every layer");
    float* logits = logits_out ? logits_out : h->logits;
    MPX_HIP(h, hipSetDevice(h->device));
    struct InForward {
        mpx_engine* e;
        explicit InForward(mpx_engine* e_) : e(e_) { e->in_forward = true; }
        ~InForward() { e->in_forward = false; }
    } scope(h);
    // (BUF_INPUT stays NULL: layer 0 reads the engine's own staging, mpx_conv_bn_act)
    auto hi = [&](int b) -> half_t* { return b == BUF_POOL ? h->pool_hi : (b == BUF_STEM ? h->stem_hi : (b >= 0 ? h->act_hi[b] : nullptr)); };
    auto lo = [&](int b) -> half_t* { return b == BUF_POOL ? h->pool_lo : (b == BUF_STEM ? h->stem_lo : (b >= 0 ? h->act_lo[b] : nullptr)); };
    int rc = 0;
    bool skip_pool = false;
    // how the B slots were staged: by K0 into the input staging (the stem conv + max pool run here) or by mpx_stem_table_apply, which has
    // written the pooled stem planes already (the stem and the pool are skipped).  A batch staged by both is an error, not a guess.
    bool stem_done = false;
    {
        int n_tab = 0, n_k0 = 0;
        for (int i = 0; i < B; ++i) {
            n_tab += h->slot_src[i] == 2;
            n_k0 += h->slot_src[i] == 1;
        }
        if (n_tab && n_tab != B)
            return fail(h, MPX_E_STATE, "forward: %d of the %d slots were staged by mpx_stem_table_apply and %d by mpx_mask_apply_normalize; one forward takes one kind", n_tab, B, n_k0);
        stem_done = n_tab == B;
    }
    // layer1's block tails as single launches (mpx_btail.h) whenever their planes are in and no tile override asks for the
    // layer-by-layer kernels of those layers
    bool use_bt = h->fuse_bt && h->fuse_ds && tails_ready(h);
    for (const TailBlock& tb : h->tails)
        for (int li : {tb.c1, tb.c2, tb.c3, tb.next1})
            use_bt = use_bt && h->convs[li].tile == default_tile(h->convs[li].d);
    const std::vector<Op>& ops = use_bt ? h->ops_bt : h->ops;
    for (size_t oi = 0; oi < ops.size(); ++oi) {
        const Op& o = ops[oi];
        if (stem_done && ((o.kind == OP_CONV && o.conv == 0) || (o.kind == OP_MAXPOOL && o.out == BUF_STEM))) continue;
        switch (o.kind) {
            case OP_BTAIL:
                rc = launch_btail(h, o.conv, hi(o.in), lo(o.in), hi(o.res), lo(o.res), hi(o.out), lo(o.out), hi(o.z), lo(o.z), B, as_stream(stream));
                break;
            case OP_CONV:
                if (h->fuse_pool && o.conv == 0 && oi + 1 < ops.size() && ops[oi + 1].kind == OP_MAXPOOL && stem_pool_eligible(h) &&
                    h->convs[0].tile == 1) {
                    const Op& pool = ops[oi + 1];        // the stem writes the pooled planes; the pool op is skipped
                    rc = launch_stem_pool(h, hi(pool.out), lo(pool.out), B, as_stream(stream));
                    skip_pool = true;
                    break;
                }
                if (h->fuse_ds && h->convs[o.conv].fuse_partner >= 0) {
                    const ConvLayer& CL = h->convs[o.conv];
                    const ConvLayer& MAIN = CL.fuse_main ? CL : h->convs[CL.fuse_partner];
                    if (MAIN.fused_loaded && (MAIN.tile == 2 || MAIN.tile == 7 || MAIN.tile == 8 || MAIN.tile == 10 || MAIN.tile == 11 || MAIN.tile == 14)) {      // (8, 11: experimental builds)
                        if (!CL.fuse_main) break;       // the downsample conv runs inside its main conv's launch
                        rc = launch_conv_fused(h, o.conv, hi(o.in), lo(o.in), hi(o.in2), lo(o.in2), hi(o.out), lo(o.out), B, as_stream(stream));
                        break;
                    }
                }
                if (h->convs[o.conv].is_fc)
                    rc = mpx_conv_bn_act(h, o.conv, hi(o.in), lo(o.in), nullptr, nullptr, nullptr, nullptr, logits, B, stream);
                else
                    rc = mpx_conv_bn_act(h, o.conv, hi(o.in), lo(o.in), hi(o.res), lo(o.res), hi(o.out), lo(o.out), nullptr, B, stream);
                break;
            case OP_MAXPOOL:
                if (skip_pool) { skip_pool = false; break; }
                rc = mpx_maxpool3x3s2(h, hi(o.in), lo(o.in), hi(o.out), lo(o.out), B, o.hin, o.c, stream);
                break;
            case OP_AVGPOOL: rc = mpx_global_avgpool(h, hi(o.in), lo(o.in), hi(o.out), lo(o.out), B, o.hin * o.hin, o.c, stream); break;
            case OP_HEAD: rc = mpx_head_softmax_gather(h, logits, label, score, pred, B, stream); break;
            case OP_AVGPAD: rc = mpx_avgpool2_pad(h, hi(o.in), lo(o.in), hi(o.out), lo(o.out), B, o.hin, o.c >> 16, o.c & 0xffff, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

int mpx_stem_conv_maxpool(mpx_engine* h, void* out_hi, void* out_lo, int B, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!out_hi || !out_lo || B <= 0) return fail(h, MPX_E_ARG, "stem + max pool: null pointer or empty batch");
    if (B > h->max_batch) return fail(h, MPX_E_STATE, "stem + max pool: B=%d exceeds max_batch=%d", B, h->max_batch);
    MPX_SET_DEVICE(h);
    return launch_stem_pool(h, (half_t*)out_hi, (half_t*)out_lo, B, as_stream(stream));
}

int mpx_set_fusion(mpx_engine* h, int mask) {
    if (!h) return MPX_E_ARG;
    h->fuse_pool = (mask & 1) != 0;
    h->fuse_ds = (mask & 1) != 0;
    h->fuse_bt = (mask & 2) != 0;
    return 0;
}

int mpx_bottleneck_tail(mpx_engine* h, int i, const void* t1_hi, const void* t1_lo, const void* x_hi, const void* x_lo,
                        void* out_hi, void* out_lo, void* next_hi, void* next_lo, int B, void* stream) {
    if (!h) return MPX_E_ARG;
    if (B <= 0 || (t1_hi == nullptr) != (t1_lo == nullptr) || !x_hi || !x_lo || !out_hi || !out_lo || !next_hi || !next_lo)
        return fail(h, MPX_E_ARG, "bottleneck_tail: null planes or empty batch");
    int ti = -1;
    for (size_t k = 0; k < h->tails.size(); ++k)
        if (h->tails[k].c2 == i) ti = (int)k;
    if (ti < 0) return fail(h, MPX_E_ARG, "bottleneck_tail: layer %d is not the conv2 of a block whose tail runs as one launch", i);
    {
        // the four plane pairs must not overlap: workgroups read t1's halo and the identity while others write out / next
        const TailBlock& tb = h->tails[ti];
        const size_t px = (size_t)B * h->convs[tb.c2].d.hin * h->convs[tb.c2].d.hin * sizeof(half_t);
        const size_t c_x = tb.ds >= 0 ? BT_MID : BT_OUT, c_next = (size_t)h->convs[tb.next1].d.cout;
        struct Range { const char* lo; size_t n; const char* what; };
        const Range r[8] = {{(const char*)t1_hi, px * BT_MID, "t1_hi"}, {(const char*)t1_lo, px * BT_MID, "t1_lo"},
                            {(const char*)x_hi, px * c_x, "x_hi"}, {(const char*)x_lo, px * c_x, "x_lo"},
                            {(const char*)out_hi, px * BT_OUT, "out_hi"}, {(const char*)out_lo, px * BT_OUT, "out_lo"},
                            {(const char*)next_hi, px * c_next, "next_hi"}, {(const char*)next_lo, px * c_next, "next_lo"}};
        for (int a = 0; a < 8; ++a)
            for (int b = a + 1; b < 8; ++b)
                if (r[a].lo && r[b].lo && r[a].lo < r[b].lo + r[b].n && r[b].lo < r[a].lo + r[a].n)
                    return fail(h, MPX_E_ARG, "bottleneck_tail: planes %s and %s overlap (all plane pairs must be distinct buffers)", r[a].what, r[b].what);
    }
    MPX_SET_DEVICE(h);
    return launch_btail(h, ti, (const half_t*)t1_hi, (const half_t*)t1_lo, (const half_t*)x_hi, (const half_t*)x_lo, (half_t*)out_hi,
                        (half_t*)out_lo, (half_t*)next_hi, (half_t*)next_lo, B, as_stream(stream));
}

int mpx_num_bottleneck_tails(const mpx_engine* h) { return h ? (int)h->tails.size() : MPX_E_ARG; }

int mpx_bottleneck_tail_info(const mpx_engine* h, int k, int* conv2, int* conv3, int* downsample, int* next_conv1) {
    if (!h || k < 0 || k >= (int)h->tails.size()) return MPX_E_ARG;
    if (conv2) *conv2 = h->tails[k].c2;         // (the block's own conv1 is the layer in front of it)
    if (conv3) *conv3 = h->tails[k].c3;
    if (downsample) *downsample = h->tails[k].ds;
    if (next_conv1) *next_conv1 = h->tails[k].next1;
    return 0;
}

int mpx_conv_dual_bn_act(mpx_engine* h, int i, const void* in_hi, const void* in_lo, const void* x_hi, const void* x_lo,
                         void* out_hi, void* out_lo, int B, void* stream) {
    if (!h) return MPX_E_ARG;
    if (i < 0 || i >= (int)h->convs.size() || B <= 0) return fail(h, MPX_E_ARG, "conv_dual_bn_act: bad layer index or batch");
    if (!h->convs[i].fuse_main) return fail(h, MPX_E_ARG, "conv_dual_bn_act: layer %d (%s) is not the last conv of a block with a downsample branch", i, h->convs[i].d.name);
    if (!in_hi || !in_lo || !x_hi || !x_lo || !out_hi || !out_lo) return fail(h, MPX_E_ARG, "conv_dual_bn_act: null planes");
    MPX_SET_DEVICE(h);
    return launch_conv_fused(h, i, (const half_t*)in_hi, (const half_t*)in_lo, (const half_t*)x_hi, (const half_t*)x_lo,
                             (half_t*)out_hi, (half_t*)out_lo, B, as_stream(stream));
}

int mpx_heatmap_accumulate(mpx_engine* h, const int32_t* seg, const uint8_t* onoff, const int32_t* pred,
                           const int32_t* label, int M, int S, float* heat, void* stream) {
    if (!h) return MPX_E_ARG;
    if (!seg || !onoff || !pred || !label || !heat || M <= 0 || S <= 0 || S > 4096)
        return fail(h, MPX_E_ARG, "heatmap_accumulate: null pointer, empty M/S or S > 4096");
    MPX_SET_DEVICE(h);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(heatmap_segment_count_kernel, dim3((S + 255) / 256), dim3(256), 0, st, onoff, pred, label, M, S,
                       h->seg_scratch);
    MPX_HIP(h, hipGetLastError());
    const int npix = h->img * h->img;
    hipLaunchKernelGGL(heatmap_gather_kernel, dim3((npix + 255) / 256), dim3(256), 0, st, seg, h->seg_scratch, S, npix, heat);
    MPX_HIP(h, hipGetLastError());
    return 0;
}

int mpx_input_planes(const mpx_engine* h, void** hi, void** lo) {
    if (!h || !hi || !lo) return MPX_E_ARG;
    *hi = h->in_hi;             // a pure getter: how the slots were staged (slot_src) is not touched -- a diagnostic call between
    *lo = h->in_lo;             // mpx_stem_table_apply and mpx_forward must not turn the forward onto stale input planes
    return 0;
}

int mpx_mark_input_staged(mpx_engine* h, int slot0, int M) {
    if (!h) return MPX_E_ARG;
    if (slot0 < 0 || M <= 0 || (long long)slot0 + M > h->max_batch)
        return fail(h, MPX_E_ARG, "mark_input_staged: slots [%d, %d) outside [0, max_batch=%d)", slot0, slot0 + M, h->max_batch);
    if (h->in_forward) return fail(h, MPX_E_STATE, "mark_input_staged: called during mpx_forward");
    // a caller that has WRITTEN the input planes of these slots by hand (mpx_input_planes) says so: the next mpx_forward runs the stem
    // conv + max pool on them, as after mpx_mask_apply_normalize, even if an earlier batch of the same slots came from mpx_stem_table_apply
    std::fill(h->slot_src.begin() + slot0, h->slot_src.begin() + slot0 + M, (uint8_t)1);
    return 0;
}

int mpx_stem_planes(const mpx_engine* h, void** hi, void** lo) {
    if (!h || !hi || !lo) return MPX_E_ARG;
    *hi = h->stem_hi;               // NULL for the small networks
    *lo = h->stem_lo;
    return 0;
}

int mpx_profile_enable(mpx_engine* h, int on) {
    if (!h) return MPX_E_ARG;
    if (on && h->prof_pool.empty()) {
        MPX_SET_DEVICE(h);
        h->prof_pool.resize(kProfilePairs);
        for (ProfRec& r : h->prof_pool) {
            MPX_HIP(h, hipEventCreate(&r.t0));
            MPX_HIP(h, hipEventCreate(&r.t1));
        }
    }
    h->prof_on = on != 0;
    return 0;
}

int mpx_profile_collect(mpx_engine* h, double ms_by_kind[4], long long launches_by_kind[4], double* per_conv_ms) {
    if (!h || !ms_by_kind || !launches_by_kind) return MPX_E_ARG;
    if (h->prof_used == 0) return 0;
    MPX_SET_DEVICE(h);
    MPX_HIP(h, hipEventSynchronize(h->prof_pool[h->prof_used - 1].t1));
    for (int i = 0; i < h->prof_used; ++i) {
        ProfRec& r = h->prof_pool[i];
        float ms = 0.f;
        MPX_HIP(h, hipEventElapsedTime(&ms, r.t0, r.t1));
        ms_by_kind[r.kind] += ms;
        launches_by_kind[r.kind] += 1;
        if (per_conv_ms && r.kind == OP_CONV && r.conv >= 0) per_conv_ms[r.conv] += ms;
    }
    h->prof_used = 0;
    return 0;
}

#ifdef MPX_DIAG
// diagnostic build only (not declared in include/mpx.h): DEV u64[8 * workgroups of the next conv launches]
int mpx_debug_set_stamps(mpx_engine* h, void* dev_buf) {
    if (!h) return MPX_E_ARG;
    h->stamps = (unsigned long long*)dev_buf;
    return 0;
}
#endif

double mpx_flops_per_forward(const mpx_engine* h) {
    if (!h) return 0.0;
    double macs = 0.0;
    for (const ConvLayer& L : h->convs)
        macs += (double)L.d.hout * L.d.hout * L.d.cout * L.d.cin * L.d.ksize * L.d.ksize;
    return 2.0 * macs;
}

}  // extern "C"
