/* mpx.h -- C-ABI of the MI355X masked-perturbation scoring engine ("mpx").
 *
 * Drop-in boundary for ONE path of LiliMeng/network_interpretation_imagenet: for one
 * 224x224 image, apply M superpixel on/off mask-vectors to the normalised image, run one
 * ResNet forward per mask, return softmax(logits)[label] and argmax(logits) per mask.
 * The reference has no FFI layer (it is 100 % Python); every entry point below names the
 * reference lines it replaces.  Paths are relative to the reference repository root.
 *
 * Conventions
 *   - Plain C types only.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Pointers marked DEV are device pointers owned by the caller; HOST are host pointers.
 *     The engine never frees or retains caller memory; it owns only its workspace.
 *   - Every function returns 0 on success, a positive hipError_t, or a negative MPX_E_* code,
 *     never throws, never exits, and (unless stated) does not synchronise the stream.
 *   - Activations between kernels live in HBM as TWO fp16 planes in NHWC order ("split-fp16":
 *     x ~= hi + lo, 22 significant bits, 4 bytes per element like fp32).  Convolutions run on
 *     the fp16 MFMA pipe as hi*hi + hi*lo + lo*hi with fp32 accumulation.
 *   - One engine per process per GPU.  Calls on one engine must not overlap from two threads.
 */
#ifndef MPX_H
#define MPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPX_IMG 224          /* n = 224, generate_gp_training_data_imagenet.py:84-88 */
#define MPX_IMG_PAD 230      /* 3-pixel zero border, the 7x7 stem's padding=3, stored explicitly */
#define MPX_NUM_CLASSES 1000

enum {
    MPX_E_ARG = -1,      /* bad argument (null pointer, shape, range) */
    MPX_E_STATE = -2,    /* weights missing, batch larger than max_batch, ... */
    MPX_E_NOMEM = -3,
    MPX_E_INTERNAL = -4
};

/* arch_id = the torchvision ResNet depth the reference selects with `-a` / `--arch`
 * (generate_gp_training_data_imagenet.py:45,579): 18, 34, 50, 101 or 152 -- or one of the reference's two small networks
 * (SURVEY.md 8 f4), whose trained checkpoints ship with it:
 *   MPX_ARCH_MNIST_NET            Classification_Net, 28x28x1 -> 10 (generate_gp_training_data_mnist.py:86-105)
 *   MPX_ARCH_CIFAR_RESNET + depth ResNetCifar(depth = 6n+2), 32x32x3 -> 10 (models/resnet.py:77-146; the checkpoint is depth 56)
 * Small-network engines stage inputs with mpx_mask_apply_minmax (their scorers' mask convention) instead of
 * mpx_mask_apply_normalize, keep activations as NHWC planes with channels padded to a multiple of 32, and score 10 classes
 * (logit rows are 16 floats apart: mpx_geometry). */
#define MPX_ARCH_MNIST_NET 1
#define MPX_ARCH_CIFAR_RESNET 2000
typedef struct mpx_engine mpx_engine;

typedef struct mpx_conv_desc {
    char name[48];       /* torchvision state_dict prefix of the conv ("layer3.4.conv2", "fc") */
    char bn_name[48];    /* prefix of its BatchNorm ("layer3.4.bn2"); "" for fc */
    int32_t cin, cout, ksize, stride, pad;
    int32_t hin, hout;   /* square spatial sizes at 224x224 input */
    int32_t relu;        /* ReLU in the epilogue */
    int32_t residual;    /* adds the block identity before the ReLU */
    int32_t k_packed;    /* K of the packed [cout_pad][k_packed] fp16 weight planes */
    int32_t cout_pad;    /* rows of the packed planes (multiple of 128) */
} mpx_conv_desc;

/* ---- engine life cycle -------------------------------------------------------------------
 * replaces: models.__dict__[args.arch](pretrained=True); model.cuda(); model.eval()
 *           (generate_gp_training_data_imagenet.py:579-580,159).  Allocates the whole workspace
 *           (input staging for max_batch masked images, activation planes, logits) once. */
int mpx_create(int arch_id, int max_batch, int device, mpx_engine** out);
int mpx_destroy(mpx_engine* h);
const char* mpx_last_error(const mpx_engine* h);   /* "" if none; valid until next call */
int mpx_max_batch(const mpx_engine* h);
/* Compute units of the engine's device (hipDeviceAttributeMultiprocessorCount, read by mpx_create): the persistent kernels launch
 * one workgroup per CU, so a forward batch is best a whole number of `num_cus * 256`-pixel rounds of the 14x14 maps
 * (engine.whole_round_batch). */
int mpx_num_cus(const mpx_engine* h);
/* 224/3/1000/1000 for the ImageNet ResNets, 28/1/10/16 and 32/3/10/16 for the small networks; any pointer may be NULL */
int mpx_geometry(const mpx_engine* h, int* image_size, int* in_channels, int* num_classes, int* logit_pitch);
size_t mpx_workspace_bytes(const mpx_engine* h);

/* ---- topology / weights ------------------------------------------------------------------
 * Layer i in [0, mpx_num_convs): every conv in forward order, then "fc" as the last entry. */
int mpx_num_convs(const mpx_engine* h);
int mpx_conv_info(const mpx_engine* h, int i, mpx_conv_desc* out);

/* replaces: the state_dict tensors torchvision loads (same line as above).  HOST pointers, f32:
 * w = conv weight OIHW [cout][cin][k][k]; conv_bias = the conv's own bias [cout] or NULL (torchvision's ResNet convs have
 * none; the MNIST net's nn.Conv2d do, generate_gp_training_data_mnist.py:72-77); gamma/beta/mean/var = BatchNorm
 * weight/bias/running_mean/running_var [cout]; eps = 1e-5.  For a layer without BatchNorm (bn_name "": "fc", the MNIST
 * net's "conv6"): beta = its bias, gamma/mean/var/conv_bias = NULL.  Packs on the host (mpx_pack_conv_weights) and
 * uploads synchronously. */
int mpx_set_conv_weights(mpx_engine* h, int i, const float* w, const float* conv_bias, const float* gamma,
                         const float* beta, const float* mean, const float* var, float eps);
int mpx_weights_complete(const mpx_engine* h);     /* 1 when every layer has weights */

/* Kernel variant of layer i (tuning / test hook; results are identical up to fp32 summation order).  The ids are exactly the
 * kernels some layer class runs by default:
 *   0 = 128x256 tile, 8 waves, 3-stage LDS ring, 1 workgroup per CU (wide stride-2 3x3 layers);
 *   1 = 64x256, 4 waves side by side (cout <= 64: the stem and the 64 -> 64 3x3 layers);
 *   2 = 128x128, 4 waves, 2-deep rings, 64 KB, 2 workgroups per CU (strided 1x1 layers, fc, everything without a better fit);
 *   4 = 64x192, 64 KB, 2 workgroups per CU (1x1 layers with cout = 64);
 *   6 = the 3x3 patch kernel (stride-1 3x3 layers whose input patch fits the LDS: the input is staged once per 32-channel chunk
 *       and the 9 taps read it at row offsets);
 *   7 = the 128x128 tile cut into 8 waves of 32x64 (124 VGPRs: two workgroups = 16 waves per CU; expanding 1x1 layers with
 *       K = 64 or on 7x7 maps, and the K-concatenated conv3 + downsample launches);
 *   9 = 256x256 tile, two 64-KB stages (reducing 1x1 stride-1 layers with cout % 256 == 0, cin % 64 == 0);
 *  10 = persistent pipelined 256x128 kernel, three stages running on across tiles (expanding 1x1 stride-1 layers with
 *       cout % 256 == 0, cin >= 128 that tile 14 does not take: 128 -> 512, 512 -> 2048);
 *  14 = the expanding 1x1 kernel whose weights live in registers (csrc/mpx_convw.h): one persistent 4-wave workgroup per CU on
 *       256 x 64 tiles, every wave keeps the 64 x 256 x (hi + lo) weights of its channels in 256 AGPRs, the LDS holds only two
 *       whole pixel tiles (1x1 stride-1 layers with cout % 256 == 0 and cin = 256: the default of 256 -> 1024 on 14x14 maps;
 *       bit-identical to 7 and 10);
 *  13 = the 256x256 kernel (9) as ONE persistent workgroup per CU: the two-stage ring runs on across tiles, register epilogue (layers
 *       eligible for 9 without a residual operand: their default; bit-identical to 9);
 *  12 = the patch kernel (6) as ONE persistent workgroup per CU: weight ring and patch buffers run on across tiles, register
 *       epilogue (layers eligible for 6 with cout >= 128 and no residual operand; the default on 28x28 / 14x14 maps; bit-identical to 6).
 * A tile a layer is not eligible for, or any other id, returns MPX_E_ARG; tile < 0 = the layer's default.  (Ids 3, 5, 8 and 11 of
 * earlier rounds -- kernels that were measured and never became a default -- exist only in probe builds,
 * tools/probes/build_experimental.sh.)  A non-default tile on a layer of a block tail makes mpx_forward run that block layer by
 * layer (mpx_bottleneck_tail). */
int mpx_set_conv_tile(mpx_engine* h, int i, int tile);
int mpx_get_conv_tile(const mpx_engine* h, int i);
/* Test hook: which kernels the LAST mpx_conv_bn_act call launched, as a bit mask over the tile ids above (bit t = the kernel of tile
 * t ran).  A layer's tile is a request: a launch under one round of tiles of a persistent kernel (10, 12, 13; 14: under two rounds) runs on the small-tile
 * kernel that sums in the same order (7, 6, 2; 14: 7), a residual operand sends 13 to 9, and the 256x256 kernels hand the images behind the
 * last whole round to tile 2 -- so a test that means to cover a persistent walk asserts that it ran. */
int mpx_last_conv_kernels(const mpx_engine* h);

/* Host-only packer (no GPU needed; what mpx_set_conv_weights runs before the upload).
 * Produces the fp16 planes w_hi/w_lo of cout_pad x k_packed elements (k order = (ky,kx,ci), ci fastest;
 * for the 7x7 stem k = ky*32 + px*4 + c over the NHWC4 padded input) in PIECE-major order: element (row, k) of a plane is at
 *   ((((row/16) * (k_packed/32) + k/32) * 16 + row%16) * 4 + ((k/8)%4 ^ (((row%16)/8) * 2))) * 8 + k%8
 * -- [cout_pad/16][k_packed/32][16 rows][four 16-byte chunks, XOR-swizzled by the row], so that each 1-KiB LDS-DMA piece of
 * the conv kernels is one contiguous run of 8 cache lines already in the order of its LDS image (cout_pad % 16 == 0,
 * k_packed % 32 == 0).  Each output channel is
 * multiplied by 2^e so that max|w| lies in [512,1024), and the fp32 epilogue
 * scale = gamma/sqrt(var+eps) * 2^-e, shift = beta + (conv_bias - mean)*gamma/sqrt(var+eps).  k_packed = k*k*cin_pad
 * with cin_pad >= cin the channels per pixel of the input planes (padding channels get zero weights).
 * All outputs are HOST buffers sized from mpx_conv_desc (uint16_t = raw fp16 bits). */
int mpx_pack_conv_weights(const mpx_conv_desc* d, const float* w, const float* conv_bias, const float* gamma,
                          const float* beta, const float* mean, const float* var, float eps,
                          uint16_t* w_hi, uint16_t* w_lo, float* scale, float* shift);
/* ---- K0: mask-apply + normalise ----------------------------------------------------------
 * replaces: transforms.ToTensor + Normalize (generate_gp_training_data_imagenet.py:598-599),
 *           the per-segment pixel-mask build (:234-237), `input[0].numpy().copy() * mask`
 *           (:240) and the per-mask H2D copy (:242-245).
 * Exactly one of img_u8_hwc (DEV u8[224][224][3], raw pixels; normalised in-kernel as
 * (u8/255 - mean_c)/std_c in fp32) and img_f32_chw (DEV f32[3][224][224], already normalised,
 * what the reference's val_loader yields) is non-NULL.
 * seg: DEV i32[224][224], labels in [0,S) (rank in np.unique(segments) order).
 * onoff: DEV u8[M][S]; onoff[m][s] != 0 keeps superpixel s in mask m (normalise THEN mask:
 * removed pixels become 0.0 in normalised space).
 * Writes masked image m into engine input slot slot0+m (slot0+M <= max_batch) and, if
 * out_f32_nchw (DEV f32[M][3][224][224]) is non-NULL, the reference-layout tensor as well. */
int mpx_mask_apply_normalize(mpx_engine* h, const uint8_t* img_u8_hwc, const float* img_f32_chw,
                             const int32_t* seg, const uint8_t* onoff, int M, int S,
                             const float mean[3], const float std[3], int slot0,
                             float* out_f32_nchw, void* stream);

/* ---- K0 + stem + max pool for the masks of ONE image, by superposition ---------------------------------------
 * replaces: the same lines as mpx_mask_apply_normalize (generate_gp_training_data_imagenet.py:598-599,234-245) AND the first
 *           `x = self.conv1(x); x = self.bn1(x); x = self.relu(x); x = self.maxpool(x)` of model(masked_img_tensor) (:246) for
 *           every mask of one image.  conv1 is linear and a mask is a union of superpixels, so conv1(x * mask_m) at an output pixel is
 *           the sum, over the superpixels its 7x7 window touches, of onoff[m][s] * (the window's taps inside s) -- terms that do not
 *           depend on the mask.  mpx_stem_table_build computes them once per image (one fp32 conv of the normalised image, taps
 *           bucketed by label; the table lives in the engine and holds one image at a time); mpx_stem_table_apply then writes
 *           relu(bn1(sum of the kept terms)) max-pooled 3x3 / 2 for M mask rows into the engine's pooled stem planes, slots
 *           [slot0, slot0 + M) -- no masked image is ever materialised.  Same arguments as mpx_mask_apply_normalize (img: exactly
 *           one of u8 HWC / f32 CHW; seg ranks in [0, S); onoff[m][s] != 0 keeps superpixel s), S <= 4096.  fp32 FMA chains in tap
 *           order instead of the MFMA stem's split-fp16 products: equal up to rounding (~1e-7 relative).
 * mpx_forward runs the B slots from the pooled planes when ALL of them were staged this way since they were last staged by
 * mpx_mask_apply_normalize, from the input staging when none was, and fails (MPX_E_STATE) on a mixed batch.
 * The table's device memory (160 MB, independent of max_batch) is allocated by the FIRST mpx_stem_table_build of an engine -- the one
 * allocation behind this boundary after mpx_create (an engine that only stages through mpx_mask_apply_normalize never holds it);
 * mpx_workspace_bytes includes it from then on, mpx_destroy frees it. */
int mpx_stem_table_build(mpx_engine* h, const uint8_t* img_u8_hwc, const float* img_f32_chw, const int32_t* seg, int S,
                         const float mean[3], const float std[3], void* stream);
int mpx_stem_table_apply(mpx_engine* h, const uint8_t* onoff, int M, int S, int slot0, void* stream);

/* ---- K0 of the small networks: the CIFAR / MNIST scorers' mask convention ---------------------------
 * replaces: the in-place min-max rescale of the picture to [0,255] (generate_gp_training_data_cifar.py:274-279,
 *           generate_gp_training_data_mnist.py:167-171), `mask.fill(255); mask[segments == segVal] = 0` for the SELECTED
 *           superpixels (:310-313 / :213-217), `masked_img = org_img * mask`, the second in-place min-max rescale and
 *           normalize_image = * f32(1/255) (:315-321 / :220-242, utils.py:92-94), and the per-mask H2D copy.
 * img_f32_chw: DEV f32[C][H][W] as the loader yields it; seg: DEV i32[H][W] ranks in [0,S), S <= 4096;
 * removed: DEV u8[M][S], removed[m][s] != 0 switches superpixel s OFF in mask m.  Writes input slots [slot0, slot0+M) and,
 * if out_f32_nchw (DEV f32[M][C][H][W]) is non-NULL, the network input in the reference's layout (bit-exact against the
 * NumPy arithmetic; a mask that removes every pixel gives NaN, as 0/0 does upstream). */
int mpx_mask_apply_minmax(mpx_engine* h, const float* img_f32_chw, const int32_t* seg, const uint8_t* removed,
                          int M, int S, int slot0, float* out_f32_nchw, void* stream);

/* ---- DownsampleB (models/resnet.py:64-74): AvgPool2d(2) on the identity + zero channels; planes
 * [B][hin][hin][cin_p] -> [B][hin/2][hin/2][cout_p] (channel counts as stored: multiples of 8, cout_p >= cin_p). */
int mpx_avgpool2_pad(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int B,
                     int hin, int cin_p, int cout_p, void* stream);

/* ---- K1/K2: conv + BN (+ residual) (+ ReLU), one layer ------------------------------------
 * replaces: one nn.Conv2d -> nn.BatchNorm2d (-> `out += identity`) (-> nn.ReLU) group inside
 *           model(masked_img_tensor) (generate_gp_training_data_imagenet.py:246).
 * in_hi|lo, res_hi|lo, out_hi|lo: DEV fp16 NHWC planes [B][h][w][c] (layer 0 reads the engine's padded
 * NHWC4 input staging instead: pass in_hi = in_lo = NULL).  res_* may be NULL.
 * For the last entry ("fc") out_hi/out_lo are ignored and out_f32 (DEV f32[B][1000]) is written;
 * for every other layer out_f32 must be NULL. */
int mpx_conv_bn_act(mpx_engine* h, int i, const void* in_hi, const void* in_lo,
                    const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                    float* out_f32, int B, void* stream);

/* ---- K1 with the downsample branch fused in ---------------------------------------------------
 * replaces: `out = self.bn3(self.conv3(out)); identity = self.downsample(x); out += identity; out = self.relu(out)`
 *           of the first Bottleneck of a stage (torchvision resnet.py, reached through model(masked_img_tensor),
 *           generate_gp_training_data_imagenet.py:246) as ONE launch: the 1x1 conv3 and the 1x1 (strided) downsample
 *           conv are K-concatenated, with the two BatchNorm scales folded into the weight rows as ratios <= 1
 *           (s = max(|s3|,|sd|): out = relu(s*(W3*s3/s . t2 + Wd*sd/s . x) + shift3 + shiftd)), so the downsample
 *           output never goes to HBM.  i = index of the block's conv3 ("layerN.0.conv3"); in_* = its input planes
 *           [B][h][w][cin3], x_* = the block input planes [B][H][W][cin_ds] (H = h * stride).  mpx_forward uses this
 *           path by default; mpx_set_fusion(h, 0) makes it run the two convs separately (bit-different, same
 *           tolerance; for tests and ablation).  Fused planes are built once both layers have weights. */
int mpx_conv_dual_bn_act(mpx_engine* h, int i, const void* in_hi, const void* in_lo, const void* x_hi,
                         const void* x_lo, void* out_hi, void* out_lo, int B, void* stream);
/* mask bit 0: the downsample fusion above and the stem + max-pool fusion below; bit 1: the block tails of layer1
 * (mpx_bottleneck_tail; needs bit 0 as well).  mpx_create starts with 3; 0 = one launch per layer. */
int mpx_set_fusion(mpx_engine* h, int mask);

/* ---- the tail of a 64-channel bottleneck block in ONE launch ------------------------------------------------
 * replaces: `out = self.relu(self.bn2(self.conv2(out))); out = self.bn3(self.conv3(out)); out += identity; out = self.relu(out)`
 *           of a layer1 Bottleneck AND `out = self.relu(self.bn1(self.conv1(x)))` of the block that follows it (torchvision
 *           resnet.py, reached through model(masked_img_tensor), generate_gp_training_data_imagenet.py:246).  layer1 works on
 *           56x56 maps and is HBM-bound; layer by layer a block moves 16 units (64 channels x 4 B per pixel) through HBM, this
 *           launch 10.4: conv2's output never leaves the registers (its accumulators are conv3's MFMA operand), and the
 *           256-channel trunk is read once -- as the identity -- instead of twice, because the next block's conv1 runs on the
 *           output tile while it is still on chip.  Only the 64-channel conv2 input needs a one-pixel halo (everything after
 *           the 3x3 conv is pointwise).
 * i = index of the block's conv2 ("layer1.N.conv2"); mpx_num_bottleneck_tails / mpx_bottleneck_tail_info list the blocks
 * that have this path (ResNet-50/101/152: the three blocks of layer1; the last one produces layer2.0.conv1's output).
 * t1_*: conv2's input planes [B][56][56][64]; x_*: the block's identity planes [B][56][56][256] -- for a block with a
 * downsample branch (layer1.0) the BLOCK INPUT planes [B][56][56][64], the branch being K-concatenated as in
 * mpx_conv_dual_bn_act; out_*: block output [B][56][56][256]; next_*: the following conv1's output [B][56][56][64 or 128].
 * For the block with the downsample branch t1_hi = t1_lo = NULL makes the launch compute t1 itself: the block's own
 * `out = self.relu(self.bn1(self.conv1(x)))` (64 -> 64, 1x1) runs on the halo tile of the block input it has staged anyway, so
 * layer1.0 is ONE launch with one read of its 64-channel input and neither t1 nor t2 ever in memory (what mpx_forward does).
 * All four plane pairs must be distinct buffers (workgroups read t1's halo while others write).  Same arithmetic as the
 * layer-by-layer path up to fp32 summation order.  mpx_forward takes this path by default. */
int mpx_bottleneck_tail(mpx_engine* h, int i, const void* t1_hi, const void* t1_lo, const void* x_hi, const void* x_lo,
                        void* out_hi, void* out_lo, void* next_hi, void* next_lo, int B, void* stream);
int mpx_num_bottleneck_tails(const mpx_engine* h);
/* layer indices of tail k: its conv2, conv3, downsample conv (-1 if none) and the following block's conv1; any pointer may be NULL */
int mpx_bottleneck_tail_info(const mpx_engine* h, int k, int* conv2, int* conv3, int* downsample, int* next_conv1);

/* ---- K3: maxpool 3x3 s2 p1 (nn.MaxPool2d inside the same forward), NHWC split planes ------ */
int mpx_maxpool3x3s2(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi,
                     void* out_lo, int B, int hin, int c, void* stream);

/* ---- K1 + K3 in one launch: the ImageNet stem and its max pool -------------------------------------------
 * replaces: `x = self.conv1(x); x = self.bn1(x); x = self.relu(x); x = self.maxpool(x)` (torchvision resnet.py, reached through
 *           model(masked_img_tensor), generate_gp_training_data_imagenet.py:246): the 7x7 stride-2 conv + BN + ReLU of layer 0 reads
 *           the engine's own staged input (mpx_mask_apply_normalize) like mpx_conv_bn_act(h, 0, ...) and writes the POOLED planes
 *           [B][56][56][64]; a workgroup computes the 15 x 17 conv outputs under a 7 x 8 block of pooled pixels, so the 112 x 112 conv
 *           output (3.2 MB per image) is never written nor re-read.  Bit-identical to mpx_conv_bn_act + mpx_maxpool3x3s2.
 *           mpx_forward uses it by default (ImageNet ResNets, stem on its default tile); mpx_set_fusion(h, 0) turns it off together
 *           with the downsample fusion. */
int mpx_stem_conv_maxpool(mpx_engine* h, void* out_hi, void* out_lo, int B, void* stream);

/* ---- K4a: global average pool [B][hw][c] -> [B][c] (nn.AvgPool2d(7) + view) ---------------- */
int mpx_global_avgpool(mpx_engine* h, const void* in_hi, const void* in_lo, void* out_hi,
                       void* out_lo, int B, int hw, int c, void* stream);

/* ---- K4b: softmax + gather(label) + argmax ------------------------------------------------
 * replaces: F.softmax(mask_output) ... [0][label] (bayesian_active_learning_imagenet.py:196-198)
 *           and mask_output.data.max(1, keepdim=True)[1] (generate_gp_training_data_imagenet.py:248).
 * logits DEV f32[B][1000]; label DEV i32[B]; score DEV f32[B]; pred DEV i32[B]. */
int mpx_head_softmax_gather(mpx_engine* h, const float* logits, const int32_t* label,
                            float* score, int32_t* pred, int B, void* stream);

/* ---- whole network -------------------------------------------------------------------------
 * replaces: mask_output = model(masked_img_tensor) + score extraction for B masked images
 *           already staged in input slots [0,B) by mpx_mask_apply_normalize.
 * logits_out (DEV f32[B][1000]) may be NULL. */
int mpx_forward(mpx_engine* h, const int32_t* label, float* score, int32_t* pred,
                float* logits_out, int B, void* stream);

/* ---- K5: heat-map accumulation ---------------------------------------------
 * replaces: summed_superpixel_labels[segments == v] += 1 for every selected superpixel of a correctly
 *           predicted mask (gp_superpixel_data_imagenet.py:322-323) and the pure-Python read-back loops of
 *           gp_regression.py:82-94:  heat[p] += sum_m [pred[m] == label[m]] * onoff[m][seg[p]].
 * seg DEV i32[224][224] ranks, onoff DEV u8[M][S], pred/label DEV i32[M], heat DEV f32[224][224] (accumulated
 * in place: zero it first; with several GPUs all-reduce it afterwards).  Uses a small engine-owned scratch
 * (S <= 4096). */
int mpx_heatmap_accumulate(mpx_engine* h, const int32_t* seg, const uint8_t* onoff, const int32_t* pred,
                           const int32_t* label, int M, int S, float* heat, void* stream);

/* ---- introspection for tests / benchmarks --------------------------------------------------- */
/* DEV pointers of the input staging planes: fp16 [max_batch][230][230][4] (padded NHWC4) for the ImageNet ResNets, [max_batch][H][W][32] for
 * the small networks.  A pure getter: the record of how each slot was staged is not touched, so a diagnostic call between
 * mpx_stem_table_apply and mpx_forward changes nothing. */
int mpx_input_planes(const mpx_engine* h, void** hi, void** lo);
/* A caller that has WRITTEN the input planes of slots [slot0, slot0+M) by hand declares them staged, as mpx_mask_apply_normalize does for the
 * slots it fills: the next mpx_forward runs the stem conv + max pool on them even if an earlier batch of the same slots came from
 * mpx_stem_table_apply (which marks the slots it writes as its own again).  MPX_E_ARG outside [0, max_batch). */
int mpx_mark_input_staged(mpx_engine* h, int slot0, int M);
/* DEV pointers of the pooled stem output planes: fp16 [max_batch][56][56][64], written by the stem + max pool launch of mpx_forward or by
 * mpx_stem_table_apply (NULL for the small networks, which have no such stem). */
int mpx_stem_planes(const mpx_engine* h, void** hi, void** lo);
/* When enabled, every kernel launch of mpx_forward / mpx_mask_apply_normalize is bracketed by
 * HIP events on the launch stream (bounded pool; launches beyond it are not recorded). */
int mpx_profile_enable(mpx_engine* h, int on);
/* Synchronises the recorded events, ADDS their durations into the caller's arrays and clears
 * the pool.  kind: 0 = conv (K1/K2), 1 = mask_apply_normalize (K0), 2 = pools (K3/K4a),
 * 3 = head (K4b).  per_conv_ms (HOST f64[mpx_num_convs], may be NULL) gets the per-layer split. */
int mpx_profile_collect(mpx_engine* h, double ms_by_kind[4], long long launches_by_kind[4],
                        double* per_conv_ms);
/* Algorithmic FLOPs (2*MAC, convs + fc) of one masked forward. */
double mpx_flops_per_forward(const mpx_engine* h);

#ifdef __cplusplus
}
#endif
#endif /* MPX_H */
