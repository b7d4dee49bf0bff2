"""CPU study (test infrastructure, uses the oracle): which MFMA operand format keeps the class
probability within 1e-4 of the fp32 reference through 18 / 101 layers?  Every conv/linear input and
weight is rounded to the format under test, arithmetic runs in fp64.

    python oracle/precision_study.py resnet18 ; python oracle/precision_study.py resnet101

Result in this container (8 masked copies of one blob image, seeded synthetic weights):
    format            resnet18 score err   resnet101 score err
    fp32 oracle       9.4e-08              3.6e-07        (the reference's own rounding noise)
    f16   (1 pass)    1.3e-04              2.6e-04        FAILS 1e-4
    bf16  (1 pass)    2.4e-03              2.2e-03        FAILS 1e-4
    bf16x2 (3 pass)   1.0e-06              3.6e-06
    f16x2  (3 pass)   9.8e-08              3.3e-07        <- chosen: hi*hi + hi*lo + lo*hi on the fp16 MFMA pipe

Round 2, `python oracle/precision_study.py <arch> tensors`: weights hi+lo everywhere, activations per tensor class
(a conv whose input has no `lo` plane would need two MFMA products instead of three):
    resnet101, 8 masks of the blob image:  t1 hi only 2.3e-05 | t2 hi only 2.3e-05 | t1 and t2 hi only 2.2e-05 | trunk hi only 9.0e-05
    resnet18:                              t1 hi only 1.2e-05 | trunk hi only 2.6e-05
  over more pictures (3 blob + 2 noise images x 8 masks, same script logic): t1 and t2 hi only reaches 4.6e-05 (blobs) and
  8.1e-05 (uniform-noise images) on ResNet-101; t1 alone 1.3e-05 .. 3.0e-05, t2 alone up to 5.4e-05 -- inside the 1e-4
  tolerance but outside the 2e-5 the parity tests hold the engine to, so NOT adopted: every activation keeps its `lo` plane.
"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, torch.nn.functional as F
from network_interpretation_imagenet_amd import synth
from oracle import resnet_ref as R, scorer as S
arch = sys.argv[1]
sd = synth.make_state_dict(arch)
img = synth.make_images(1)[0]
x = S.to_tensor_normalize(img)
seg = synth.grid_segments()
onoff = synth.random_onoff(8, 196)
xb = torch.from_numpy(np.stack([S.apply_mask(x, S.onoff_mask_u8(seg, onoff[i])) for i in range(8)]))
sd64 = R.cast_state_dict(sd, torch.float64)
def rnd(t, mode):
    if mode=='f64': return t
    if mode=='f16': return t.to(torch.float16).double()
    if mode=='bf16': return t.to(torch.bfloat16).double()
    if mode=='f16x2':
        hi = t.to(torch.float16).double(); lo = (t-hi).to(torch.float16).double(); return hi+lo
    if mode=='bf16x2':
        hi = t.to(torch.bfloat16).double(); lo = (t-hi).to(torch.bfloat16).double(); return hi+lo
orig = F.conv2d
def run(mode):
    def conv(x, w, b, s, p): return orig(rnd(x,mode), rnd(w,mode), b, s, p)
    F.conv2d = conv
    lin = F.linear
    F.linear = lambda x,w,b: lin(rnd(x,mode), rnd(w,mode), b)
    with torch.no_grad(): lg = R.forward(sd64, xb.double(), arch)
    F.conv2d = orig; F.linear = lin
    return lg
ref = run('f64'); label = int(ref[0].argmax()) 
with torch.no_grad(): l32 = R.forward(sd, xb, arch).double()
pr = torch.softmax(ref,1)[:,label]
print('label', label, 'scores', pr.numpy().round(4))
print('fp32 oracle vs f64: logit %.2e score %.2e'%((l32-ref).abs().max(), (torch.softmax(l32,1)[:,label]-pr).abs().max()))
for m in ['f16','bf16','f16x2','bf16x2']:
    l = run(m)
    print(m, 'logit err %.2e score err %.2e'%((l-ref).abs().max(), (torch.softmax(l,1)[:,label]-pr).abs().max()))

# ---- round 2: per-tensor activation formats (weights always hi+lo) ----------------------------------------------------------
# Which activations need the `lo` plane?  The block input / output ("trunk") is what every later block adds to; the tensors
# inside a block (t1 = conv1's output, t2 = conv2's output) feed exactly one conv.  A conv whose input has no `lo` plane needs two
# MFMA products instead of three and half the pixel-operand bytes.
#     python oracle/precision_study.py resnet101 tensors
def run_tensors(policy):
    """policy: dict conv-name-suffix -> activation format of that conv's INPUT ('f16x2' default)."""
    real = R._conv_bn
    def conv_bn(sd_, x_, name, stride, pad, relu):
        suffix = name.split('.')[-1] if '.' in name else name
        if name.endswith('downsample.0'):
            suffix = 'downsample'
        fmt = policy.get(suffix, 'f16x2')
        w = sd_[name + '.weight']
        sd_local = dict(sd_)
        sd_local[name + '.weight'] = rnd(w, 'f16x2')
        return real(sd_local, rnd(x_, fmt), name, stride, pad, relu)
    R._conv_bn = conv_bn
    lin = F.linear
    F.linear = lambda x_, w_, b_: lin(rnd(x_, 'f16x2'), rnd(w_, 'f16x2'), b_)
    try:
        with torch.no_grad():
            lg = R.forward(sd64, xb.double(), arch)
    finally:
        R._conv_bn = real
        F.linear = lin
    return lg

if len(sys.argv) > 2 and sys.argv[2] == 'tensors':
    kind = R.ARCHS[arch][0]
    inner = {'bottleneck': [('t1 (conv2 input) hi only', {'conv2': 'f16'}), ('t2 (conv3 input) hi only', {'conv3': 'f16'}),
                            ('t1 and t2 hi only', {'conv2': 'f16', 'conv3': 'f16'}),
                            ('trunk hi only (conv1 / downsample / stem inputs)', {'conv1': 'f16', 'downsample': 'f16'})],
             'basic': [('t1 (conv2 input) hi only', {'conv2': 'f16'}),
                       ('trunk hi only (conv1 / downsample inputs)', {'conv1': 'f16', 'downsample': 'f16'})]}[kind]
    base = run_tensors({})
    print('per-tensor formats, %s (weights hi+lo everywhere); all activations hi+lo: logit err %.2e score err %.2e' % (
        arch, (base - ref).abs().max(), (torch.softmax(base, 1)[:, label] - pr).abs().max()))
    for title, pol in inner:
        l = run_tensors(pol)
        print('  %-52s logit err %.2e  score err %.2e' % (title, (l - ref).abs().max(), (torch.softmax(l, 1)[:, label] - pr).abs().max()))
