"""Regenerates tests/golden/segments_blobs.npz (felzenszwalb label maps of the seeded blob images).

Two interpreters are needed in this image: the system python has torch but no skimage, the conda
python3.9 has skimage 0.18.3 but no torch.  Run:
    python tests/golden/make_segments.py stage1          # system python: writes /tmp/mpx_imgshow.npy
    /opt/conda/bin/python3.9 tests/golden/make_segments.py stage2   # skimage: /tmp/mpx_segments.npy
    python tests/golden/make_segments.py stage3          # packs the fixture

The segmentation call is the reference's own
(felzenszwalb(img_as_float(img_show), scale=100, sigma=0.5, min_size=50),
generate_gp_training_data_imagenet.py:183) applied to img_show, the min-max-rescaled u8 picture
(:171-178) of the normalised tensor.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
N_IMAGES = 2

if sys.argv[1] == "stage1":
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from network_interpretation_imagenet_amd import synth, api
    from oracle import scorer
    imgs = synth.make_images(N_IMAGES, seed=1234, kind="blobs")
    shows = np.stack([api.img_show_u8(scorer.to_tensor_normalize(im).numpy()) for im in imgs])
    np.save("/tmp/mpx_imgshow.npy", shows)
    print("stage1", shows.shape, shows.dtype)
elif sys.argv[1] == "stage2":
    from skimage.segmentation import felzenszwalb
    from skimage.util import img_as_float
    import skimage
    shows = np.load("/tmp/mpx_imgshow.npy")
    segs = np.stack([felzenszwalb(img_as_float(s), scale=100, sigma=0.5, min_size=50) for s in shows])
    np.save("/tmp/mpx_segments.npy", segs)
    print("stage2 skimage", skimage.__version__, [len(np.unique(s)) for s in segs])
else:
    segs = np.load("/tmp/mpx_segments.npy")
    assert segs.max() < 32767
    np.savez_compressed(os.path.join(HERE, "segments_blobs.npz"), segments=segs.astype(np.int16),
                        image_seed=np.int64(1234), skimage_version="0.18.3")
    print("stage3", segs.shape, [len(np.unique(s)) for s in segs])
