"""Regenerates tests/golden/segments_slic.npz: SLIC label maps of the seeded blob pictures (BASELINE configs[4] says "SLIC superpixel
masks (generate_superpixels.py)").  The reference imports slic next to felzenszwalb (generate_superpixels.py:2) and only ever calls
felzenszwalb (:10,15), so felzenszwalb is what the scorers run -- but a caller may hand the engine any integer label map, and a SLIC
map is the one the config names: labels as scikit-image returns them, which (start_label=1, the library's announced default) do NOT
start at 0.

Two interpreters, as tests/golden/make_segments.py (system python: torch, no skimage; conda python3.9: skimage 0.18.3, no torch):
    python tests/golden/make_slic_segments.py stage1                     # writes /tmp/mpx_imgshow.npy (img_show of the two pictures)
    /opt/conda/bin/python3.9 tests/golden/make_slic_segments.py stage2   # skimage.segmentation.slic -> /tmp/mpx_slic.npy
    python tests/golden/make_slic_segments.py stage3                     # packs the fixture
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
    import warnings
    from skimage.segmentation import slic
    from skimage.util import img_as_float
    import skimage
    shows = np.load("/tmp/mpx_imgshow.npy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # picture 0: the call as a 2018-era script would write it (start_label left at the library's default of that version);
        # picture 1: start_label=1, the default scikit-image announces for later versions -- labels 1 .. S
        segs = np.stack([slic(img_as_float(shows[0]), n_segments=100, compactness=10),
                         slic(img_as_float(shows[1]), n_segments=100, compactness=10, start_label=1)])
    np.save("/tmp/mpx_slic.npy", segs)
    print("stage2 skimage", skimage.__version__, [(int(s.min()), int(s.max()), len(np.unique(s))) for s in segs])
else:
    segs = np.load("/tmp/mpx_slic.npy")
    assert segs.max() < 32767
    np.savez_compressed(os.path.join(HERE, "segments_slic.npz"), segments=segs.astype(np.int16), image_seed=np.int64(1234),
                        skimage_version="0.18.3", call="slic(img_as_float(img_show), n_segments=100, compactness=10[, start_label=1 for picture 1])")
    print("stage3", segs.shape, [(int(s.min()), int(s.max()), len(np.unique(s))) for s in segs])
