"""Image preparation on the GPU (csrc/preprocess.hip through the C ABI) against the fixtures the reference's own
evaluate_one_img / Datasets.__getitem__ produced, and against the oracle on seeded inputs.  Integer resampling and a table
look-up: every comparison is bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def _cases():
    z = np.load(os.path.join(GOLD, "preprocess_cases.npz"))
    names = sorted({k.rsplit("/", 1)[0] for k in z.files})
    return [(n, z[n + "/src"], tuple(int(v) for v in z[n + "/input_shape"]), z[n + "/out"]) for n in names]


def test_preprocess_matches_reference_fixtures():
    from ccnet_amd.utils import preprocess as pp
    for name, src, shape, exp in _cases():
        out, info = pp.preprocess([src], shape, DEV, letterbox=name.startswith("letterbox/"))
        assert out.shape == (1, 3) + shape and out.dtype == torch.float32
        assert info == [[0, src.shape[1], src.shape[0]]]
        assert np.array_equal(out[0].cpu().numpy(), exp), name


@pytest.mark.parametrize("letterbox", [False, True])
def test_preprocess_random_sizes_match_oracle(letterbox):
    """Ragged batch: sizes repeat out of order (grouping + scatter), widths that are and are not multiples of 4 (vector and
    scalar store paths), a grey image, a 1-pixel image."""
    from ccnet_amd.utils import preprocess as pp
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(5)
    for shape in [(64, 96), (63, 95), (224, 224)]:
        sizes = [(200, 200), (37, 91), (200, 200), (300, 150), (37, 91), (1, 1), (200, 200), (64, 96)]
        imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
        imgs.append(rng.integers(0, 256, (45, 33), dtype=np.uint8))
        out, info = pp.preprocess(imgs, shape, DEV, letterbox=letterbox, names=["f%d" % i for i in range(len(imgs))])
        got = out.cpu().numpy()
        for i, im in enumerate(imgs):
            rgb = im if im.ndim == 3 else np.repeat(im[..., None], 3, 2)
            exp = (po.preprocess_letterbox if letterbox else po.preprocess_eval)(rgb, shape)
            assert np.array_equal(got[i], exp), (shape, i, im.shape)
            assert info[i] == ["f%d" % i, im.shape[1], im.shape[0]]


def test_preprocess_full_size_batch_and_downscale():
    """640x640 from NEU-DET-sized sources (the BASELINE workload's input) and a large downscale (many taps per pixel)."""
    from ccnet_amd.utils import preprocess as pp
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(6)
    src = rng.integers(0, 256, (16, 200, 200, 3), dtype=np.uint8)
    out, _ = pp.preprocess(torch.from_numpy(src), (640, 640), DEV)
    got = out.cpu().numpy()
    for i in (0, 7, 15):
        assert np.array_equal(got[i], po.preprocess_eval(src[i], (640, 640)))
    big = rng.integers(0, 256, (2, 1080, 1920, 3), dtype=np.uint8)
    out, _ = pp.preprocess(torch.from_numpy(big), (640, 640), DEV)
    assert np.array_equal(out[1].cpu().numpy(), po.preprocess_eval(big[1], (640, 640)))
    # a reduction too strong for the LDS-staged horizontal pass (707 taps per pixel): the direct kernel takes over
    wide = rng.integers(0, 256, (1, 40, 3000, 3), dtype=np.uint8)
    out, _ = pp.preprocess(torch.from_numpy(wide), (16, 17), DEV)
    assert np.array_equal(out[0].cpu().numpy(), po.preprocess_eval(wide[0], (16, 17)))
    # constant image: every weight row sums to 2^22 up to rounding, so the value must survive both passes
    flat = np.full((1, 123, 77, 3), 201, np.uint8)
    out, _ = pp.preprocess(torch.from_numpy(flat), (640, 640), DEV, letterbox=True)
    lut = pp.normalisation_lut()
    vals = torch.unique(out[0, 0]).cpu().numpy()
    assert set(vals.tolist()) <= {float(lut[0, 201]), float(lut[0, 128])}


def test_preprocess_rejects_bad_arguments():
    from ccnet_amd.utils import preprocess as pp
    with pytest.raises(ValueError):
        pp.preprocess([np.zeros((4, 4, 4), np.uint8)], (32, 32), DEV)
    with pytest.raises(ValueError):
        pp.preprocess([np.zeros((4, 4, 3), np.float32)], (32, 32), DEV)
    src = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=DEV)
    with pytest.raises(ValueError):
        pp.preprocess_group(src, torch.empty((1, 3, 16, 17), device=DEV), (16, 16))
    with pytest.raises(TypeError):
        pp.preprocess_group(src.float(), torch.empty((1, 3, 16, 16), device=DEV), (16, 16))
