"""Harness input path (reference train_utils/kaist_train_eval_utils.py:54-71): uint8 -> float/255 -> multi-scale
bilinear resize.  CPU: oracle/inputs.py against the torch-generated fixture tests/golden/inputs.npz.  GPU: the
`dyk_image_prep` pass (dyk.functional.prepare_images / multi_scale_pair, and uint8 batches handed straight to
models.YOLO) against the oracle and the fixture."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

from oracle import inputs as oin

TOL = 5e-7           # float32 values in 0..1: a few roundings between fused and unfused blends (indices/lambdas exact)


def _gold():
    return np.load(os.path.join(GOLDEN, "inputs.npz"))


def test_oracle_plain_conversion_exact():
    g = _gold()
    assert np.array_equal(oin.prepare_images(g["u8"]), g["plain"])


def test_oracle_multi_scale_sizes_and_values():
    g = _gold()
    for img_size, h, w in g["sizes"]:
        ns = oin.multi_scale_size(g["u8"].shape[2:], int(img_size))
        assert (ns or list(g["u8"].shape[2:])) == [h, w]
        got = oin.prepare_images(g["u8"], ns)
        assert got.shape == g["ms_%d" % img_size].shape
        assert np.abs(got - g["ms_%d" % img_size]).max() <= TOL
    assert np.abs(oin.prepare_images(g["u8"], [37, 53]) - g["odd_37x53"]).max() <= TOL
    assert np.abs(oin.prepare_images(g["one_src"], [5, 3]) - g["one_5x3"]).max() <= TOL


@pytest.mark.gpu
def test_gpu_prepare_images_fixture():
    from dyk import functional as Fn
    g = _gold()
    u8 = torch.from_numpy(g["u8"]).cuda()
    assert np.array_equal(Fn.prepare_images(u8).cpu().numpy(), g["plain"])            # division by 255: exact
    for img_size, h, w in g["sizes"]:
        v, l = Fn.multi_scale_pair(u8, u8.flip(0), int(img_size))
        assert tuple(v.shape[2:]) == (h, w) and v.shape == l.shape
        assert np.abs(v.cpu().numpy() - g["ms_%d" % img_size]).max() <= TOL
        assert np.abs(v.cpu().numpy() - oin.prepare_images(g["u8"], [h, w])).max() <= TOL
    assert np.abs(Fn.prepare_images(u8, [37, 53]).cpu().numpy() - g["odd_37x53"]).max() <= TOL
    one = torch.from_numpy(g["one_src"]).cuda()
    assert np.abs(Fn.prepare_images(one, [5, 3]).cpu().numpy() - g["one_5x3"]).max() <= TOL


@pytest.mark.gpu
def test_gpu_prepare_images_baseline_size():
    """[16,3,512,640] uint8 -> 384x480 (img_size 480 of the multi-scale range) against the oracle, and the
    size-independent properties: a constant image stays constant, output within [0,1]."""
    from dyk import functional as Fn
    gen = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (16, 3, 512, 640), dtype=torch.uint8, generator=gen)
    ns = oin.multi_scale_size((512, 640), 480)
    assert ns == [384, 480]
    got = Fn.prepare_images(u8.cuda(), ns).cpu().numpy()
    ref = oin.prepare_images(u8[:2].numpy(), ns)
    assert np.abs(got[:2] - ref).max() <= TOL
    assert got.min() >= 0.0 and got.max() <= 1.0
    const = torch.full((1, 3, 512, 640), 77, dtype=torch.uint8).cuda()
    c = Fn.prepare_images(const, ns).cpu().numpy()
    assert np.abs(c - np.float32(77) / np.float32(255)).max() <= 1e-7
    with pytest.raises(TypeError):
        Fn.prepare_images(u8.cuda().half())


@pytest.mark.gpu
def test_model_accepts_uint8_batches():
    """models.YOLO(v_u8, l_u8) == models.YOLO(v_u8.float()/255, l_u8.float()/255) bit for bit"""
    from build_utils.parse_config import materialize_cfg
    from helpers import C3, oracle_net
    from models import YOLO
    torch.manual_seed(0)
    m = YOLO(materialize_cfg(C3))
    m.load_state_dict(oracle_net(C3).synth_state(0))
    m = m.cuda().eval()
    gen = torch.Generator().manual_seed(5)
    v = torch.randint(0, 256, (2, 3, 128, 160), dtype=torch.uint8, generator=gen).cuda()
    l = torch.randint(0, 256, (2, 3, 128, 160), dtype=torch.uint8, generator=gen).cuda()
    with torch.no_grad():
        a = m(v, l)[0]
        b = m(v.float() / 255.0, l.float() / 255.0)[0]
    assert torch.equal(a, b)
