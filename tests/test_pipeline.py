"""Input pipeline: host logic on CPU (camera files, projection matrices, PFM round trip, collate) and the
device side on the GPU (uint8 normalisation kernel, prefetcher feeding the engine)."""
import numpy as np
import pytest
import torch

from casmvsnet_pl_amd import pipeline as P
from casmvsnet_pl_amd.synthetic import dtu_like_cameras, make_inputs


def test_cam_file_and_proj_mats_follow_dtu_py(tmp_path):
    K = np.array([[361.54, 0, 82.9], [0, 360.4, 66.4], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32)
    E[:3, 3] = [10.0, -20.0, 30.0]
    lines = ["extrinsic"] + [" ".join(f"{v:.6f}" for v in row) for row in E] + ["", "intrinsic"] + \
            [" ".join(f"{v:.6f}" for v in row) for row in K] + ["", "425.0 2.5"]
    f = tmp_path / "00000000_cam.txt"
    f.write_text("\n".join(lines) + "\n")
    k, e, dmin = P.read_cam_file(str(f))
    assert np.allclose(k, K) and np.allclose(e, E) and dmin == 425.0
    mats = P.build_proj_mats(k, e, levels=3)                          # fine -> coarse, like dtu.py:66-74
    assert mats.shape == (3, 4, 4)
    assert torch.allclose(mats[2, :3], torch.tensor(K @ E[:3]), atol=1e-4)           # coarsest = the file's intrinsics
    assert torch.allclose(mats[0, :2], mats[2, :2] * 4, rtol=1e-5) and torch.allclose(mats[0, 2], mats[2, 2])
    # the relative matrices equal what the synthetic rig (the tests' / bench's input generator) produces
    cams = dtu_like_cameras(3, 64, 96)
    rel = P.relative_proj_mats(cams[0], cams[1:])
    _, proj, _, _ = make_inputs(1, 3, 64, 96, seed=0)
    assert rel.shape == (2, 3, 3, 4) and torch.equal(rel, proj[0])


def test_pfm_round_trip(tmp_path):
    g = np.random.default_rng(0)
    for shape in ((7, 5), (4, 6, 3), (3, 2, 1)):
        a = g.random(shape).astype(np.float32)
        P.save_pfm(str(tmp_path / "a.pfm"), a)
        b, scale = P.read_pfm(str(tmp_path / "a.pfm"))
        assert scale == 1.0 and np.array_equal(b, a.reshape(b.shape))
    with pytest.raises(ValueError):
        P.save_pfm(str(tmp_path / "b.pfm"), np.zeros((2, 2)))         # float64 is refused like the reference does


def test_collate_builds_b1_depth_ranges():
    samples = [dict(imgs=torch.zeros(3, 3, 8, 8), proj_mats=torch.zeros(2, 3, 3, 4), init_depth_min=torch.tensor([425.0 + i]),
                    depth_interval=torch.tensor([2.65]), scan_vid=("scan1", i)) for i in range(3)]
    b = P.collate(samples)
    assert b["imgs"].shape == (3, 3, 3, 8, 8) and b["init_depth_min"].shape == (3, 1) and b["depth_interval"].shape == (3, 1)
    assert b["init_depth_min"][2, 0] == 427.0 and b["scan_vid"][1] == ("scan1", 1)


@pytest.mark.gpu
def test_normalize_kernel_equals_totensor_normalize():
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (2, 3, 40, 56, 3), generator=g, dtype=torch.uint8)
    want = u8.permute(0, 1, 4, 2, 3).float().div(255)                 # T.ToTensor
    mean, std = torch.tensor(P.IMAGENET_MEAN).view(1, 1, 3, 1, 1), torch.tensor(P.IMAGENET_STD).view(1, 1, 3, 1, 1)
    want = want.sub(mean).div(std)                                     # T.Normalize
    got = P.normalize_images_u8(u8.cuda()).cpu()
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.gpu
def test_prefetcher_feeds_the_engine_in_order():
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    dev = torch.device("cuda:0")
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=1)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    batches, direct = [], []
    for i in range(5):
        _, proj, dmin, dint = make_inputs(2, 3, 64, 96, seed=i)
        u8 = torch.randint(0, 256, (2, 3, 64, 96, 3), generator=g, dtype=torch.uint8)
        samples = [dict(imgs_u8=u8[b], proj_mats=proj[b], init_depth_min=torch.tensor([dmin + b]), depth_interval=torch.tensor([dint]), idx=i) for b in range(2)]
        batches.append(P.collate(samples))
        imgs = P.normalize_images_u8(u8.to(dev))
        out = model(imgs, proj.to(dev), batches[-1]["init_depth_min"].to(dev), batches[-1]["depth_interval"].to(dev))
        direct.append(out["depth_0"].clone())
    n = 0
    for i, b in enumerate(P.DevicePrefetcher(batches, dev, depth=2)):
        assert b["idx"] == [i, i] and b["imgs"].is_cuda and b["imgs"].shape == (2, 3, 3, 64, 96)
        out = model(b["imgs"], b["proj_mats"], b["init_depth_min"], b["depth_interval"])
        assert torch.equal(out["depth_0"], direct[i])
        n += 1
    assert n == 5
