"""The dataset readers of casmvsnet_pl_amd/pipeline.py against the UNMODIFIED reference classes
(/root/reference/datasets/{dtu,blendedmvs,tanks}.py, run through oracle/reference_loader.load_reference_datasets() behind
the cv2 / torchvision import shims; PIL does the decoding and the bilinear resize on both sides) on synthetic trees in the
datasets' on-disk formats: EVERY field of EVERY sample is compared.  The reference tree exists only in the build
container: on the GPU box these tests skip (tests/test_pipeline.py holds the hand-derived expectations that travel)."""
import os

import numpy as np
import pytest
import torch

from casmvsnet_pl_amd import pipeline as P
from oracle import reference_loader as RL

pytestmark = pytest.mark.skipif(not RL.reference_available(), reason="reference tree not on this machine")

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def _cam_text(K, E, last):
    return "\n".join(["extrinsic"] + [" ".join(f"{v:.6f}" for v in row) for row in E] + ["", "intrinsic"] +
                     [" ".join(f"{v:.6f}" for v in row) for row in K] + ["", last]) + "\n"


def _same_sample(ours, ref, what):
    """every key of the reference's sample dict (imgs: ToTensor + Normalize of our uint8 images, computed the way
    casmvs_normalize_images_u8 does on the device: u8 / 255, minus mean, over std - float32)"""
    assert set(ref) == (set(ours) - {"imgs_u8"}) | {"imgs"}, (what, sorted(ref), sorted(ours))
    x = ours["imgs_u8"].permute(0, 3, 1, 2).to(torch.float32).div(255)
    assert torch.equal((x - MEAN) / STD, ref["imgs"]), what
    assert ours["proj_mats"].dtype == ref["proj_mats"].dtype and ours["proj_mats"].shape == ref["proj_mats"].shape
    assert torch.equal(ours["proj_mats"], ref["proj_mats"]), (what, float((ours["proj_mats"] - ref["proj_mats"]).abs().max()))
    for k in ("init_depth_min", "depth_interval"):
        assert ours[k].dtype == ref[k].dtype and torch.equal(ours[k], ref[k]), (what, k, ours[k], ref[k])
    assert tuple(ours["scan_vid"]) == tuple(ref["scan_vid"])
    for k in ("depths", "masks"):
        if k in ref:
            assert sorted(ours[k]) == sorted(ref[k])
            for lv in ref[k]:
                assert ours[k][lv].dtype == ref[k][lv].dtype and torch.equal(ours[k][lv], ref[k][lv]), (what, k, lv)


def _write_dtu(root, test_layout, g, scan="scan9"):
    from PIL import Image
    cam_dir = root / ("Cameras" if test_layout else "Cameras/train")
    cam_dir.mkdir(parents=True, exist_ok=True)
    (root / "Cameras").mkdir(exist_ok=True)
    (root / "Cameras" / "pair.txt").write_text("3\n0\n3 1 0.9 2 0.8 5 0.1\n1\n2 0 0.9 2 0.7\n2\n2 1 0.6 0 0.5\n")
    for vid in range(49):                                     # dtu.py:53 reads all 49 camera files
        K = np.array([[2892.33 if test_layout else 361.54, 0, 823.2 if test_layout else 82.9],
                      [0, 2883.18 if test_layout else 360.4, 619.07 if test_layout else 66.4], [0, 0, 1]])
        E = np.eye(4)
        a = 0.02 * vid
        E[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        E[:3, 3] = [10.0 * vid, -5.0 * vid, 2.0 + vid]
        (cam_dir / f"{vid:08d}_cam.txt").write_text(_cam_text(K, E, f"{425.0 + vid} 2.5"))
    img_dir = root / "Rectified" / (scan if test_layout else scan + "_train")
    img_dir.mkdir(parents=True)
    hw = (1200, 1600) if test_layout else (512, 640)
    for vid in (0, 1, 2, 5):
        for light in ([3] if test_layout else range(7)):
            Image.fromarray(g.integers(0, 256, hw + (3,), dtype=np.uint8)).save(img_dir / f"rect_{vid + 1:03d}_{light}_r5000.png")
    if not test_layout:
        (root / "Depths" / scan).mkdir(parents=True)
        for vid in range(3):
            P.save_pfm(str(root / "Depths" / scan / f"depth_map_{vid:04d}.pfm"), (500.0 + 100.0 * g.random((1200, 1600))).astype(np.float32))
            Image.fromarray((g.random((1200, 1600)) > 0.3).astype(np.uint8) * 255).save(root / "Depths" / scan / f"depth_visual_{vid:04d}.png")
    return scan


@pytest.mark.parametrize("test_layout,n_views", [(False, 3), (False, 2), (True, 4)])
def test_dtu_reader_equals_the_reference_dataset(tmp_path, monkeypatch, test_layout, n_views):
    """dtu.py:10-192.  The reference reads its scan list from the cwd-relative datasets/lists/dtu/<split>.txt: the test
    provides one (data, not code) in a scratch working directory."""
    g = np.random.default_rng(5)
    root = tmp_path / "dtu"
    scan = _write_dtu(root, test_layout, g)
    (tmp_path / "cwd" / "datasets" / "lists" / "dtu").mkdir(parents=True)
    split = "test" if test_layout else "val"
    (tmp_path / "cwd" / "datasets" / "lists" / "dtu" / f"{split}.txt").write_text(scan + "\n")
    monkeypatch.chdir(tmp_path / "cwd")
    img_wh = (160, 128) if test_layout else None
    ref = RL.load_reference_datasets().DTUDataset(str(root) + "/", split, n_views=n_views, depth_interval=2.65, img_wh=img_wh)
    ours = P.DTUReader(str(root) + "/", [scan], n_views=n_views, depth_interval=2.65, img_wh=img_wh)
    assert len(ours) == len(ref) == (3 if test_layout else 21) and [tuple(m[:3]) + (tuple(m[3]),) for m in ours.metas] == [tuple(m[:3]) + (tuple(m[3]),) for m in ref.metas]
    for vid in range(49):
        assert torch.equal(ours.proj_mats[vid][0], ref.proj_mats[vid][0]) and ours.proj_mats[vid][1] == ref.proj_mats[vid][1]
    for i in range(len(ref)):
        _same_sample(ours[i], ref[i], f"dtu sample {i}")


def test_blendedmvs_reader_equals_the_reference_dataset(tmp_path):
    """blendedmvs.py:11-187, validation split (the training split's ColorJitter is random)."""
    from PIL import Image
    g = np.random.default_rng(6)
    root = tmp_path / "dataset_low_res"
    scans = ["5a3ca9cb270f0e3f14d0eddb", "5a6464143d809f1d8208c43c"]
    (tmp_path / "validation_list.txt").write_text("\n".join(scans) + "\n")
    for si, scan in enumerate(scans):
        for d in ("cams", "blended_images", "rendered_depth_maps"):
            (root / scan / d).mkdir(parents=True)
        (root / scan / "cams" / "pair.txt").write_text("4\n0\n3 1 0.9 2 0.8 3 0.7\n1\n3 0 0.9 2 0.7 3 0.6\n2\n1 0 0.5\n3\n3 0 0.9 1 0.8 2 0.7\n")
        for vid in range(4):
            K = np.array([[570.0 + si, 0, 384.0], [0, 571.0, 288.0], [0, 0, 1]])
            E = np.eye(4)
            E[:3, 3] = [0.1 * vid, 0.2 + si, 0.3]
            (root / scan / "cams" / f"{vid:08d}_cam.txt").write_text(_cam_text(K, E, f"{4.0 + vid + si} 0.05 128 10.4"))
            Image.fromarray(g.integers(0, 256, (576, 768, 3), dtype=np.uint8)).save(root / scan / "blended_images" / f"{vid:08d}.jpg")
            P.save_pfm(str(root / scan / "rendered_depth_maps" / f"{vid:08d}.pfm"), (3.0 + 6.0 * g.random((576, 768))).astype(np.float32))
    for img_wh in ((384, 288), (768, 576)):
        ref = RL.load_reference_datasets().BlendedMVSDataset(str(root), "val", n_views=3, depth_interval=192.0, img_wh=img_wh)
        ours = P.BlendedMVSReader(str(root), scans, n_views=3, n_coarse_intervals=192.0, img_wh=img_wh)
        assert len(ours) == len(ref) == 6 and ours.scale_factors == ref.scale_factors
        for i in range(len(ref)):
            _same_sample(ours[i], ref[i], f"blendedmvs {img_wh} sample {i}")


def test_tanks_reader_equals_the_reference_dataset(tmp_path):
    """tanks.py:11-163: the reference opens every scan of the split, so all eight get a (tiny) tree."""
    from PIL import Image
    g = np.random.default_rng(8)
    for scan, (w, h) in P.TanksReader.IMAGE_SIZES["intermediate"].items():
        base = tmp_path / "intermediate" / scan
        (base / "cams").mkdir(parents=True)
        (base / "images").mkdir()
        (base / "pair.txt").write_text("3\n0\n2 1 0.9 2 0.8\n1\n2 0 0.9 2 0.7\n2\n2 1 0.6 0 0.5\n")
        for vid in range(3):
            K = np.array([[1165.0, 0, w / 2], [0, 1166.0, 540.0], [0, 0, 1]])
            E = np.eye(4)
            E[:3, 3] = [0.05 * vid, 0.01 * len(scan), 0.1]
            (base / "cams" / f"{vid:08d}_cam.txt").write_text(_cam_text(K, E, f"{0.4 + 0.1 * vid} 0.002"))
            Image.fromarray(g.integers(0, 256, (h // 4, w // 4, 3), dtype=np.uint8)).save(base / "images" / f"{vid:08d}.jpg")
    ref = RL.load_reference_datasets().TanksDataset(str(tmp_path), "intermediate", n_views=3, img_wh=(256, 128))
    ours = P.TanksReader(str(tmp_path), "intermediate", n_views=3, img_wh=(256, 128))
    assert len(ours) == len(ref) == 24
    for i in range(len(ref)):
        _same_sample(ours[i], ref[i], f"tanks sample {i}")
