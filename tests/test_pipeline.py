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


def test_parallel_loader_delivers_the_sequential_batches_in_order():
    """pipeline.ParallelLoader (train.py:85-97 `DataLoader(num_workers=..)`): samples read by worker threads finishing
    out of order, batches delivered in index order, last partial batch kept / dropped, a permutation honoured."""
    import random
    import time

    class Reader:
        def __len__(self):
            return 23

        def __getitem__(self, i):
            time.sleep(random.random() * 0.004)
            return {"imgs_u8": torch.full((3, 2, 2, 3), i, dtype=torch.uint8), "proj_mats": torch.zeros(2, 3, 3, 4),
                    "init_depth_min": torch.tensor([float(i)]), "depth_interval": torch.tensor([2.5]), "scan_vid": ("s", i)}
    got = list(P.ParallelLoader(Reader(), batch_size=4, num_workers=8))
    assert len(got) == 6 and [b["imgs_u8"].shape[0] for b in got] == [4, 4, 4, 4, 4, 3]
    assert [v for b in got for _, v in b["scan_vid"]] == list(range(23))
    assert got[2]["init_depth_min"].shape == (4, 1) and got[2]["init_depth_min"][:, 0].tolist() == [8.0, 9.0, 10.0, 11.0]
    perm = [5, 1, 22, 7, 0]
    got = list(P.ParallelLoader(Reader(), batch_size=2, num_workers=3, indices=perm, drop_last=True))
    assert [v for b in got for _, v in b["scan_vid"]] == perm[:4] and len(P.ParallelLoader(Reader(), 2, 3, perm, drop_last=True)) == 2


def _write_dtu_tree(root, test_layout, g, lights=None):
    """A two-camera-pair DTU-format tree with random images / depths (file names and text formats of the real dataset)."""
    from PIL import Image
    os = __import__("os")
    cam_dir = root / ("Cameras" if test_layout else "Cameras/train")
    cam_dir.mkdir(parents=True, exist_ok=True)
    (root / "Cameras").mkdir(exist_ok=True)
    (root / "Cameras" / "pair.txt").write_text("3\n0\n2 1 0.9 2 0.8\n1\n2 0 0.9 2 0.7\n2\n2 1 0.6 0 0.5\n")
    for vid in range(3):
        K = np.array([[2892.33 if test_layout else 361.54, 0, 823.2 if test_layout else 82.9], [0, 2883.18 if test_layout else 360.4, 619.07 if test_layout else 66.4], [0, 0, 1]])
        E = np.eye(4)
        E[:3, 3] = [10.0 * vid, -5.0 * vid, 2.0]
        lines = ["extrinsic"] + [" ".join(f"{v:.6f}" for v in row) for row in E] + ["", "intrinsic"] + \
                [" ".join(f"{v:.6f}" for v in row) for row in K] + ["", f"{425.0 + vid} 2.5"]
        (cam_dir / f"{vid:08d}_cam.txt").write_text("\n".join(lines) + "\n")
    scan = "scan9"
    img_dir = root / "Rectified" / (scan if test_layout else scan + "_train")
    img_dir.mkdir(parents=True)
    hw = (1200, 1600) if test_layout else (512, 640)
    imgs = {}
    for vid in range(3):
        for light in (lights if lights is not None else ([3] if test_layout else range(7))):
            a = g.integers(0, 256, hw + (3,), dtype=np.uint8)
            Image.fromarray(a).save(img_dir / f"rect_{vid + 1:03d}_{light}_r5000.png")
            imgs[(vid, light)] = a
    depths = {}
    if not test_layout:
        (root / "Depths" / scan).mkdir(parents=True)
        for vid in range(3):
            d = (500.0 + 100.0 * g.random((1200, 1600))).astype(np.float32)
            P.save_pfm(str(root / "Depths" / scan / f"depth_map_{vid:04d}.pfm"), d)
            m = (g.random((1200, 1600)) > 0.3).astype(np.uint8) * 255
            Image.fromarray(m).save(root / "Depths" / scan / f"depth_visual_{vid:04d}.png")
            depths[vid] = (d, m)
    return scan, imgs, depths


def test_dtu_reader_training_layout(tmp_path):
    g = np.random.default_rng(0)
    scan, imgs, depths = _write_dtu_tree(tmp_path, False, g)
    r = P.DTUReader(str(tmp_path), [scan], n_views=3, n_cameras=3)
    assert len(r) == 3 * 7                                     # 3 reference views x 7 light conditions (dtu.py:37-49)
    s = r[8]                                                   # reference view 1, light 1
    assert s["scan_vid"] == (scan, 1) and s["imgs_u8"].shape == (3, 512, 640, 3) and s["imgs_u8"].dtype == torch.uint8
    assert np.array_equal(s["imgs_u8"][0].numpy(), imgs[(1, 1)]) and np.array_equal(s["imgs_u8"][1].numpy(), imgs[(0, 1)])
    assert float(s["init_depth_min"]) == 426.0 and abs(float(s["depth_interval"]) - 2.65) < 1e-6
    assert s["proj_mats"].shape == (2, 3, 3, 4)
    want = P.relative_proj_mats(r.proj_mats[1][0], [r.proj_mats[0][0], r.proj_mats[2][0]])
    assert torch.equal(s["proj_mats"], want)
    # ground truth: half-size nearest-neighbour map, (44:556, 80:720) crop, then two more halvings (dtu.py:93-131)
    d, m = depths[1]
    d0 = d[::2, ::2][44:556, 80:720]
    assert np.array_equal(s["depths"]["level_0"].numpy(), d0) and np.array_equal(s["depths"]["level_2"].numpy(), d0[::4, ::4])
    assert np.array_equal(s["masks"]["level_1"].numpy(), (m[::2, ::2][44:556, 80:720] > 0)[::2, ::2])
    b = P.collate([r[0], r[8]])
    assert b["imgs_u8"].shape == (2, 3, 512, 640, 3) and b["init_depth_min"].shape == (2, 1)


def test_dtu_reader_test_layout_resizes_images_and_intrinsics(tmp_path):
    from PIL import Image
    g = np.random.default_rng(1)
    scan, imgs, _ = _write_dtu_tree(tmp_path, True, g)
    r = P.DTUReader(str(tmp_path), [scan], n_views=2, img_wh=(160, 128), n_cameras=3)
    assert len(r) == 3                                         # light condition 3 only
    s = r[0]
    assert s["imgs_u8"].shape == (2, 128, 160, 3) and "depths" not in s
    want = np.asarray(Image.fromarray(imgs[(0, 3)]).resize((160, 128), Image.BILINEAR))
    assert np.array_equal(s["imgs_u8"][0].numpy(), want)
    # intrinsics scaled to the coarsest level of the resized image: fx * W / 1600 / 4 (dtu.py:61-63)
    K, E, _ = P.read_cam_file(str(tmp_path / "Cameras" / "00000000_cam.txt"))
    K[0] *= 160 / 1600 / 4
    K[1] *= 128 / 1200 / 4
    assert torch.allclose(r.proj_mats[0][0][2, :3], torch.tensor(K @ E[:3]), rtol=1e-5)
    with pytest.raises(ValueError):
        P.DTUReader(str(tmp_path), [scan], img_wh=(100, 128))


def test_resize_nearest_matches_cv2_index_rule():
    a = np.arange(7 * 10).reshape(7, 10)
    assert np.array_equal(P.resize_nearest(a, fx=0.5, fy=0.5), a[[0, 2, 4, 6]][:, [0, 2, 4, 6, 8]])      # round(3.5) = 4 rows
    assert np.array_equal(P.resize_nearest(a, out_hw=(3, 4)), a[[0, 2, 4]][:, [0, 2, 5, 7]])


@pytest.mark.gpu
def test_files_to_depth_maps_through_reader_prefetcher_and_engine(tmp_path):
    """eval.py:213-222 end to end on files: DTU-format tree -> DTUReader (PIL decode, camera files) -> collate -> DevicePrefetcher
    (uint8 upload, device normalisation, side stream) -> CascadeMVSNet.forward; equals the forward on hand-normalised inputs."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    dev = torch.device("cuda:0")
    g = np.random.default_rng(2)
    scan, _, _ = _write_dtu_tree(tmp_path, True, g)
    r = P.DTUReader(str(tmp_path), [scan], n_views=3, img_wh=(160, 128), n_cameras=3)
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=1)
    model = model.to(dev).eval()
    samples = [r[i] for i in range(len(r))]
    batches = [P.collate([s]) for s in samples]
    n = 0
    for i, b in enumerate(P.DevicePrefetcher(batches, dev, depth=2)):
        out = model(b["imgs"], b["proj_mats"], b["init_depth_min"], b["depth_interval"])
        u8 = samples[i]["imgs_u8"].unsqueeze(0)
        mean, std = torch.tensor(P.IMAGENET_MEAN).view(1, 1, 3, 1, 1), torch.tensor(P.IMAGENET_STD).view(1, 1, 3, 1, 1)
        imgs = u8.permute(0, 1, 4, 2, 3).float().div(255).sub(mean).div(std).to(dev)
        want = model(imgs, samples[i]["proj_mats"].unsqueeze(0).to(dev), samples[i]["init_depth_min"].view(1, 1).to(dev),
                     samples[i]["depth_interval"].view(1, 1).to(dev))
        assert b["scan_vid"] == [(scan, i)] and out["depth_0"].shape == (1, 128, 160)
        assert torch.equal(out["depth_0"], want["depth_0"]) and torch.isfinite(out["depth_0"]).all()
        n += 1
    assert n == 3


@pytest.mark.gpu
def test_files_to_point_cloud(tmp_path):
    """eval.py end to end on a (tiny, synthetic) DTU-format tree: images + cameras -> depth maps -> fused point cloud -> PLY."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.reconstruct import reconstruct_scan
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    g = np.random.default_rng(4)
    scan, _, _ = _write_dtu_tree(tmp_path, True, g)
    reader = P.DTUReader(str(tmp_path), [scan], n_views=3, img_wh=(160, 128), n_cameras=3)
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=2)
    model = model.to("cuda").eval()
    ply = tmp_path / "scan9.ply"
    # random weights give no geometrically consistent depth: accept every pixel so that the path to the file is exercised
    pts, cols = reconstruct_scan(model, reader, scan, out_ply=str(ply), conf=-1.0, min_geo_consistent=0, skip=4)
    assert pts.shape[1] == 3 and pts.shape == cols.shape and len(pts) == 3 * (128 * 160 // 4) and torch.isfinite(pts).all()
    head = ply.read_bytes().split(b"end_header\n")[0]
    assert (b"element vertex %d\n" % len(pts)) in head


@pytest.mark.gpu
@pytest.mark.order_tier(2)   # asserts a trend (the loss after four steps is below the first): behind every parity test (tests/conftest.py)
def test_training_loop_on_files(tmp_path):
    """train.py's loop on a DTU-format tree (training layout): DTUReader -> collate -> prefetcher -> train-mode forward ->
    SL1 loss on the ground-truth depth / mask pyramids -> backward -> SGD; the loss falls."""
    from casmvsnet_pl_amd import CascadeMVSNet, InPlaceABN
    from casmvsnet_pl_amd import training as T
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    g = np.random.default_rng(5)
    scan, _, _ = _write_dtu_tree(tmp_path, False, g, lights=[0])
    reader = P.DTUReader(str(tmp_path), [scan], n_views=3, n_cameras=3)
    idx = [i for i, m in enumerate(reader.metas) if m[1] == 0]          # the light condition that was written
    assert len(idx) == 3
    model = CascadeMVSNet(norm_act=InPlaceABN)
    randomize_state_dict(model.state_dict(), seed=3)
    model = model.to("cuda")
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    sample = reader[idx[0]]
    assert sample["depths"]["level_0"].shape == (512, 640) and sample["masks"]["level_2"].shape == (128, 160)
    batches = [P.collate([sample])] * 4                                 # the same batch four times
    losses = T.train_steps(model, batches, opt)
    assert len(losses) == 4 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_blendedmvs_reader_scales_the_scene_and_derives_the_interval(tmp_path):
    """blendedmvs.py: per-scan scale 100 / depth_min(first camera) on depth range, translations and depth maps; reference
    views with too few valid views skipped; depth_interval = (depth_max - depth_min) / 192 per sample (BASELINE config 5)."""
    from PIL import Image
    g = np.random.default_rng(6)
    root = tmp_path / "dataset_low_res"
    scan = "5a3ca9cb270f0e3f14d0eddb"
    (root / scan / "cams").mkdir(parents=True)
    (root / scan / "blended_images").mkdir()
    (root / scan / "rendered_depth_maps").mkdir()
    (root / scan / "cams" / "pair.txt").write_text("4\n0\n3 1 0.9 2 0.8 3 0.7\n1\n3 0 0.9 2 0.7 3 0.6\n2\n1 0 0.5\n3\n3 0 0.9 1 0.8 2 0.7\n")   # view 2: one valid view only
    dmins = [4.0, 5.0, 6.0, 7.0]
    depth0 = None
    for vid in range(4):
        K = np.array([[570.0, 0, 384.0], [0, 570.0, 288.0], [0, 0, 1]])
        E = np.eye(4)
        E[:3, 3] = [0.1 * vid, 0.2, 0.3]
        lines = ["extrinsic"] + [" ".join(f"{v:.6f}" for v in row) for row in E] + ["", "intrinsic"] + \
                [" ".join(f"{v:.6f}" for v in row) for row in K] + ["", f"{dmins[vid]} 0.05 128 10.4"]
        (root / scan / "cams" / f"{vid:08d}_cam.txt").write_text("\n".join(lines) + "\n")
        Image.fromarray(g.integers(0, 256, (576, 768, 3), dtype=np.uint8)).save(root / scan / "blended_images" / f"{vid:08d}.jpg")
        d = (3.0 + 6.0 * g.random((576, 768))).astype(np.float32)
        P.save_pfm(str(root / scan / "rendered_depth_maps" / f"{vid:08d}.pfm"), d)
        if vid == 0:
            depth0 = d
    r = P.BlendedMVSReader(str(root), [scan], n_views=3, img_wh=(384, 288))
    assert len(r) == 3 and [m[2] for m in r.metas] == [0, 1, 3]               # view 2 skipped (1 < n_views valid views)
    sf = 100 / 4.0
    assert r.scale_factors[scan] == sf
    s = r[0]
    assert s["imgs_u8"].shape == (3, 288, 384, 3) and s["proj_mats"].shape == (2, 3, 3, 4)
    assert abs(float(s["init_depth_min"]) - 100.0) < 1e-4
    d0 = P.resize_nearest(depth0 * np.float32(sf), out_hw=(288, 384))
    assert np.allclose(s["depths"]["level_0"].numpy(), d0, rtol=1e-6)
    assert abs(float(s["depth_interval"]) - (float(d0.max()) - 100.0) / 192.0) < 1e-4
    assert np.array_equal(s["masks"]["level_1"].numpy(), s["depths"]["level_1"].numpy() > 100.0)
    assert abs(r.proj_mats[scan][1][1] - 5.0 * sf) < 1e-4                      # every camera's range uses the SCAN's factor
    # translation scaled, intrinsics to the coarsest level of the resized image (768 -> 384: x 0.5 / 4)
    Pm = r.proj_mats[scan][0][0][2]
    assert abs(float(Pm[0, 0]) - 570.0 * 384 / 768 / 4) < 1e-3 and abs(float(Pm[2, 3]) - 0.3 * sf) < 1e-4


def test_tanks_reader(tmp_path):
    from PIL import Image
    g = np.random.default_rng(8)
    scan = "Lighthouse"
    base = tmp_path / "intermediate" / scan
    (base / "cams").mkdir(parents=True)
    (base / "images").mkdir()
    (base / "pair.txt").write_text("3\n0\n2 1 0.9 2 0.8\n1\n2 0 0.9 2 0.7\n2\n2 1 0.6 0 0.5\n")
    for vid in range(3):
        K = np.array([[1165.0, 0, 1024.0], [0, 1165.0, 540.0], [0, 0, 1]])
        E = np.eye(4)
        E[:3, 3] = [0.05 * vid, 0.0, 0.1]
        lines = ["extrinsic"] + [" ".join(f"{v:.6f}" for v in row) for row in E] + ["", "intrinsic"] + \
                [" ".join(f"{v:.6f}" for v in row) for row in K] + ["", f"{0.4 + 0.1 * vid} 0.002"]
        (base / "cams" / f"{vid:08d}_cam.txt").write_text("\n".join(lines) + "\n")
        Image.fromarray(g.integers(0, 256, (270, 512, 3), dtype=np.uint8)).save(base / "images" / f"{vid:08d}.jpg")
    r = P.TanksReader(str(tmp_path), "intermediate", scans=[scan], n_views=3, img_wh=(256, 128))
    assert len(r) == 3
    s = r[1]
    assert s["scan_vid"] == (scan, 1) and s["imgs_u8"].shape == (3, 128, 256, 3) and s["proj_mats"].shape == (2, 3, 3, 4)
    assert abs(float(s["init_depth_min"]) - 0.5) < 1e-6 and abs(float(s["depth_interval"]) - 1.5e-2) < 1e-8   # the scan's hand-tuned interval
    assert abs(float(r.proj_mats[scan][0][0][2][0, 0]) - 1165.0 * 256 / 2048 / 4) < 1e-3                     # native width 2048 for this scan
