"""The C++ host layer (rvpt_amd/host/): GPU-free self test against a recording fake of the C ABI, and — on a GPU —
the headless CLI rvpt_render (load_model -> add_material x2 -> initialize -> update/draw loop, as the reference's
main()) checked against the oracle on the scene, BVH and camera block the CLI itself dumps."""
import json
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host_bins():
    from rvpt_amd import build
    return build.build_host()


def test_host_selftest_runs_clean(host_bins, tmp_path):
    res = subprocess.run([str(host_bins / "host_selftest"), str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host_selftest ok" in res.stdout


def test_cli_reports_a_missing_model(host_bins, tmp_path):
    res = subprocess.run([str(host_bins / "rvpt_render"), "--obj", str(tmp_path / "nope.obj")], capture_output=True, text=True)
    assert res.returncode == 1 and "MODEL-LOADING" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("traversal,batch", [("bvh", 1), ("brute", 1), ("bvh_ordered", 2)])
def test_cli_renders_the_demo_scene_like_the_oracle(host_bins, oracle, tmp_path, traversal, batch):
    from rvpt_amd import imageio, scene
    obj = tmp_path / "model.obj"
    scene.write_obj(obj, scene.default_model_positions())
    W, H, frames, spp = 96, 48, 3, 2
    out, prefix = tmp_path / "frame.pfm", tmp_path / "dump"
    cmd = [str(host_bins / "rvpt_render"), "--obj", str(obj), "--width", str(W), "--height", str(H), "--spp", str(spp), "--frames", str(frames),
           "--traversal", traversal, "--translate", "0.2", "0.9", "-2.4", "--rotate", "-5", "4", "0", "--fov", "80", "--out", str(out),
           "--dump-prefix", str(prefix), "--batch", str(batch)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["triangles"] == 143 and info["last_frame"] == frames - 1
    cam = np.fromfile(f"{prefix}.camera.f32", dtype=np.float32)
    tris = np.fromfile(f"{prefix}.triangles.f32", dtype=np.float32).reshape(-1, 16)
    mats = np.fromfile(f"{prefix}.materials.f32", dtype=np.float32).reshape(-1, 12)
    nodes = np.fromfile(f"{prefix}.nodes.bin", dtype=np.uint32).reshape(-1, 8)
    assert tris.shape[0] == 143 and mats.shape[0] == 2 and (tris[:, 12] == 1).all()
    prev = None
    trav = {"bvh": oracle.TRAVERSAL_BVH, "brute": oracle.TRAVERSAL_BRUTE, "bvh_ordered": oracle.TRAVERSAL_BVH_ORDERED}[traversal]
    for f in range(frames):
        prev, _ = oracle.render(oracle.settings_bytes(aa=spp, current_frame=f), cam, nodes, tris, mats, W, H, trav, prev=prev)
    got = imageio.read_pfm(out)
    assert np.array_equal(got, prev[..., :3])


@pytest.mark.gpu
def test_cli_multi_gpu_path_through_the_rccl_group(host_bins, tmp_path):
    """rvpt_render --gpus N builds one RCCL group over its per-GPU contexts (rvpt_hip_comm_init_all) and reads the frame
    through the gather.  One GPU here: --force-collective runs the same code with a group of one; the image is unchanged."""
    from rvpt_amd import imageio, scene
    obj = tmp_path / "model.obj"
    scene.write_obj(obj, scene.default_model_positions())
    imgs = []
    for extra in ([], ["--force-collective"]):
        out = tmp_path / f"frame{len(imgs)}.pfm"
        cmd = [str(host_bins / "rvpt_render"), "--obj", str(obj), "--width", "80", "--height", "48", "--spp", "2", "--frames", "3", "--traversal", "brute",
               "--translate", "0.2", "0.9", "-2.4", "--out", str(out)] + extra
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
        info = json.loads(res.stdout.strip().splitlines()[-1])
        assert info["collective"] == bool(extra) and info["gpus"] == 1
        imgs.append(imageio.read_pfm(out))
    assert np.array_equal(imgs[0], imgs[1])


@pytest.mark.gpu
def test_cli_renders_an_obj_mtl_scene_like_the_oracle(host_bins, oracle, tmp_path):
    """--scene: OBJ + MTL through the C++ loader == the Python loader (same triangles, materials, ids) == the oracle's image."""
    from rvpt_amd import imageio, native, scene
    tris0, mats0 = scene.materials_showcase_scene()
    scene.write_obj_scene(tmp_path / "show.obj", tris0, mats0)
    W, H, frames, spp = 96, 64, 3, 2
    out, prefix = tmp_path / "frame.pfm", tmp_path / "dump"
    cmd = [str(host_bins / "rvpt_render"), "--scene", str(tmp_path / "show.obj"), "--width", str(W), "--height", str(H), "--spp", str(spp),
           "--frames", str(frames), "--batch", "3", "--traversal", "bvh", "--translate", "0.2", "1.0", "-2.3", "--out", str(out), "--dump-prefix", str(prefix)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    cam = np.fromfile(f"{prefix}.camera.f32", dtype=np.float32)
    tris = np.fromfile(f"{prefix}.triangles.f32", dtype=np.float32).reshape(-1, 16)
    mats = np.fromfile(f"{prefix}.materials.f32", dtype=np.float32).reshape(-1, 12)
    nodes = np.fromfile(f"{prefix}.nodes.bin", dtype=np.uint32).reshape(-1, 8)
    ptris, pmats, _ = scene.load_obj_scene(tmp_path / "show.obj")
    assert np.array_equal(mats, pmats)
    pnodes, order = native.build_bvh(ptris)
    assert np.array_equal(tris, ptris[order])
    prev = None
    for f in range(frames):
        prev, _ = oracle.render(oracle.settings_bytes(aa=spp, current_frame=f), cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BVH, prev=prev)
    assert np.array_equal(imageio.read_pfm(out), prev[..., :3])


@pytest.mark.gpu
def test_cli_two_gpus_equal_one(host_bins, tmp_path):
    """rvpt_render --gpus 2: the image tile-split over two devices and gathered through the library's RCCL group equals the
    single-GPU image bit for bit.  Needs two GPUs (skipped on the one-GPU test boxes)."""
    from rvpt_amd import imageio, native, scene
    if native.device_count() < 2:
        pytest.skip("needs two GPUs")
    obj = tmp_path / "model.obj"
    scene.write_obj(obj, scene.default_model_positions())
    imgs = []
    for gpus in (1, 2):
        out = tmp_path / f"frame{gpus}.pfm"
        cmd = [str(host_bins / "rvpt_render"), "--obj", str(obj), "--width", "208", "--height", "112", "--spp", "2", "--frames", "4", "--traversal", "bvh",
               "--translate", "0.2", "0.9", "-2.4", "--gpus", str(gpus), "--out", str(out)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
        imgs.append(imageio.read_pfm(out))
    assert np.array_equal(imgs[0], imgs[1])
