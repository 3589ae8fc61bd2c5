"""The C++ host adaptors (dynfu_amd/host: kfusion::cuda::TsdfVolume, DualQuaternion, Node, Warpfield,
CombinedSolver with the reference's interfaces) driven by C++ test programs that mirror the
reference's gtest files (tests/cpp/*.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exes():
    from dynfu_amd import build as B
    import oracle
    oracle.build()
    return B.build_cpp_tests()


def _run(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_host_dual_quaternion_known_answers(exes):
    out = _run(exes["test_host_dq"])
    assert "23 tests, 0 failed" in out


def test_tsdf_run_classification_model_matches_oracle(exes):
    """CPU model of the run-classified TSDF sweep (csrc/tsdf_classify.hpp) vs oracle/tsdf_oracle.c, bit for bit."""
    out = _run(exes["test_tsdf_classify"])
    assert "6 tests, 0 failed" in out


def test_host_io_formats(exes):
    """PNG depth codec, PCD / VTK writers, sequence listing (dfa_host/io.hpp; reference: src/apps/demo.cpp)."""
    out = _run(exes["test_host_io"])
    assert "5 tests, 0 failed" in out


def _decode_with_ours(exe, png, tmp_path):
    import numpy as np
    raw = str(tmp_path / "decoded.bin")
    r = subprocess.run([exe, "decode", str(png), raw], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = open(raw, "rb").read()
    cols, rows = np.frombuffer(buf[:8], np.int32)
    return np.frombuffer(buf[8:], np.uint16).reshape(rows, cols)


def test_depth_png_decoder_against_pillow_and_the_golden_frame(exes, tmp_path):
    """Files written by an encoder that is not ours (Pillow: adaptive filter types, several deflate levels) decode to
    the same pixels; a file written by ours reads back in Pillow; the committed golden frame
    (tests/golden/make_depth_png.py) is the synthetic depth frame it was made from."""
    import numpy as np
    from PIL import Image
    from dynfu_amd import synth
    exe = exes["test_host_io"]
    rng = np.random.default_rng(5)
    smooth = (np.add.outer(np.arange(97), np.arange(131)) * 37 % 65536).astype(np.uint16)  # filters 1-4 pay off here
    for k, img in enumerate([rng.integers(0, 65536, (37, 53)).astype(np.uint16), smooth, np.zeros((1, 1), np.uint16),
                             np.full((480, 640), 65535, np.uint16)]):
        for level in (1, 9):
            png = tmp_path / ("pil_%d_%d.png" % (k, level))
            Image.fromarray(img).save(png, compress_level=level)
            assert np.array_equal(_decode_with_ours(exe, png, tmp_path), img)
    ours = tmp_path / "ours.png"
    r = subprocess.run([exe, "encode", "83", "61", str(ours)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    back = np.asarray(Image.open(ours))
    assert back.dtype == np.uint16 and back.shape == (61, 83)
    assert np.array_equal(back, _decode_with_ours(exe, ours, tmp_path))
    golden = os.path.join(ROOT, "tests", "golden", "depth_t0_frame7.png")
    assert np.array_equal(_decode_with_ours(exe, golden, tmp_path), synth.depth_frame(synth.CONFIGS["T0"], 7, noise_mm=1.0))
    eight = tmp_path / "eight.png"
    Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16)).save(eight)
    assert np.array_equal(_decode_with_ours(exe, eight, tmp_path), np.arange(256).reshape(16, 16))  # 8-bit grey: widened
    rgb = tmp_path / "rgb.png"
    Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(rgb)
    r = subprocess.run([exe, "decode", str(rgb), str(tmp_path / "x.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "greyscale" in (r.stdout + r.stderr)


def test_host_library_exports_the_adaptor_classes(exes):
    lib = os.path.join(ROOT, "dynfu_amd", "libdynfu_amd_host.so")
    syms = subprocess.run(["nm", "-DC", lib], capture_output=True, text=True).stdout
    for name in ("kfusion::cuda::TsdfVolume::integrate", "kfusion::cuda::TsdfVolume::raycast",
                 "kfusion::cuda::TsdfVolume::clear", "kfusion::cuda::computeDists", "Warpfield::calcDQB",
                 "Warpfield::warpToLive", "Warpfield::findNeighborsIndex", "CombinedSolver::initializeProblemInstance",
                 "CombinedSolver::solveAll", "DynFusion::warpCanonicalToLiveOpt", "DynFusion::findCorrespondingFrame",
                 "DynFusion::init", "DynFusion::fuse", "DynFusion::operator()", "Warpfield::update",
                 "Warpfield::getUnsupportedVertices", "dfa::voxelGridFilter", "kfusion::cuda::MarchingCubes::run",
                 "kfusion::cuda::depthBilateralFilter", "kfusion::cuda::computePointNormals",
                 "kfusion::cuda::resizeDepthNormals", "kfusion::cuda::ProjectiveICP::estimateTransform",
                 "dfa::io::readDepthPng", "dfa::io::savePCDFileASCII", "dfa::io::saveVTKFile", "dfa::io::listSequence",
                 "dfa::convertToMesh", "runSequence", "kfusion::KinFu::operator()", "kfusion::KinFu::reset"):
        assert name in syms, name


@pytest.mark.gpu
def test_host_combined_solver_runs_reference_opttests(exes):
    out = _run(exes["test_host_solver"])
    assert "8 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_tsdf_volume_matches_oracle(exes):
    out = _run(exes["test_host_tsdf"])
    assert "5 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_dynfusion_sequence(exes):
    out = _run(exes["test_host_dynfusion"])
    assert "12 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_kinfu_rigid_pipeline(exes):
    out = _run(exes["test_host_kinfu"])
    assert "3 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_projective_icp(exes):
    out = _run(exes["test_host_icp"])
    assert "2 tests, 0 failed" in out


@pytest.mark.gpu
def test_png_sequence_through_dynfusion_to_pcd_and_vtk(exes, tmp_path):
    """The reference demo's loop (src/apps/demo.cpp:68-124) on a synthetic sequence: depth PNGs in, DynFusion::operator()
    per frame, out/pcl_canonical_to_live<i>.pcd and out/<i>_tsdf_mesh.vtk for every frame after the first."""
    import numpy as np
    from PIL import Image
    from dynfu_amd import synth
    cfg = synth.CONFIGS["T1"]
    (tmp_path / "depth").mkdir()
    (tmp_path / "color").mkdir()
    n = 4
    for f in range(n):
        Image.fromarray(synth.depth_frame(cfg, f)).save(tmp_path / "depth" / ("frame-%06d.depth.png" % f))
        Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(tmp_path / "color" / ("frame-%06d.color.png" % f))
    r = subprocess.run([exes["test_host_io"], "sequence", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = dict(zip(r.stdout.split()[0::2], r.stdout.split()[1::2]))
    assert int(rep["frames"]) == n and int(rep["saved"]) == n - 1 and int(rep["nodes"]) > 10
    out = tmp_path / "out"
    assert sorted(p.name for p in out.iterdir()) == sorted(["pcl_canonical_to_live%d.pcd" % i for i in range(1, n)] +
                                                           ["%d_tsdf_mesh.vtk" % i for i in range(1, n)])
    pcd = (out / "pcl_canonical_to_live3.pcd").read_text().splitlines()
    assert pcd[0].startswith("# .PCD v0.7") and pcd[2] == "FIELDS x y z" and pcd[10] == "DATA ascii"
    npts = int(pcd[9].split()[1])
    assert npts == int(rep["canonical_vertices"]) == len(pcd) - 11 and npts > 1000
    pts = np.array([[float(v) for v in line.split()] for line in pcd[11:]])
    assert np.isfinite(pts).all() and 0.5 < pts[:, 2].mean() < 3.5  # inside the 3 m volume in front of the camera
    vtk = (out / "3_tsdf_mesh.vtk").read_text().splitlines()
    assert vtk[:4] == ["# vtk DataFile Version 3.0", "vtk output", "ASCII", "DATASET POLYDATA"]
    nv = int(vtk[4].split()[1])
    at = 5 + nv
    assert vtk[at] == "" and vtk[at + 1] == "VERTICES %d %d" % (nv, 2 * nv)
    at += 2 + nv
    npoly = int(vtk[at + 1].split()[1])
    assert vtk[at + 1] == "POLYGONS %d %d" % (npoly, 4 * npoly) and npoly == nv // 3 == int(rep["mesh_polygons"])
    tri = np.array([[int(v) for v in line.split()] for line in vtk[at + 2:at + 2 + npoly]])
    assert (tri[:, 0] == 3).all() and tri[:, 1:].max() == nv - 1
    assert np.array_equal(tri[:3, 1:], [[0, 2, 1], [3, 5, 4], [6, 8, 7]])  # KinFu::convertToMesh's winding


@pytest.mark.gpu
def test_png_sequence_through_dynfusion_in_north_star_mode(exes, tmp_path):
    """The same loop with DynFuParams::north_star: every frame after the first is solved against its depth map (6-DoF
    node transforms, NorthStarSolver); the written clouds are the canonical surface warped into the camera frame."""
    import numpy as np
    from PIL import Image
    from dynfu_amd import synth
    cfg = synth.CONFIGS["T1"]
    (tmp_path / "depth").mkdir()
    (tmp_path / "color").mkdir()
    n = 3
    for f in range(n):
        Image.fromarray(synth.depth_frame(cfg, f)).save(tmp_path / "depth" / ("frame-%06d.depth.png" % f))
        Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(tmp_path / "color" / ("frame-%06d.color.png" % f))
    env = dict(os.environ, DFA_SEQ_NORTHSTAR="1", DFA_SEQ_DIM="128")
    r = subprocess.run([exes["test_host_io"], "sequence", str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = dict(zip(r.stdout.split()[0::2], r.stdout.split()[1::2]))
    assert int(rep["frames"]) == n and int(rep["saved"]) == n - 1 and int(rep["nodes"]) > 10
    pcd = (tmp_path / "out" / "pcl_canonical_to_live2.pcd").read_text().splitlines()
    pts = np.array([[float(v) for v in line.split()] for line in pcd[11:]])
    assert len(pts) == int(rep["canonical_vertices"]) > 1000 and np.isfinite(pts).all()
    # camera frame: the sphere of the synthetic scene sits around (0, 0, 1.5), the wall at z = 2.5
    assert abs(np.median(pts[:, 0])) < 0.3 and abs(np.median(pts[:, 1])) < 0.3 and 0.9 < pts[:, 2].min() < 1.2 and pts[:, 2].max() < 2.7
