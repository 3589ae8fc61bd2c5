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
                 "kfusion::cuda::resizeDepthNormals", "kfusion::cuda::ProjectiveICP::estimateTransform"):
        assert name in syms, name


@pytest.mark.gpu
def test_host_combined_solver_runs_reference_opttests(exes):
    out = _run(exes["test_host_solver"])
    assert "8 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_tsdf_volume_matches_oracle(exes):
    out = _run(exes["test_host_tsdf"])
    assert "4 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_dynfusion_sequence(exes):
    out = _run(exes["test_host_dynfusion"])
    assert "7 tests, 0 failed" in out


@pytest.mark.gpu
def test_host_projective_icp(exes):
    out = _run(exes["test_host_icp"])
    assert "2 tests, 0 failed" in out
