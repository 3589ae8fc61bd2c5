// projective_icp.cpp — kfusion::cuda::ProjectiveICP on the dynfu_amd C ABI
// (reference: src/kfusion/projective_icp.cpp:62-200).
#include <kfusion/cuda/projective_icp.hpp>

#include <cmath>

#include "../../../include/dynfu_amd.h"

namespace kfusion {
namespace cuda {

ProjectiveICP::ProjectiveICP() : gate_angle_(20.f * 0.017453293f), gate_dist_(0.1f) {  // :62
    setIterationsNum({10, 5, 4, 0});                                                      // :63-65
    sums_.create(27);
}
ProjectiveICP::~ProjectiveICP() {}

void ProjectiveICP::setIterationsNum(const std::vector<int>& iters) {  // :82-89
    schedule_.assign(MAX_PYRAMID_LEVELS, 0);
    for (size_t i = 0; i < iters.size() && i < (size_t)MAX_PYRAMID_LEVELS; ++i) schedule_[i] = iters[i];
}

int ProjectiveICP::getUsedLevelsNum() const {  // :91-96
    int i = MAX_PYRAMID_LEVELS - 1;
    for (; i >= 0 && !schedule_[i]; --i) {
    }
    return i + 1;
}

namespace {
// pivoted LU of a 6x6 system in double: determinant and solution (the reference: cv::determinant, then
// cv::solve(..., DECOMP_SVD) — for the non-singular systems that pass its determinant test the same solution)
bool solve6(const double A_in[36], const double b_in[6], double x[6], double& det) {
    double A[36], b[6];
    for (int i = 0; i < 36; ++i) A[i] = A_in[i];
    for (int i = 0; i < 6; ++i) b[i] = b_in[i];
    det = 1.0;
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (std::fabs(A[6 * r + c]) > std::fabs(A[6 * piv + c])) piv = r;
        if (A[6 * piv + c] == 0.0) {
            det = 0.0;
            return false;
        }
        if (piv != c) {
            for (int j = 0; j < 6; ++j) std::swap(A[6 * c + j], A[6 * piv + j]);
            std::swap(b[c], b[piv]);
            det = -det;
        }
        det *= A[6 * c + c];
        for (int r = c + 1; r < 6; ++r) {
            const double f = A[6 * r + c] / A[6 * c + c];
            for (int j = c; j < 6; ++j) A[6 * r + j] -= f * A[6 * c + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
        for (int j = r + 1; j < 6; ++j) s -= A[6 * r + j] * x[j];
        x[r] = s / A[6 * r + r];
    }
    return true;
}

// cv::Affine3f(rvec, t): Rodrigues rotation vector -> matrix
Affine3f from_rvec(const double r[3], const double t[3]) {
    Affine3f a;
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th > 2.220446049250313e-16) {
        const double c = std::cos(th), s = std::sin(th), c1 = 1.0 - c, k[3] = {r[0] / th, r[1] / th, r[2] / th};
        const double R[9] = {c + c1 * k[0] * k[0],        c1 * k[0] * k[1] - s * k[2], c1 * k[0] * k[2] + s * k[1],
                             c1 * k[0] * k[1] + s * k[2], c + c1 * k[1] * k[1],        c1 * k[1] * k[2] - s * k[0],
                             c1 * k[0] * k[2] - s * k[1], c1 * k[1] * k[2] + s * k[0], c + c1 * k[2] * k[2]};
        for (int i = 0; i < 9; ++i) a.R[i] = (float)R[i];
    }
    for (int i = 0; i < 3; ++i) a.t[i] = (float)t[i];
    return a;
}
}  // namespace

bool ProjectiveICP::runLevel(Affine3f& affine, const Intr& intr, int level, bool depth_variant, const void* curr, int curr_step,
                            const float* ncurr, int ncurr_step, const void* prev, int prev_step, const float* nprev,
                            int nprev_step, int cols, int rows) {
    const int div = 1 << level;  // setLevelIntr, :15-20
    for (int iter = 0; iter < schedule_[level]; ++iter) {
        float aff[12];
        affine.to12(aff);
        dfa::check(dfa_icp_sums(depth_variant ? 1 : 0, curr, curr_step, ncurr, ncurr_step, prev, prev_step, nprev, nprev_step,
                                cols, rows, aff, intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div, gate_dist_,
                                gate_angle_, sums_.ptr(), nullptr, nullptr),
                   "ProjectiveICP::estimateTransform");
        std::vector<float> h;
        sums_.download(h);  // synchronises (StreamHelper::get, :39-57)
        double A[36], b[6], x[6], det;
        int shift = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 7; ++j) {
                const double v = h[shift++];
                if (j == 6) b[i] = v;
                else A[6 * j + i] = A[6 * i + j] = v;
            }
        const bool ok = solve6(A, b, x, det);
        if (!ok || std::fabs(det) < 1e-15 || std::isnan(det)) return false;  // :136-142
        affine = from_rvec(x, x + 3) * affine;                                // :144-147
    }
    return true;
}

bool ProjectiveICP::estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr,
                                      const DepthPyr dprev, const NormalsPyr nprev) {
    affine = Affine3f::Identity();  // :124
    for (int level = getUsedLevelsNum() - 1; level >= 0; --level) {
        const Normals& n = nprev[level];
        if (!runLevel(affine, intr, level, true, dcurr[level].ptr(), (int)dcurr[level].step(), (const float*)ncurr[level].ptr(),
                     (int)ncurr[level].step(), dprev[level].ptr(), (int)dprev[level].step(), (const float*)n.ptr(),
                     (int)n.step(), n.cols(), n.rows()))
            return false;
    }
    return true;
}

bool ProjectiveICP::estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr,
                                      const PointsPyr vprev, const NormalsPyr nprev) {
    affine = Affine3f::Identity();  // :159
    for (int level = getUsedLevelsNum() - 1; level >= 0; --level) {
        const Normals& n = nprev[level];
        if (!runLevel(affine, intr, level, false, vcurr[level].ptr(), (int)vcurr[level].step(), (const float*)ncurr[level].ptr(),
                     (int)ncurr[level].step(), vprev[level].ptr(), (int)vprev[level].step(), (const float*)n.ptr(),
                     (int)n.step(), n.cols(), n.rows()))
            return false;
    }
    return true;
}

}  // namespace cuda
}  // namespace kfusion
