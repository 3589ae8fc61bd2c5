// ref_nanoflann.cpp — thin C wrapper that runs the REFERENCE's own k-NN
// (/root/reference/include/nanoflann/nanoflann.hpp, vendored nanoflann 0x123) exactly as
// Warpfield does (src/dynfu/warp_field.cpp:23-27,111-122): L2_Simple_Adaptor<float>,
// 3-D KDTreeSingleIndexAdaptor, leaf size 10, knnSearch.  TEST INFRASTRUCTURE ONLY.
// The reference's dataset adaptor (include/nanoflann/pointcloud.hpp) stores cv::Vec3f, which
// needs OpenCV; nanoflann is templated on the dataset, so this file supplies a plain
// float[3] dataset instead — no reference header is replaced or stubbed.
#include <nanoflann.hpp>

#include <cstddef>
#include <cstdint>
#include <vector>

namespace {
struct Cloud {
    const float* pts;
    size_t n;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, int dim) const { return pts[3 * idx + dim]; }
    template <class BBOX>
    bool kdtree_get_bbox(BBOX&) const {
        return false;
    }
};
typedef nanoflann::L2_Simple_Adaptor<float, Cloud> Adaptor;
typedef nanoflann::KDTreeSingleIndexAdaptor<Adaptor, Cloud, 3> Tree;
}  // namespace

extern "C" void ref_knn(const float* nodes, int D, const float* query, int n_query, int k, int32_t* idx,
                        float* dist_sqr) {
    Cloud cloud = {nodes, (size_t)D};
    Tree tree(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    std::vector<size_t> ret(k);
    std::vector<float> d(k);
    for (int v = 0; v < n_query; ++v) {
        size_t n = tree.knnSearch(query + 3 * (size_t)v, (size_t)k, &ret[0], &d[0]);
        for (int j = 0; j < k; ++j) {
            idx[(size_t)v * k + j] = j < (int)n ? (int32_t)ret[j] : -1;
            if (dist_sqr) dist_sqr[(size_t)v * k + j] = j < (int)n ? d[j] : -1.f;
        }
    }
}
