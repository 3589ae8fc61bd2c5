// dynfu/dyn_fusion.hpp — class DynFusion with the reference's interface (include/dynfu/dyn_fusion.hpp:25-90,
// src/dynfu/dyn_fusion.cpp).  As in the reference (:45) it IS-A kfusion::KinFu: the volume, the marching-cubes
// object, the frame counter, the pose chain and the inherited accessors (tsdf(), mc(), icp(), getCameraPose(),
// KinFu::params()) are the base class's; DynFusion::params() returns the DynFuParams and operator() replaces the
// rigid loop by the per-frame warp-field sequence of dyn_fusion.cpp:48-242 on the dynfu_amd C ABI:
//   init                    node seeding, every 128th canonical vertex            (:147-168)
//   addLiveFrame            store the live cloud                                  (:177-180)
//   warpCanonicalToLiveOpt  warp -> correspond -> build -> solve -> write-back    (:182-210)
//   findCorrespondingFrame  nearest warped-canonical vertex of every live vertex  (:212-242)
// plus fuse(): the TSDF steps of operator() (:58,:108-115 — dists, clear, integrate) on a caller-
// owned kfusion::cuda::TsdfVolume.  Vertices come from the caller (marching cubes is a later row).
#pragma once
#include <memory>
#include <vector>

#include <dfa_host/io.hpp>
#include <dynfu/utils/frame.hpp>
#include <dynfu/utils/northstar_solver.hpp>
#include <dynfu/utils/opt_solver.hpp>
#include <dynfu/warp_field.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/cuda/marching_cubes.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/kinfu.hpp>

struct DynFuParams {  // dyn_fusion.hpp:25-42
    static DynFuParams defaultParams();  // dyn_fusion.cpp:6-31
    kfusion::KinFuParams kinfuParams;
    kfusion::Intr intr;  // = kinfuParams.intr (kept for callers of fuse())
    float tukeyOffset;
    float lambda;
    float psi_data;
    float psi_reg;
    int L;
    int beta;
    float epsilon;
    // Extension (off = the reference's behaviour: default-constructed normals, dyn_fusion.cpp:80-88): normals of the
    // extracted vertices from the TSDF gradient (MarchingCubes::computeNormals)
    bool mesh_normals = false;
    // Extension — the north-star mode (BASELINE.json): operator() solves full 6-DoF node transforms
    // against the live DEPTH FRAME (NorthStarSolver: dual-quaternion blend, projective point-to-plane, ARAP) instead of
    // the reference's translations against the nearest live marching-cubes vertices.  The canonical cloud (with normals
    // from the TSDF gradient), the nodes and getCanonicalWarpedToLive() are then in the CAMERA frame (the frame of the
    // depth maps); getLiveFrame() / getMesh() stay the volume-frame marching-cubes output.
    bool north_star = false;
    NorthStarParameters northStarParams;
};

class DynFusion : public kfusion::KinFu {  // include/dynfu/dyn_fusion.hpp:45
public:
    explicit DynFusion(const DynFuParams& params);
    ~DynFusion();

    DynFuParams& params();

    // the whole per-frame sequence of the reference (dyn_fusion.cpp:48-145): pre-process the depth frame, fuse it,
    // extract the zero level set by marching cubes; frame 0 seeds the canonical frame and the warp field, later
    // frames become the live frame, the warp field is solved against it and grown.  Returns as the reference
    // does: false for the first frame, true afterwards.  The volume, its dimensions and the case tables come from
    // params().kinfuParams / the MarchingCubes passed to useMarchingCubes() (default: the library's tables).
    bool operator()(const kfusion::cuda::Depth& depth);
    void useMarchingCubes(std::shared_ptr<kfusion::cuda::MarchingCubes> mc) { mc_ = mc; }
    std::shared_ptr<dynfu::Frame> getLiveFrame() { return liveFrame; }
    // KinFu::getMesh (kinfu.cpp:262): the marching-cubes triangles of the last frame, KinFu::convertToMesh's layout
    // (built on first use: one small vector per triangle is not something every frame should pay for)
    std::shared_ptr<dfa::PolygonMesh> getMesh();
    // lets go of the mesh getMesh() built (one small vector per triangle: freeing them takes milliseconds, which a
    // caller may want outside its timed region; the next frame would do it otherwise)
    void dropMesh() { mesh_.reset(); }
    // north-star mode: energy before / after the last frame's solve and the data rows that found an association
    double northStarInitialCost() const { return ns_initial_cost_; }
    double northStarFinalCost() const { return ns_final_cost_; }
    long long northStarValidRows() const { return ns_valid_rows_; }

    void init(dfa::PointCloud<dfa::PointXYZ>& canonicalVertices, dfa::PointCloud<dfa::Normal>& canonicalNormals);
    void initCanonicalFrame(dfa::PointCloud<dfa::PointXYZ>& vertices, dfa::PointCloud<dfa::Normal>& normals);
    void addLiveFrame(int frameID, dfa::PointCloud<dfa::PointXYZ>& vertices, dfa::PointCloud<dfa::Normal>& normals);
    // the same two steps on frames that may live in HBM (what operator() uses: nothing is staged through the host)
    void initFromFrame(std::shared_ptr<dynfu::Frame> frame);
    void addLiveFrame(int frameID, std::shared_ptr<dynfu::Frame> frame);
    void warpCanonicalToLiveOpt(dfa::Affine3f affine);
    std::shared_ptr<dynfu::Frame> getCanonicalWarpedToLive();

    // private in the reference (dyn_fusion.hpp:86-89); public here so that it can be tested alone
    std::shared_ptr<dynfu::Frame> findCorrespondingFrame(dfa::PointCloud<dfa::PointXYZ> canonicalVertices,
                                                         dfa::PointCloud<dfa::Normal> canonicalNormals,
                                                         dfa::PointCloud<dfa::PointXYZ> liveVertices);
    std::shared_ptr<dynfu::Frame> findCorrespondingFrame(std::shared_ptr<dynfu::Frame> canonical,
                                                         std::shared_ptr<dynfu::Frame> live);

    // dists -> clear -> integrate of operator() (dyn_fusion.cpp:58, :108-115) as one fused sweep
    void fuse(const kfusion::cuda::Depth& depth, kfusion::cuda::TsdfVolume& volume, const dfa::Affine3f& camera_pose);

    // the iteration budget warpCanonicalToLiveOpt hands to the solver (dyn_fusion.cpp:183-189)
    CombinedSolverParameters solverParams;
    // vertices per deformation node at seeding (dyn_fusion.cpp:151)
    int nodeStep = 128;

    std::shared_ptr<Warpfield> getWarpfield() { return warpfield; }

private:
    DynFuParams dynfuParams;
    std::shared_ptr<dynfu::Frame> canonicalFrame;
    std::shared_ptr<dynfu::Frame> canonicalFrameWarpedToLive;
    std::shared_ptr<dynfu::Frame> liveFrame;
    std::shared_ptr<Warpfield> warpfield;
    std::shared_ptr<dfa::PolygonMesh> mesh_;
    // the last frame's marching-cubes output (:76 / :122): in HBM (a view of mc_buffer_, valid until the next frame),
    // downloaded when getMesh() asks for it
    dfa::DeviceArray<kfusion::cuda::MarchingCubes::PointType> mc_buffer_, mesh_source_;
    std::vector<dfa::PointXYZ> mesh_triangles_;
    bool mesh_downloaded_ = false;
    double ns_initial_cost_ = 0.0, ns_final_cost_ = 0.0;
    long long ns_valid_rows_ = 0;
    void seedNodes(const std::vector<dfa::PointXYZ>& canonicalVertices);
    // vertices of the volume's zero level set as a point cloud (dyn_fusion.cpp:73-88 / :119-134), device-resident
    std::shared_ptr<dynfu::Frame> extractSurface(int frame_id, bool with_normals);
    bool northStarFrame(const kfusion::cuda::Depth& depth);  // operator() in north-star mode
    kfusion::cuda::Cloud live_points_;
    kfusion::cuda::Normals live_normals_;
};

// DynFuApp::execute of the reference's demo (src/apps/demo.cpp:68-124) without its windows and command line: every
// depth PNG of <dir>/depth in lexicographic order is uploaded and handed to DynFusion::operator(); whenever that
// returns true, <dir>/out/pcl_canonical_to_live<i>.pcd and <dir>/out/<i>_tsdf_mesh.vtk are written (demo.cpp:21-37).
// <dir>/color must exist and hold at least as many files (the demo reads them for display only; they are not decoded
// here).  max_frames < 0: all.
struct SequenceReport {
    int frames = 0, saved = 0;
    double dynfu_ms = 0;  // time inside DynFusion::operator() (the demo's SampledScopeTime)
    std::vector<double> frame_ms;  // ... per frame
};
SequenceReport runSequence(DynFusion& dynfu, const std::string& dir, int max_frames = -1);
