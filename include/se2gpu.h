/* se2gpu.h - C ABI of the MI355X-native se2lam hot path (libse2gpu.so).
 *
 * The reference (izhengfan/se2lam) has no plugin / FFI layer: the boundary of its hot path
 * is three C++ header surfaces.  This C ABI is what a binding for those
 * surfaces calls; include/se2lam_amd/{ORBextractor,ORBmatcher,optimizer}.h are header-only C++
 * adapters over it that keep the reference's names and argument meaning.
 *
 * Conventions: every function returns an int status (SE2GPU_OK = 0, negative = error) and never
 * throws; the caller owns all host buffers; a handle owns its device memory and HIP stream and is
 * NOT thread-safe (one handle per calling thread, as the reference uses one ORBextractor per
 * thread and one SlamOptimizer per localBA call).  There is no CPU fallback: without a visible
 * gfx950 device every compute entry point returns SE2GPU_ERR_NO_DEVICE.
 */
#ifndef SE2GPU_H
#define SE2GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE2GPU_OK 0
#define SE2GPU_ERR_INVALID (-1)    /* bad argument / handle / id                      */
#define SE2GPU_ERR_NO_DEVICE (-2)  /* no HIP device (the library has no CPU fallback)  */
#define SE2GPU_ERR_HIP (-3)        /* a HIP runtime call failed; see se2gpu_last_error */
#define SE2GPU_ERR_CAPACITY (-4)   /* an internal or caller-supplied capacity overflowed */
#define SE2GPU_ERR_STATE (-5)      /* call sequence error (e.g. optimize before initialize) */

const char* se2gpu_last_error(void);    /* thread-local message of the last failing call */
int se2gpu_device_count(void);          /* number of visible HIP devices (0 = none)      */
const char* se2gpu_version(void);

/* ------------------------------------------------------------------------------------------
 * ORB extractor  -  replaces se2lam::ORBextractor
 *   ctor          /root/reference/include/se2lam/ORBextractor.h:44, src/ORBextractor.cpp:463-520
 *   operator()    /root/reference/include/se2lam/ORBextractor.h:49-51, src/ORBextractor.cpp:727-788
 * ------------------------------------------------------------------------------------------ */
typedef struct se2gpu_orb se2gpu_orb;

typedef struct se2gpu_orb_params {   /* defaults of ORBextractor.h:44 */
    int32_t nfeatures;     /* 1000 */
    float scale_factor;    /* 1.2f */
    int32_t nlevels;       /* 8    */
    int32_t score_type;    /* cv::ORB enum: 1 = FAST_SCORE (the reference's default), 0 = HARRIS_SCORE */
    int32_t fast_th;       /* 20   */
    int32_t max_rows, max_cols;  /* largest image the handle will see (0,0 -> 480,640) */
    int32_t max_batch;     /* frames per batched call (0 -> 1) */
} se2gpu_orb_params;

/* Layout-compatible with cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct se2gpu_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} se2gpu_keypoint;

int se2gpu_orb_create(const se2gpu_orb_params* params, se2gpu_orb** out);
void se2gpu_orb_destroy(se2gpu_orb* h);
int se2gpu_orb_levels(const se2gpu_orb* h);            /* GetLevels()      ORBextractor.h:53 */
float se2gpu_orb_scale_factor(const se2gpu_orb* h);    /* GetScaleFactor() ORBextractor.h:56 */

/* operator()(image, mask, keypoints, descriptors): `img` is a rows x cols CV_8UC1 host image with
 * row pitch `step`; `mask` is accepted and ignored, as in the reference (its cellMask never reaches cv::FAST,
 * ORBextractor.cpp:610-622; its only caller passes an empty mask, Frame.cpp:25).
 * Writes up to `cap` keypoints / cap*32 descriptor bytes, *n_out = number of keypoints.
 * An empty image (rows==0 || cols==0) returns OK with *n_out = 0 (ORBextractor.cpp:730-731). */
int se2gpu_orb_extract(se2gpu_orb* h, const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask,
                       se2gpu_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batched, device-resident form used for throughput (frames are independent):
 * d_imgs  = nframes contiguous rows x cols u8 images already in HBM (pitch = cols);
 * d_kps   = nframes * cap keypoints, d_desc = nframes * cap * 32 bytes, d_counts = nframes ints (device).
 * Asynchronous on the handle's stream; se2gpu_orb_sync() waits for it. */
int se2gpu_orb_extract_batch_device(se2gpu_orb* h, const uint8_t* d_imgs, int nframes, int rows, int cols,
                                    se2gpu_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap);
int se2gpu_orb_sync(se2gpu_orb* h);
int se2gpu_orb_set_stream(se2gpu_orb* h, void* hip_stream);  /* NULL -> the handle's own stream */

/* Introspection for the parity tests: copies pyramid level `level` of frame `frame` of the last call
 * (un-blurred if bit 0 of `blurred` is clear) into `out` (rows*cols of that level, tight pitch); rows and cols are set.
 * Bit 1 of `blurred` includes the 16 px reflect-101 frame: (rows + 32) x (cols + 32). */
int se2gpu_orb_debug_level(se2gpu_orb* h, int frame, int level, int blurred, uint8_t* out, size_t out_cap,
                           int* rows, int* cols);
/* FAST score map S of a level of the last call, same geometry as the level. */
int se2gpu_orb_debug_score(se2gpu_orb* h, int frame, int level, uint8_t* out, size_t out_cap, int* rows, int* cols);
/* The device routine behind both retainBest cuts (/root/reference/src/ORBextractor.cpp:692-694, 708-709 = cv::KeyPointsFilter::
 * retainBest + resize = std::nth_element by response, first n kept), run on caller data for the parity tests: `entries` (host,
 * in / out) are n values whose HIGH 32 bits are the comparison key (larger = better response) and whose low 32 bits travel
 * along; on return they are permuted exactly as libstdc++'s std::nth_element(first, first + nth, last, key greater) leaves
 * them.  force_global != 0 runs the in-place global-memory path that cells beyond the LDS capacity take. */
int se2gpu_orb_debug_nth_element(uint64_t* entries, int n, int nth, int force_global);

/* ------------------------------------------------------------------------------------------
 * ORB matcher  -  replaces se2lam::ORBmatcher
 *   DescriptorDistance   /root/reference/src/ORBmatcher.cpp:110-126
 *   MatchByWindow        /root/reference/src/ORBmatcher.cpp:278-381   (+ Frame::GetFeaturesInArea,
 *                        PosInGrid: /root/reference/src/Frame.cpp:209-286)
 *   MatchByProjection    /root/reference/src/ORBmatcher.cpp:383-454
 *   SearchByBoW          /root/reference/src/ORBmatcher.cpp:128-276
 * ------------------------------------------------------------------------------------------ */
int se2gpu_hamming(const uint8_t* a, const uint8_t* b);  /* host utility, 256-bit Hamming distance */
/* ORBmatcher::ComputeThreeMaxima (/root/reference/include/se2lam/ORBmatcher.h:57, src/ORBmatcher.cpp:64-105): host utility
 * over the bin COUNTS (histo[i].size()); ind1..3 in/out, the reference's callers start them at -1 */
int se2gpu_three_maxima(const int32_t* counts, int L, int* ind1, int* ind2, int* ind3);

typedef struct se2gpu_matcher se2gpu_matcher;
int se2gpu_matcher_create(int max_features, int max_batch, se2gpu_matcher** out);
void se2gpu_matcher_destroy(se2gpu_matcher* h);
int se2gpu_matcher_set_stream(se2gpu_matcher* h, void* hip_stream);
int se2gpu_matcher_sync(se2gpu_matcher* h);
/* Number of calls of this handle in which some query's search window held more than the 128 candidates the fast path lists
 * and was resolved by the exact grid scan instead (same result, a latency cliff on clustered features): lets the tracking
 * thread see the slow path, which is otherwise silent. */
int se2gpu_matcher_spill_calls(const se2gpu_matcher* h, long long* calls);

/* Image bounds / grid of Frame (Frame.cpp:37-44): minX,minY,maxX,maxY of the undistorted image. */
typedef struct se2gpu_frame_bounds {
    float min_x, min_y, max_x, max_y;
} se2gpu_frame_bounds;

/* MatchByWindow(frame1, frame2, vbPrevMatched, winSize, vnMatches12, levelOffset, minLevel, maxLevel)
 * with ORBmatcher(nnratio, checkOri=true).  Host buffers.  prev_xy (n1 x 2 floats) is updated in place
 * as the reference updates vbPrevMatched (ORBmatcher.cpp:375-377).  matches12: n1 ints (-1 = none).
 * Returns the match count in *n_matches. */
int se2gpu_match_window(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds,
                        const se2gpu_keypoint* kps1, const uint8_t* desc1, int n1,
                        const se2gpu_keypoint* kps2, const uint8_t* desc2, int n2,
                        float* prev_xy, int win_size, int level_offset, int min_level, int max_level,
                        float nnratio, int32_t* matches12, int* n_matches);

/* Batched device-resident MatchByWindow over `npairs` independent frame pairs.  Pair p matches
 * frame a[p] against frame b[p] of the same device arrays the batched extractor wrote
 * (d_kps: nframes*cap, d_desc: nframes*cap*32, d_counts: nframes).  prev_xy = keypoint positions of
 * frame a (Track::resetLocalTrack, Track.cpp:194).  d_matches12: npairs*cap ints, d_nmatches: npairs. */
int se2gpu_match_window_batch_device(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds,
                                     const se2gpu_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts,
                                     int cap, const int32_t* d_pair_a, const int32_t* d_pair_b, int npairs,
                                     int win_size, int level_offset, int min_level, int max_level, float nnratio,
                                     int32_t* d_matches12, int32_t* d_nmatches);

/* MatchByProjection(pNewKF, localMPs, winSize, levelOffset, vMatchesIdxMP) with ORBmatcher(nnratio).
 *  map points: mp_pos (m x 3 float, world), mp_desc (m x 32), mp_octave (m), mp_skip (m; 1 = the
 *  reference would `continue` at ORBmatcher.cpp:392-395: null / bad parallax / already observed);
 *  key frame:  Tcw (3x4 row-major float), K = fx, fy, cx, cy (float), kps/desc (n),
 *  kf_observed (n; 1 = hasObservation(idx), ORBmatcher.cpp:417).
 *  match_idx_mp: n ints (index into the map-point array, -1 = none). */
int se2gpu_match_projection(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds,
                            const float* mp_pos, const uint8_t* mp_desc, const int32_t* mp_octave,
                            const uint8_t* mp_skip, int m,
                            const float* Tcw, float fx, float fy, float cx, float cy,
                            const se2gpu_keypoint* kps, const uint8_t* desc, const uint8_t* kf_observed, int n,
                            int win_size, int level_offset, float nnratio, int32_t* match_idx_mp, int* n_matches);

/* SearchByBoW(pKF1, pKF2, mapMatches12, bIfMPOnly) with ORBmatcher(nnratio, checkOri) - ORBmatcher.cpp:128-276.
 * The DBoW2::FeatureVector of each key frame (std::map<NodeId, std::vector<unsigned>>, computed by the host
 * vocabulary) is passed as CSR: fv_nodes[nn] ascending node ids, fv_ptr[nn+1], fv_idx[fv_ptr[nn]] feature indices.
 * has_mp1/2 (n1 / n2 bytes): 1 = the feature has a non-null map point (only read when mp_only != 0).
 * matches12: n1 ints, index into key frame 2 or -1 (the reference's std::map<int,int> as a dense array). */
int se2gpu_search_by_bow(se2gpu_matcher* h,
                         const se2gpu_keypoint* kps1, const uint8_t* desc1, int n1,
                         const int32_t* fv1_nodes, const int32_t* fv1_ptr, const int32_t* fv1_idx, int nn1,
                         const uint8_t* has_mp1,
                         const se2gpu_keypoint* kps2, const uint8_t* desc2, int n2,
                         const int32_t* fv2_nodes, const int32_t* fv2_ptr, const int32_t* fv2_idx, int nn2,
                         const uint8_t* has_mp2,
                         int mp_only, float nnratio, int check_orientation, int32_t* matches12, int* n_matches);

/* ------------------------------------------------------------------------------------------
 * SE(2)-XYZ bundle adjustment  -  replaces the g2o::SparseOptimizer built by
 *   LocalMapper::localBA          /root/reference/src/LocalMapper.cpp:232-302
 *   Map::loadLocalGraph           /root/reference/src/Map.cpp:891-1053
 * through the free functions of  /root/reference/include/se2lam/optimizer.h:
 *   addCamPara (:85) addVertexSE2 (:104) addVertexSBAXYZ (:91) addEdgeSE2XYZ (:100) addEdgeSE2 (:109)
 *   estimateVertexSE2 (:107) estimateVertexSBAXYZ (:141) initOptimizer (:78)
 * and SlamOptimizer::{setForceStopFlag, initializeOptimization, optimize, clear}.
 * Vertex ids are the caller's (Map.cpp:925,966,985); poses and landmarks share one id space.
 * ------------------------------------------------------------------------------------------ */
typedef struct se2gpu_ba se2gpu_ba;

typedef struct se2gpu_ba_stats {
    int32_t iterations;   /* outer iterations executed (SparseOptimizer::optimize return value) */
    int32_t trials;       /* LM trials = linear solves */
    int32_t terminated;   /* 1 = algorithm returned Terminate (10 failed trials or rho == 0)   */
    int32_t stopped;      /* 1 = left early because *stop_flag became true */
    double chi2_init, chi2_final, lambda_final;
    double chi2_hist[64], lambda_hist[64];
    int32_t trials_hist[64];
} se2gpu_ba_stats;

int se2gpu_ba_create(se2gpu_ba** out);
void se2gpu_ba_destroy(se2gpu_ba* h);
/* Start-up pre-warm: runs one throw-away window of P key frames / L landmarks / E observations and parks its handle (code
 * objects loaded, stream + mailbox + every device buffer allocated at that size), so that the FIRST localBA of the process
 * (a SlamOptimizer constructed on the stack, LocalMapper.cpp:239) costs what the later ones do.  Off the tracking thread. */
int se2gpu_ba_reserve(int P, int L, int E);
int se2gpu_ba_clear(se2gpu_ba* h);                                   /* optimizer.clear(); clearParameters() */
int se2gpu_ba_set_stream(se2gpu_ba* h, void* hip_stream);
int se2gpu_ba_add_cam(se2gpu_ba* h, double f, double cx, double cy);  /* addCamPara: single focal length */
int se2gpu_ba_set_Tbc(se2gpu_ba* h, const double R[9], const double t[3]); /* setExtParameter(Tbc), row-major R */
int se2gpu_ba_add_vertex_se2(se2gpu_ba* h, int id, double x, double y, double theta, int fixed);
int se2gpu_ba_add_vertex_xyz(se2gpu_ba* h, int id, const double xyz[3], int marginal, int fixed);
int se2gpu_ba_add_edge_se2xyz(se2gpu_ba* h, int id_kf, int id_mp, const double uv[2], const double info[4],
                              double huber_delta);
int se2gpu_ba_add_edge_se2(se2gpu_ba* h, int id0, int id1, const double meas[3], const double info[9]);

/* Bulk form of the same calls (SoA arena filled once per key frame):
 * vertex ids: poses 0..P-1, landmarks P..P+L-1.  e_info: E x 3 (xx, xy, yy).  o_info: O x 9.
 * The four edge arrays are BORROWED: they must stay valid and unchanged until se2gpu_ba_initialize returns (which
 * validates them and copies them once, straight into the pinned upload arena); everything else is copied here. */
int se2gpu_ba_load(se2gpu_ba* h, int P, int L, int E, int O,
                   const double* poses, const uint8_t* fixed, const double* lms,
                   const int32_t* e_kf, const int32_t* e_lm, const double* e_uv, const double* e_info,
                   const int32_t* o_i, const int32_t* o_j, const double* o_meas, const double* o_info,
                   double huber_delta);

/* ---- SE3-expmap graphs: the marginalising local bundle adjustment of
 *   Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx)   /root/reference/src/Map.cpp:414-566
 *   LocalMapper::removeOutlierChi2                          /root/reference/src/LocalMapper.cpp:172-230
 * on the same handle type: a graph is EITHER SE(2) (add_vertex_se2 ...) OR SE3 (the calls below), decided by its first
 * pose vertex.  pose12 = rotation row-major (9) then translation (3) of Tcw (x_c = R x_w + t); 6x6 information matrices
 * row-major in g2o's vector order (rotation, translation).  Replaces, of /root/reference/include/se2lam/optimizer.h:
 *   addVertexSE3Expmap (:88)                     -> se2gpu_ba_add_vertex_se3
 *   addPlaneMotionSE3Expmap (:82)                -> se2gpu_plane_motion_prior (host) + se2gpu_ba_add_prior_se3
 *   addEdgeSE3Expmap (:94)                       -> se2gpu_ba_add_edge_se3   (error = log(T_id1^-1 * measure * T_id0))
 *   addEdgeXYZ2UV (:97)                          -> se2gpu_ba_add_edge_xyz2uv (information = inv_sigma2 * I, Huber)
 *   addVertexSBAXYZ / estimateVertexSBAXYZ       -> se2gpu_ba_add_vertex_xyz / se2gpu_ba_get_xyz (shared)
 *   estimateVertexSE3Expmap (:138)               -> se2gpu_ba_get_se3
 *   EdgeProjectXYZ2UV::computeError() + chi2()   -> se2gpu_ba_edge_chi2 (all edges, in the order they were added)
 * initialize / optimize / optimize_batch / chi2 / clear are the common ones.  Single GPU only. */
int se2gpu_ba_add_vertex_se3(se2gpu_ba* h, int id, const double pose12[12], int fixed);
/* ---- pose graphs: GlobalMapper::GlobalBA (/root/reference/src/GlobalMapper.cpp:328-535).
 * A graph whose first pose vertex is added with se2gpu_ba_add_vertex_iso3 is a g2o::VertexSE3 pose graph (poses T_w_c,
 * update estimate * fromVectorMQT(d)); se2gpu_ba_add_prior_se3 then adds an EdgeSE3Prior (what
 * addVertexSE3PlaneMotion, optimizer.h:123, attaches: build it with se2gpu_plane_motion_prior_iso3), se2gpu_ba_add_edge_se3
 * an EdgeSE3 (addEdgeSE3, optimizer.h:129: error toVectorMQT(measure^-1 * T_id0^-1 * T_id1); several edges between the
 * same two key frames are allowed - odometry + feature edge).  Vector order (translation, rotation) for errors and
 * information matrices.  se2gpu_ba_get_se3 = estimateVertexSE3 (:135), se2gpu_ba_edge_chi2 = EdgeSE3::chi2() of every
 * edge in insertion order (the feature-edge rejection of GlobalMapper.cpp:415-437, 455-476).  No landmarks. */
int se2gpu_ba_add_vertex_iso3(se2gpu_ba* h, int id, const double Twc12[12], int fixed);
int se2gpu_plane_motion_prior_iso3(const double* Twc12, const double* Tbc12, double xrot_info, double yrot_info,
                                   double z_info, double* meas12, double* info36);
int se2gpu_ba_add_prior_se3(se2gpu_ba* h, int id, const double meas12[12], const double info36[36]);
int se2gpu_ba_add_edge_se3(se2gpu_ba* h, int id0, int id1, const double meas12[12], const double info36[36]);
int se2gpu_ba_add_edge_xyz2uv(se2gpu_ba* h, int id_mp, int id_kf, const double uv[2], double inv_sigma2, double huber_delta);
int se2gpu_ba_get_se3(se2gpu_ba* h, int id, double pose12[12]);
int se2gpu_ba_edge_chi2(se2gpu_ba* h, double* chi2, int cap);

/* Sparsifier::DoMarginalizeSE3XYZ (/root/reference/src/sparsifier.cpp:105-275) for a batch of key-frame pairs - SURVEY.md
 * section 8f.4: the feature constraint (relative pose + 6x6 information, order (translation, rotation) of
 * SE3Quat::toMinimalVector) that GlobalMapper::CreateFeatEdge (src/GlobalMapper.cpp:744-840) stores in mFtrMeasureFrom and
 * GlobalBA turns into an EdgeSE3.  Pair p: kf12[p] = T_w_c of the two key frames (2 x 12), its map points
 * mp_xyz[mp_ptr[p] .. mp_ptr[p+1]) and its measurements [m_ptr[p], m_ptr[p+1]): m_kf in {0, 1} (others are ignored, as
 * :117-119), m_mp = index of the point inside the pair, m_info = 3x3 information (row-major; MeasSE3XYZ::info).
 * z_out12[p] = pose12 of KF0^-1 * KF1, info_out36[p] row-major.  One wave per pair; host buffers. */
int se2gpu_sparsify_se3xyz(int npairs, const double* kf12, const int32_t* mp_ptr, const double* mp_xyz,
                           const int32_t* m_ptr, const int32_t* m_kf, const int32_t* m_mp, const double* m_info,
                           double* z_out12, double* info_out36);

/* Map::loadLocalGraph(SlamOptimizer&) (/root/reference/src/Map.cpp:891-1022) as ONE call on a POD view of the local
 * window - SURVEY.md section 8f.1.  The caller flattens its pointer graph once per key frame (INTEGRATION.md shows the
 * 30 lines that do it with one hash map instead of the reference's std::find per observation); the library applies
 * the reference's vertex numbering (local key frame i -> id i, reference key frame i -> n_local + i, map point i ->
 * n_local + n_ref + 1 + i), its fixed rule (no reference key frames: the local one with the smallest KeyFrame::id is
 * fixed; KeyFrame::id == 1 always; reference key frames always), inverts the PreSE2 covariances, and evaluates the
 * per-observation information matrices (:1024-1049) on the device during se2gpu_ba_initialize.  The optimizer must be
 * empty; afterwards initialize / optimize / get_* as usual with those ids. */
typedef struct se2gpu_local_graph {
    int32_t n_local_kf, n_ref_kf, n_mp, n_obs;
    /* key frames: mLocalGraphKFs first, then mRefKFs                                    [n_local_kf + n_ref_kf] entries */
    const int32_t* kf_id;     /* KeyFrame::id */
    const float* kf_Twb;      /* Se2 Twb: x, y, theta                                      x 3 */
    const float* kf_Rcw;      /* rotation block of KeyFrame::Tcw (CV_32F), row-major       x 9 */
    /* odometry of the local key frames (KeyFrame::preOdomFromSelf)                       [n_local_kf] entries, may be NULL */
    const int32_t* odo_to;    /* position of preOdomFromSelf.first in mLocalGraphKFs, -1 = not in the window / null */
    const double* odo_meas;   /* PreSE2::meas                                              x 3 */
    const double* odo_cov;    /* PreSE2::cov, row-major                                    x 9 */
    /* map points (mLocalGraphMPs) and their observations, grouped by map point            [n_mp] / [n_obs] entries */
    const float* mp_pos;      /* MapPoint::getPos()                                        x 3 */
    const int32_t* obs_mp;    /* map point of the observation, nondecreasing */
    const int32_t* obs_kf;    /* observing key frame: position in the list above, -1 = in neither list (skipped, :1016) */
    const float* obs_uv;      /* keyPointsUn[ftrIdx].pt                                    x 2 */
    const float* obs_lc;      /* mViewMPs[ftrIdx]                                          x 3 */
    const float* obs_sigma2;  /* mvLevelSigma2[octave] */
    float fx, cx, cy;         /* Config::Kcam (addCamPara uses fx for both axes) */
    double Rbc[9], tbc[3];    /* Config::bTc, rotation row-major */
    float huber_delta;        /* Config::TH_HUBER */
    float xrot_info, z_info;  /* Config::PLANEMOTION_XROT_INFO, PLANEMOTION_Z_INFO */
} se2gpu_local_graph;
int se2gpu_ba_load_local_graph(se2gpu_ba* h, const se2gpu_local_graph* graph);

/* Map::updateLocalGraph() (/root/reference/src/Map.cpp:285-331) on a CSR view of the map - SURVEY.md section 8 row a25: the
 * local window of the current key frame = key frames within `search_level` (3 in the reference) covisibility hops, the
 * map points they observe, and as reference key frames every other observer of those points.  Pure host code (no device
 * needed); the three outputs are positions in the view's arrays, ordered by KeyFrame::id / MapPoint::id like the
 * reference's sets, i.e. exactly mLocalGraphKFs, mRefKFs, mLocalGraphMPs - the lists se2gpu_ba_load_local_graph takes.
 * Output arrays may be NULL (sizes only); capacities n_kf / n_kf / n_mp always suffice. */
typedef struct se2gpu_map_view {
    int32_t n_kf, n_mp;
    const int32_t* kf_id;                        /* KeyFrame::id                                  [n_kf] */
    const int32_t* covis_ptr; const int32_t* covis_idx;   /* KeyFrame::getAllCovisibleKFs() as CSR        [n_kf + 1] */
    const int32_t* kf_mp_ptr; const int32_t* kf_mp_idx;   /* KeyFrame::getAllObsMPs(false) as CSR         [n_kf + 1] */
    const int32_t* mp_id;                        /* MapPoint::id                                  [n_mp] */
    const int32_t* mp_kf_ptr; const int32_t* mp_kf_idx;   /* MapPoint::getObservations() as CSR           [n_mp + 1] */
} se2gpu_map_view;
int se2gpu_map_update_local_graph(const se2gpu_map_view* map, int current_kf, int search_level, int32_t* local_kfs,
                                  int* n_local, int32_t* ref_kfs, int* n_ref, int32_t* local_mps, int* n_mps);

/* initializeOptimization(0): freezes the graph, builds the device-side SoA + reduction plans (on the device).
 * A second call on an initialised handle is g2o's "initialise again over the edges of level 0": supported for the pose graph
 * (VertexSE3 / EdgeSE3, the graph of GlobalMapper::GlobalBA, /root/reference/src/GlobalMapper.cpp:421-483): the vertices keep
 * their current estimates, edges moved to another level by se2gpu_ba_set_edge_level are left out.  For the landmark models
 * it returns SE2GPU_ERR_STATE. */
int se2gpu_ba_initialize(se2gpu_ba* h);
/* g2o::OptimizableGraph::Edge::setLevel(level) of the `edge`-th EdgeSE3 added to a pose graph (the index
 * se2gpu_ba_edge_chi2 reports it under); takes effect at the next se2gpu_ba_initialize. */
int se2gpu_ba_set_edge_level(se2gpu_ba* h, int edge, int level);
/* restores every vertex estimate to the value the device graph was built with (device-to-device): the value it was added with -
 * or, on a pose graph that was initialised a second time, the estimates that second se2gpu_ba_initialize started from (g2o has
 * no such reset; a re-initialise moves the reset point) */
int se2gpu_ba_reset_estimates(se2gpu_ba* h);
/* ... of `count` windows with one launch (the companion of se2gpu_ba_optimize_batch for a mapper that re-optimises the same
 * windows): ordered before any later operation on each of the windows */
int se2gpu_ba_reset_estimates_batch(se2gpu_ba** handles, int count);

#define SE2GPU_BA_LM 0  /* OptimizationAlgorithmLevenberg, g2o policy (optimizer.h:32) */
#define SE2GPU_BA_GN 1  /* plain Gauss-Newton: lambda = 0, every step accepted           */
/* optimize(iters); stop_flag mirrors setForceStopFlag(bool*) (LocalMapper.cpp:246), may be NULL. */
int se2gpu_ba_optimize(se2gpu_ba* h, int iters, int mode, const volatile uint8_t* stop_flag, int verbose,
                       se2gpu_ba_stats* stats);
/* optimize(iters) of `count` independent windows at once - one initialised handle per window, each on its own stream.
 * The Levenberg-Marquardt controller of every window runs on the device, so all windows are enqueued before the first
 * wait and the GPU works on them concurrently (a 50-KF local window occupies a few per cent of an MI355X).  This is the
 * throughput form for a mapper that keeps several local windows (or several robots' maps) in flight; results are
 * identical to calling se2gpu_ba_optimize on every handle in turn.  stats: NULL or `count` entries. */
int se2gpu_ba_optimize_batch(se2gpu_ba** handles, int count, int iters, int mode, const volatile uint8_t* stop_flag,
                             se2gpu_ba_stats* stats);
/* Which of its three paths the calling thread's last se2gpu_ba_optimize_batch took (-1: none yet): 0 = one stream per window,
 * 1 = lock step (one launch per stage for all windows), 2 = resident (one workgroup per window for its whole optimize():
 * batches of SE2GPU_BA_RESIDENT_MIN = 96 windows or more whose windows fit a compute unit's LDS, csrc/ba_window.hip; its sums
 * are atomic, so its results equal the other paths' to rounding, not to the bit).  SE2GPU_BA_RESIDENT=0 / 1 forces the choice. */
int se2gpu_ba_last_batch_path(void);
int se2gpu_ba_get_se2(se2gpu_ba* h, int id, double xyt[3]);   /* estimateVertexSE2   */
int se2gpu_ba_get_xyz(se2gpu_ba* h, int id, double xyz[3]);   /* estimateVertexSBAXYZ */
int se2gpu_ba_get_all(se2gpu_ba* h, double* poses /*P*3, in pose-add order*/, double* lms /*L*3*/);
double se2gpu_ba_chi2(se2gpu_ba* h);                          /* activeRobustChi2() at the current estimate; <0 on error */

/* Parity introspection: reduced (Schur) system at the current estimate and damping `lambda`:
 * S (3P x 3P row-major, fixed poses -> identity rows), bs (3P).  P counts poses in add order. */
int se2gpu_ba_debug_reduced_system(se2gpu_ba* h, double lambda, double* S, double* bs);
/* ... and its dense solve on the device (the LL^T that stands in for CHOLMOD): x (3P) with S x = bs;
 * *factor_ok = 0 when a pivot was not positive. */
int se2gpu_ba_debug_solve(se2gpu_ba* h, double lambda, double* x, int* factor_ok);
/* Which dense solver an initialised handle runs: 0 = the dataflow launch (k_chol_tiles), 1 = one launch per block column
 * by configuration (SE2GPU_BA_CHOL=steps, more than 64 tile rows), 2 = one launch per block column because a dataflow
 * dependency timed out earlier in the handle's life (reported once on stderr), 3 = host solve (SE2GPU_BA_HOST_SOLVE=1).
 * Tests use it to make sure that results were not produced by the fallback. */
int se2gpu_ba_debug_solver_path(const se2gpu_ba* h);
/* Test introspection: idle sets in the library's two process-wide lease pools - {plan caches of the lock-step batch driver,
 * staging buffers of se2gpu_ba_reset_estimates_batch}.  A set is leased per call and returned, whichever thread calls. */
int se2gpu_ba_debug_pool_sizes(int out2[2]);
/* Soak-test introspection of the dataflow solve's tile hand-offs.  With SE2GPU_BA_CHOL_VERIFY=1 in the environment every
 * published half-slab carries a checksum and every consumer checks its loads against it: counts2 = {mismatches, half-slabs
 * checked}; records (cap x 8 words, may be NULL): {epoch, consumer task, kind << 32 | tile row << 16 | column, slab << 8 |
 * part, consumer XCC, got, want, 100 MHz time stamp}.  SE2GPU_ERR_STATE when the handle does not run in that mode. */
int se2gpu_ba_debug_chol_verify(se2gpu_ba* h, unsigned long long* counts2, unsigned long long* records, int cap);
/* The plan of the dense pose solve (tile tasks, their dependency lists, the fill-reducing order of the poses) for a P x P
 * block pattern, computed on the host without a device - what stands in for CHOLMOD's symbolic analysis behind
 * /root/reference/include/se2lam/optimizer.h:31.  Test introspection (tests/test_solve_plan.py). */
int se2gpu_ba_debug_solve_plan(int P, int D, const uint8_t* pattern, int allow_nd, int* nsys, int* nbc, int* depth, int* ntask,
                               int* ndep, int32_t* pose_off, int32_t* tasks4, int task_cap, int32_t* deps, int dep_cap);
/* the same with the tile size of the dense solve as an argument (32 or 64) */
int se2gpu_ba_debug_solve_plan_tile(int P, int D, const uint8_t* pattern, int allow_nd, int tile, int* nsys, int* nbc, int* depth, int* ntask,
                                    int* ndep, int32_t* pose_off, int32_t* tasks4, int task_cap, int32_t* deps, int dep_cap);

/* Track::doTriangulate (/root/reference/src/Track.cpp:378-419) for every match of a frame pair in one device pass -
 * SURVEY section 8(f).3.  Per feature i of the reference key frame with match_idx[i] >= 0 and no map point yet:
 *   pos = cvu::triangulate(kps_ref[i].pt, kps_cur[match_idx[i]].pt, P_ref, P_cur)   (cvutil.cpp:46-59: DLT, smallest
 *         right singular vector of the 4x4 system; one-sided Jacobi like cv::SVD, FP64 inside, FP32 in / out)
 *   Config::acceptDepth(pos.z) (Config.cpp:188-190)  ? pos_out[i] = pos, good_parallax[i] = cvu::checkParallax(0, Ocam,
 *         pos, min_degree) (cvutil.cpp:92-98)        : match_idx[i] = -1
 * Features with has_observation[i] != 0 are only counted (*n_tracked_old; the caller keeps the key frame's map point).
 * P_ref = Config::PrjMtrxEye, P_cur = Config::Kcam * Tcr.rowRange(0,3): 3x4 row-major float.  Host buffers. */
int se2gpu_triangulate(int n, const se2gpu_keypoint* kps_ref, const se2gpu_keypoint* kps_cur, int n_cur,
                       int32_t* match_idx, const uint8_t* has_observation, const float* P_ref, const float* P_cur,
                       const float* Ocam, float lower_depth, float upper_depth, int min_degree, float* pos_out,
                       uint8_t* good_parallax, int* n_good, int* n_tracked_old);

/* Track::removeOutliers (/root/reference/src/Track.cpp:308-344) - SURVEY section 8(f).3: the epipolar filter applied to
 * the MatchByWindow result in Track::mTrack (Track.cpp:134).  The handle owns a stream and the staging / device buffers
 * of the calling thread (Track is single-threaded); it is not thread-safe.
 *   se2gpu_track_fundamental_mask  = cv::findFundamentalMat(pt1, pt2, mask) with the defaults FM_RANSAC, 3 px, 0.99
 *       [OpenCV 3.2]: n < 7 no mask (all 0 here, *n_inliers = 0); n == 7 all 1; 8..14 LMedS; >= 15 RANSAC over 7-point
 *       samples drawn by cv::RNG(-1), at most 1000 iterations, adaptive stop.  pt1 / pt2: n x (x, y) float.
 *       No model found => mask all 0 (the reference leaves the mask uninitialised in that case).
 *   se2gpu_track_remove_outliers   = the whole member function: matches[i] (n1 entries, index into kps2 or -1) is set to
 *       -1 for the outliers; with fewer than 10 inliers every match is dropped and *n_inliers = 0.
 *   se2gpu_track_last_ransac       = {inliers, winning sample, winning model of that sample, iterations the reference's
 *       loop would have run} of the last call, for tracing. */
typedef struct se2gpu_track se2gpu_track;
int se2gpu_track_create(se2gpu_track** out);
void se2gpu_track_destroy(se2gpu_track* h);
int se2gpu_track_fundamental_mask(se2gpu_track* h, const float* pt1, const float* pt2, int n, uint8_t* mask,
                                  int* n_inliers);
int se2gpu_track_remove_outliers(se2gpu_track* h, const se2gpu_keypoint* kps1, int n1, const se2gpu_keypoint* kps2,
                                 int n2, int32_t* matches, int* n_inliers);
int se2gpu_track_last_ransac(const se2gpu_track* h, int info[4]);
/* se2gpu_triangulate (below) on the workspace of the tracking thread: one packed upload / download, no allocation */
int se2gpu_track_triangulate(se2gpu_track* h, int n, const se2gpu_keypoint* kps_ref, const se2gpu_keypoint* kps_cur,
                             int n_cur, int32_t* match_idx, const uint8_t* has_observation, const float* P_ref,
                             const float* P_cur, const float* Ocam, float lower_depth, float upper_depth,
                             int min_degree, float* pos_out, uint8_t* good_parallax, int* n_good, int* n_tracked_old);

/* Localizer::DoLocalBA (/root/reference/src/Localizer.cpp:233-302) - SURVEY section 8(f).2: pose-only bundle adjustment of one
 * key frame against the fixed map points it observes, the whole optimize(iters) in one launch.
 *   pose12      = rotation row-major (9) then translation (3) of a rigid transform, x_c = R x_w + t (Tcw)
 *   se2gpu_plane_motion_prior  = addPlaneMotionSE3Expmap (src/optimizer.cpp:236-314): measurement = the pose with the body's
 *       roll, pitch and height removed (Tbc = Config::bTc), information = adj(Tbc)^T diag(xrot, yrot, 1e-4, 1e-4, 1e-4, z)
 *       adj(Tbc), vector order (rotation, translation); graph construction, runs on the host
 *   se2gpu_track_pose_ba       = VertexSE3Expmap + n EdgeProjectXYZ2UV (information inv_sigma2[i] * I, Huber huber_delta,
 *       single focal length f as addCamPara builds g2o::CameraParameters) + EdgeSE3ExpmapPrior(prior_meas, prior_info),
 *       Levenberg-Marquardt with g2o's policy; xyz: n x 3 world points, uv: n x 2 key-point positions.
 *       stats has the meaning it has for se2gpu_ba_optimize. */
int se2gpu_plane_motion_prior(const double* Tcw12, const double* Tbc12, double xrot_info, double yrot_info, double z_info,
                              double* meas12, double* info36);
int se2gpu_track_pose_ba(se2gpu_track* h, const double* Tcw12, const double* prior_meas12, const double* prior_info36, int n,
                         const double* xyz, const double* uv, const double* inv_sigma2, double f, double cx, double cy,
                         double huber_delta, int iters, double* Tcw_out12, se2gpu_ba_stats* stats);

/* Per-observation information matrices of Map::loadLocalGraph (/root/reference/src/Map.cpp:1024-1049), SURVEY §8f.1:
 *   Sigma = s_rot * J_r J_r^T + s_z * J_z J_z^T + sigma2 * I,   Omega = Sigma^-1       (2x2, FP64)
 *   J_r = (J_pi Rcw skew(lw - p))[:, 0:2],  J_z = -(J_pi Rcw)[:, 2],  J_pi from the stored camera-frame point lc.
 * Host buffers: lc, lw (E x 3 float: KeyFrame::mViewMPs[ftrIdx], MapPoint::getPos()), e_kf (E, index into the P
 * key frames), sigma2 (E float: mvLevelSigma2[octave]), Rcw (P x 9 float, row-major rotation of KeyFrame::Tcw),
 * twb_xy (P x 2 float: Twb.x, Twb.y), fx = Config::fxCam, xrot_info / z_info = Config::PLANEMOTION_XROT_INFO / _Z_INFO.
 * info_out: E x 4 doubles (row-major 2x2), ready for se2gpu_ba_add_edge_se2xyz. */
int se2gpu_ba_edge_information(int E, const float* lc, const float* lw, const int32_t* e_kf, const float* sigma2, int P,
                               const float* Rcw, const float* twb_xy, float fx, float xrot_info, float z_info,
                               double* info_out);

/* Multi-GPU (landmark-sharded) BA, SURVEY.md §8e: every rank holds all poses and a shard of the
 * landmarks (+ their edges); odometry edges live on one rank.  Once per LM trial the library calls
 *     allreduce(dev_ptr, count_doubles, hip_stream, user)
 * which must sum `count_doubles` FP64 values in place over all ranks, ordered on `hip_stream`
 * (RCCL ncclAllReduce(ncclDouble, ncclSum), or torch.distributed.all_reduce on a tensor that
 * aliases the buffer).  The fused buffer is [S (3P*3P) | bs (3P) | 4 scalars]; a second call
 * reduces the 4 trial scalars.  If `buffer` is non-NULL it must be a device allocation of at least
 * se2gpu_ba_reduce_buffer_doubles(h) doubles, used instead of an internal one (so the caller can
 * alias it with its own tensor).
 * Without a caller buffer the big exchange ships only what the dense solver reads: of row r of [S; bs^T] (r = 0 .. 3P)
 * the columns [0, 32 (r / 32 + 1)) - the lower-triangular 32 x 32 tiles and the rhs row - packed back to back;
 * se2gpu_ba_exchange_row gives the offset and length of a row in that packed buffer (host function, no device needed) and
 * se2gpu_ba_exchange_doubles(P) its total size.  With a caller buffer the rows 0 .. 3P of the rectangle are reduced in
 * place. */
typedef int (*se2gpu_allreduce_fn)(void* dev_ptr, size_t count_doubles, void* hip_stream, void* user);
size_t se2gpu_ba_reduce_buffer_doubles(se2gpu_ba* h, int P);
size_t se2gpu_ba_exchange_doubles(int P);
/* The same for an initialised handle: when the library re-orders the poses for the dense solve (nested dissection, padded
 * partitions - every rank of a sharded run chooses the same order from the merged block pattern, one extra small
 * all-reduce inside initialize) the packed exchange has the rows of THAT system; equal to se2gpu_ba_exchange_doubles(P)
 * in the natural order.  0 before initialize. */
size_t se2gpu_ba_exchange_doubles_h(const se2gpu_ba* h);
int se2gpu_ba_exchange_row(int row, size_t* offset, int* length);
int se2gpu_ba_set_allreduce(se2gpu_ba* h, se2gpu_allreduce_fn fn, void* user, void* buffer);
/* This handle holds landmark shard `rank` of `world` (rank 0 owns the odometry edges and the
 * lambda*I / fixed-pose identity terms, which must enter the sum exactly once). */
int se2gpu_ba_set_shard(se2gpu_ba* h, int rank, int world);

/* Native RCCL communicator (one process per GPU).  The library dlopen()s the system librccl.so.1 - it is NOT a link
 * dependency of single-GPU users - and calls ncclAllReduce(ncclDouble, ncclSum) in place on the handle's stream.
 * Rendezvous of the 128-byte ncclUniqueId is the caller's business (bench.py broadcasts it with torch.distributed
 * over gloo; a ROS deployment would use its own transport).  se2gpu_ba_set_comm = set_shard(rank, world) + an internal
 * all-reduce through this communicator. */
typedef struct se2gpu_comm se2gpu_comm;
int se2gpu_comm_unique_id(uint8_t id_out[128]);                       /* ncclGetUniqueId (rank 0) */
int se2gpu_comm_create(const uint8_t id[128], int rank, int world, se2gpu_comm** out);  /* ncclCommInitRank */
int se2gpu_comm_count(se2gpu_comm* c, int* nranks);                      /* ncclCommCount */
void se2gpu_comm_destroy(se2gpu_comm* c);
int se2gpu_comm_allreduce_sum_f64(se2gpu_comm* c, void* dev_ptr, size_t count, void* hip_stream);
int se2gpu_ba_set_comm(se2gpu_ba* h, se2gpu_comm* c);

/* Host-side landmark partition ("sharded by keyframe window"): owner[l] in [0, world) for the
 * L landmarks of a graph given by its edge lists.  Pure host code (no device needed). */
int se2gpu_ba_shard_landmarks(int L, int E, const int32_t* e_kf, const int32_t* e_lm, int world, int32_t* owner);

/* ------------------------------------------------------------------------------------------
 * Timing helpers for bench.py: HIP events on the stream the kernels are launched on.
 * ------------------------------------------------------------------------------------------ */
typedef struct se2gpu_timer se2gpu_timer;
int se2gpu_timer_create(se2gpu_timer** out);
void se2gpu_timer_destroy(se2gpu_timer* t);
int se2gpu_timer_start(se2gpu_timer* t, void* hip_stream);
int se2gpu_timer_stop(se2gpu_timer* t, void* hip_stream);
int se2gpu_timer_elapsed_ms(se2gpu_timer* t, float* ms);  /* synchronises on the stop event */
void* se2gpu_orb_stream(se2gpu_orb* h);
void* se2gpu_matcher_stream(se2gpu_matcher* h);
void* se2gpu_ba_stream(se2gpu_ba* h);
/* Per-kernel accumulated device time (ms) / launch count since the last reset, measured with HIP
 * events around every launch when profiling is enabled on the handle (adds sync overhead). */
int se2gpu_ba_profile(se2gpu_ba* h, int enable);
int se2gpu_ba_profile_get(se2gpu_ba* h, int idx, const char** name, double* ms, int64_t* launches);
/* FAST score kernel selection (SE2GPU_ORB_SCORE = auto | dense | sparse, default auto): info[0] = kernel the next batch
 * would use (0 = every pixel, 1 = compass pre-test + candidates only), info[1] = last measured candidate density * 1e6, -1
 * before the first measurement.  Both kernels produce the same plane. */
int se2gpu_orb_score_kernel(se2gpu_orb* h, int info[2]);
int se2gpu_orb_profile(se2gpu_orb* h, int enable);
int se2gpu_orb_profile_get(se2gpu_orb* h, int idx, const char** name, double* ms, int64_t* launches);

/* Raw device memory helpers so a non-torch caller (and the tests) can stage device buffers. */
int se2gpu_malloc(void** dev_ptr, size_t bytes);
int se2gpu_free(void* dev_ptr);
int se2gpu_memcpy_h2d(void* dst, const void* src, size_t bytes);
int se2gpu_memcpy_d2h(void* dst, const void* src, size_t bytes);
int se2gpu_device_synchronize(void);
int se2gpu_set_device(int ordinal);
/* Streaming helpers (the camera thread's upload path: pinned staging, asynchronous copies on a stream of its own,
 * cross-stream ordering through the stop event of a timer): what bench.py's `orb.streaming` leg is built from. */
int se2gpu_host_alloc(void** host_ptr, size_t bytes);   /* pinned */
int se2gpu_host_free(void* host_ptr);
int se2gpu_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* hip_stream);
int se2gpu_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* hip_stream);
int se2gpu_stream_create(void** hip_stream_out);
int se2gpu_stream_destroy(void* hip_stream);
int se2gpu_stream_synchronize(void* hip_stream);
int se2gpu_timer_stream_wait(se2gpu_timer* t, void* hip_stream);  /* hip_stream waits for the timer's stop event */

#ifdef __cplusplus
}
#endif
#endif /* SE2GPU_H */
