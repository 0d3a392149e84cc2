/*
 * orbfe.h -- C-ABI of the MI355X-native ORB front end (liborbfe.so).
 *
 * This is the drop-in boundary for ORB-SLAM2's data-parallel front end.  The
 * reference has no FFI layer: the boundary is the C++ class interface of
 * ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-111) and
 * ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-106) compiled into
 * libORB_SLAM2.so.  include/ORBextractor.h and include/ORBmatcher.h of THIS
 * repository re-declare those two classes with the reference signatures on
 * top of the entry points below, so Frame / Tracking / LocalMapping /
 * LoopClosing link unchanged.
 *
 * Conventions: plain pointers and sizes only; every function returns
 * ORBFE_OK (0) or a negative orbfe_status and never throws;
 * orbfe_last_error() gives a thread-local message.  "host" pointers are
 * ordinary memory, "dev" pointers are HIP device memory of the handle's GPU.
 * All kernels are hand-written HIP for gfx950; there is no CPU fallback: if
 * no GPU / code object is available every compute entry point fails with
 * ORBFE_ERR_DEVICE.
 */
#ifndef ORBFE_H_
#define ORBFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum orbfe_status {
    ORBFE_OK = 0,
    ORBFE_ERR_ARG = -1,      /* bad argument (null pointer, size, alignment) */
    ORBFE_ERR_DEVICE = -2,   /* HIP error / no device / kernel image missing */
    ORBFE_ERR_CAPACITY = -3, /* caller buffer or internal capacity too small */
    ORBFE_ERR_UNSUPPORTED = -4
} orbfe_status;

/* Constructor arguments of ORBextractor (include/ORBextractor.h:51-52; values per dataset in
 * Examples/Stereo/KITTI00-02.yaml:38-51 etc.). */
typedef struct orbfe_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} orbfe_params;

/* Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id = 28 bytes), the element
 * type of the `keypoints` output of ORBextractor::operator() (include/ORBextractor.h:59-61). */
typedef struct orbfe_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbfe_keypoint;

typedef struct orbfe_extractor orbfe_extractor;

const char *orbfe_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int orbfe_device_count(void);

/* ------------------------------------------------------------------ ORBextractor ------------------------- */

/* replaces ORBextractor::ORBextractor (src/ORBextractor.cc:410-470): scale tables, per-level quotas, umax, pattern.
 * One handle = one extractor object = one HIP stream + workspace on `device`.  Handles are not re-entrant; distinct
 * handles may be used from distinct threads (src/Frame.cc:78-81 runs two extractors concurrently). */
int orbfe_extractor_create(const orbfe_params *params, int device, orbfe_extractor **out);
int orbfe_extractor_destroy(orbfe_extractor *e);

/* replaces the getters GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:63-83) and exposes mnFeaturesPerLevel / umax (:103-105).  Any pointer may be NULL.
 * scale..inv_sigma2 and features_per_level hold nlevels entries, umax 16. */
int orbfe_get_tables(const orbfe_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                     int32_t *features_per_level, int32_t *umax);

/* replaces ORBextractor::operator() (src/ORBextractor.cc:1043-1105) for one host image.
 * img: 8-bit single channel, h rows of w pixels, `stride` bytes between rows (any stride, cv::Mat::step).
 * kps[cap], desc32[cap*32]: host outputs, level-major order exactly as the reference; *n_out = keypoint count.
 * An empty image (img NULL or w,h <= 0) returns ORBFE_OK with *n_out = 0 and outputs untouched (:1046-1047).
 * cap too small -> ORBFE_ERR_CAPACITY with *n_out = required count. */
int orbfe_extract(orbfe_extractor *e, const uint8_t *img, int w, int h, int stride, orbfe_keypoint *kps, int cap,
                  uint8_t *desc32, int *n_out);

/* Batched form of the same operator for independent frames that are ALREADY RESIDENT in device memory
 * (BASELINE.json config "batched offline extraction"; no counterpart call in the reference, which extracts one
 * frame per call at src/Frame.cc:247-253).  Frame f starts at dev_imgs + f*frame_stride, rows `stride` bytes
 * apart; dev_imgs, stride and frame_stride must be multiples of 16.
 * Outputs (device): dev_kps[n_frames][cap], dev_desc[n_frames][cap][32], dev_counts[n_frames].
 * The call is asynchronous on the handle's stream unless `sync` is non-zero.  If a frame yields more than `cap`
 * keypoints its count is still reported in full and only the first `cap` entries are written.
 * Asynchronous calls: the arguments (input frames and output buffers) must stay valid and unchanged until orbfe_synchronize
 * returns.  Any number of batches may be queued; if a frame of one of them overflows the running FAST kernel's per-strip
 * survivor list (dense noise -- the reference's std::vector just grows, src/ORBextractor.cc:779), orbfe_synchronize runs ALL
 * batches queued since the last synchronisation again, in order, with the large-list kernel, so the outputs are the
 * reference's whatever the content.  The handle returns to the normal kernel by itself after a number of batches.  After 4096
 * queued calls the library synchronises by itself (same status handling), so its record of them stays bounded. */
int orbfe_extract_batch(orbfe_extractor *e, const uint8_t *dev_imgs, int n_frames, int w, int h, int stride,
                        size_t frame_stride, orbfe_keypoint *dev_kps, uint8_t *dev_desc, int32_t *dev_counts, int cap,
                        int sync);
int orbfe_synchronize(orbfe_extractor *e);
/* The same operator over a whole sequence held in HOST memory (BASELINE.json config "batched offline extraction of 4541
 * frames"; the reference's drivers feed one frame per TrackStereo call, Examples/Stereo/stereo_kitti.cc:66-108).  Frame f
 * starts at host_imgs + f*frame_stride; stride and frame_stride must be multiples of 16.  The sequence is processed in chunks
 * of chunk_frames (<= 0: 256) frames with two chunks in flight: the copy-in of the next chunk and the copy-out of the
 * previous one run on their own streams under the kernels of the current chunk.  Outputs in host memory:
 * host_kps[n_frames][cap], host_desc[n_frames][cap][32], host_counts[n_frames] (a frame with more than cap keypoints
 * reports its full count and holds the first cap).  Synchronous.  Pinned host buffers (hipHostMalloc / hipHostRegister)
 * let the copies run at PCIe rate and overlap; pageable buffers work but are staged by the runtime. */
int orbfe_extract_sequence(orbfe_extractor *e, const uint8_t *host_imgs, int n_frames, int w, int h, int stride,
                           size_t frame_stride, orbfe_keypoint *host_kps, uint8_t *host_desc, int32_t *host_counts, int cap,
                           int chunk_frames);

/* replaces the public member std::vector<cv::Mat> mvImagePyramid (include/ORBextractor.h:85; read by
 * src/Frame.cc:473,563,575,580).  Copies level `level` of frame `frame` of the LAST call into host memory laid out
 * like the reference's buffer: (w+38) x (h+38) bytes, row pitch w+38, 19-px BORDER_REFLECT_101 frame
 * (src/ORBextractor.cc:1113-1128); the ROI the reference exposes starts at dst + 19*(w+38) + 19.
 * dst may be NULL to query the size only. */
int orbfe_pyramid_level(orbfe_extractor *e, int frame, int level, uint8_t *dst, size_t dst_bytes, int *w, int *h);
/* All levels of frame `frame` of the last call at once, each into its own buffer laid out as above (dst[l] holds
 * (w_l+38) x (h_l+38) bytes; sizes from orbfe_pyramid_level with dst = NULL): two device-to-host copies into pinned memory
 * and one wait for the whole pyramid instead of a copy and a wait per level.  This is what refreshes mvImagePyramid after
 * every operator() in adapters/ORBextractor_orbfe.cc. */
int orbfe_pyramid_mirror(orbfe_extractor *e, int frame, uint8_t *const *dst, const size_t *dst_bytes, int nlevels);

/* Stage outputs of the last call, for stage-wise parity tests (no counterpart in the reference; its stages are
 * private members, include/ORBextractor.h:87-95).  what: 0 = pyramid level ROI (u8 w*h), 1 = blurred level (u8 w*h),
 * 2 = FAST candidates in reference order as float triples (x, y, response), x/y relative to minBorder (16),
 * 3 = keypoints kept by DistributeOctTree for the level as float quads (x, y, response, angle), level coordinates.
 * Returns the number of bytes the stage holds in *bytes; copies min(*bytes, dst_bytes). */
int orbfe_debug_stage(orbfe_extractor *e, int frame, int level, int what, void *dst, size_t dst_bytes, size_t *bytes);

/* The rotation computeOrbDescriptor derives from a keypoint angle (src/ORBextractor.cc:111-113: angle * factorPI in
 * float, then cosf / sinf), evaluated alone on the device by the function the descriptor kernel uses: cos_sin[2*i] = a,
 * cos_sin[2*i+1] = b for angles_deg[i].  Host pointers; synchronous.  For the known-answer test over the angles where a
 * faster device routine used to differ from libm (tests/golden/sincos_census.json). */
int orbfe_debug_rotation(orbfe_extractor *e, const float *angles_deg, int n, float *cos_sin);
/* The FAST kernel variant the handle's next batch will run: 1 = dense (default), 2 = dense with the large per-strip
 * survivor list (entered after an overflow, left again after 16, 32, ... batches).  For the tests of that recovery; no counterpart
 * in the reference. */
int orbfe_debug_fast_variant(const orbfe_extractor *e);

/* Per-stage device timing of the extractor, taken with HIP events recorded on the handle's stream around each
 * stage's kernel launches inside orbfe_extract / orbfe_extract_batch (no counterpart in the reference, whose only
 * timing is the wall clock around TrackStereo in Examples/Stereo/stereo_kitti.cc:80-97).  Enabling resets the sums.
 * Recording never blocks the stream; orbfe_get_profile drains the stream and folds the recorded events into the sums.
 * enable: 0 = off, 1 = every stage (six events per call), 2 = the FAST kernel only (two events per call, for timing a
 * production-rate run with the least disturbance). */
enum { ORBFE_STAGE_PYRAMID = 0, ORBFE_STAGE_FAST = 1, ORBFE_STAGE_BLUR = 2, ORBFE_STAGE_DESCRIBE = 3,
       ORBFE_STAGE_QUADTREE = 4, ORBFE_STAGE_COUNT = 5 };
typedef struct orbfe_profile {
    double stage_ms[ORBFE_STAGE_COUNT];       /* summed event time per stage */
    int64_t stage_launches[ORBFE_STAGE_COUNT]; /* kernel launches behind that sum */
    int64_t frames;                            /* frames processed while profiling */
    int64_t candidates;                        /* FAST candidates emitted */
    int64_t keypoints;                         /* keypoints kept */
} orbfe_profile;
int orbfe_set_profiling(orbfe_extractor *e, int enable);
/* Where DistributeOctTree (src/ORBextractor.cc:539-763) runs.  DEVICE (default): one workgroup per (frame, level),
 * nothing leaves HBM between FAST and the descriptors.  HOST: candidates are copied out, culled on host threads (one
 * (frame, level) task each) and the kept keypoints copied back -- the arrangement BASELINE.json's north_star starts
 * from; kept for A/B measurement and used automatically when a level's quota exceeds the device kernel's node table
 * (nfeatures beyond ~4700 at scale 1.2).  Both give identical results. */
enum { ORBFE_QUADTREE_DEVICE = 0, ORBFE_QUADTREE_HOST = 1 };
int orbfe_set_quadtree_mode(orbfe_extractor *e, int mode);
int orbfe_get_profile(orbfe_extractor *e, orbfe_profile *out);

/* ------------------------------------------------------------------ ORBmatcher --------------------------- */

/* replaces ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663) and its twin DBoW2::FORB::distance
 * (Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101): Hamming distance of two 32-byte descriptors, host side (the callers pass
 * single cv::Mat rows; a device launch per pair would be absurd). */
int orbfe_hamming(const uint8_t *a32, const uint8_t *b32);

typedef struct orbfe_matcher orbfe_matcher;
/* A matcher context = HIP stream + scratch on `device`.  ORBmatcher objects are stateless temporaries used from three
 * threads (include/ORBmatcher.h:100-101); contexts are cheap, create one per thread. */
int orbfe_matcher_create(int device, orbfe_matcher **out);
int orbfe_matcher_destroy(orbfe_matcher *m);

/* Brute-force best / second-best Hamming search: the inner loops of ORBmatcher::SearchByBoW
 * (src/ORBmatcher.cc:205-226, 570-596) over a whole reference set, and the 1M-descriptor relocalisation config.
 * Update rule as in the reference: `if d<best {second=best; best=d; idx=j} else if d<second {second=d}` scanning
 * j ascending from best=second=256: the first minimum wins, second may equal best.  nr == 0 gives (256, -1, 256).
 * All pointers are device memory; q[nq][32], r[nr][32]; outputs int32[nq].  idx_base is added to every index
 * (global numbering of a database shard). */
int orbfe_match_knn2(orbfe_matcher *m, const uint8_t *dev_q, int nq, const uint8_t *dev_r, int nr, int32_t idx_base,
                     int32_t *dev_best_dist, int32_t *dev_best_idx, int32_t *dev_second_dist, int sync);

/* orbfe_match_knn2 for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that must
 * not reach hipMalloc / hipFree -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same results bit for bit, same kernels; the per-chunk and per-group partial results live in a
 * file-scope device array of ORBFE_KNN2_NOALLOC_SCRATCH_BYTES (24 MiB: 20 MiB of (best, second) key pairs + 4 MiB of group
 * triples) that every GPU the library runs on holds until the process ends, instead of the matcher's grow-only buffers.
 * Its device address is resolved once per device in orbfe_matcher_create.  The call allocates and frees nothing.
 * Limits: nq <= ORBFE_KNN2_NOALLOC_MAX_QUERIES, and a query / database size whose launch plan needs more scratch than the
 * array holds is refused with ORBFE_ERR_CAPACITY and a message that names both sizes (2000 queries x 1 M references, the
 * BASELINE.json configuration, needs 17 MiB; any database fits up to 10240 queries; beyond that the sampled pre-pass of
 * databases >= 32768 references may not).
 * All calls on one GPU share that GPU's array, whatever matcher or thread they come from: calls that overlap in time on one
 * GPU race; serialise them (one stream, or an event between them).  Nothing in the array has to start at zero -- every entry
 * a call reads was written by that call -- so the contents a previous call left behind are harmless. */
#define ORBFE_KNN2_NOALLOC_MAX_QUERIES 65536
#define ORBFE_KNN2_NOALLOC_SCRATCH_BYTES (24u << 20)
int orbfe_match_knn2_noalloc(orbfe_matcher *m, const uint8_t *dev_q, int nq, const uint8_t *dev_r, int nr, int32_t idx_base,
                             int32_t *dev_best_dist, int32_t *dev_best_idx, int32_t *dev_second_dist, int sync);

/* Candidate-list form: the inner loops of SearchByProjection / Fuse / SearchBySim3 / SearchForInitialization
 * (src/ORBmatcher.cc:83-115, 372-392, 436-461, 1400-1424, ...), where the host produces for query i the candidate
 * rows cand_idx[cand_off[i] .. cand_off[i+1]) of r (GetFeaturesInArea, src/Frame.cc:327-380).  Same update rule,
 * candidates scanned in list order.  Device pointers; cand_off has nq+1 entries. */
int orbfe_match_csr(orbfe_matcher *m, const uint8_t *dev_q, int nq, const uint8_t *dev_r, int nr,
                    const int32_t *dev_cand_off, const int32_t *dev_cand_idx, int32_t *dev_best_dist,
                    int32_t *dev_best_idx, int32_t *dev_second_dist, int sync);

/* replaces the Hamming row-band search of Frame::ComputeStereoMatches (src/Frame.cc:476-549): for every left keypoint
 * the right keypoint with the smallest distance among those whose row band [floor(y-r), ceil(y+r)], r = 2*scale[octave],
 * contains (int)vL, whose octave is within +-1 and whose uR lies in [uL-max_d, uL-min_d]; strict '<' from
 * best = TH_HIGH = 100, lowest right index wins ties.  Outputs best_dist[nl] (100 when nothing beat it) and
 * best_idx[nl] (0 in that case, as the reference's bestIdxR).  Device pointers; kps are orbfe_keypoint arrays. */
int orbfe_stereo_rowband(orbfe_matcher *m, const orbfe_keypoint *dev_kl, const uint8_t *dev_dl, int nl,
                         const orbfe_keypoint *dev_kr, const uint8_t *dev_dr, int nr, const float *host_scale_factors,
                         int nlevels, int nrows, float min_d, float max_d, int32_t *dev_best_dist, int32_t *dev_best_idx,
                         int sync);
/* replaces Frame::ComputeStereoMatches as a whole (src/Frame.cc:466-640): the row-band search above with
 * minD = 0, maxD = mbf / mb (:496-498), then for every match below thOrbDist = 75 the 11x11 centre-subtracted SAD over
 * +-5 px on pyramid level octL (:551-592), the parabola (:598-605), the disparity gate (:610-622) and the removal of
 * matches whose SAD is >= 1.5*1.4*median (:626-639).  The un-blurred pyramids are read in place from the two
 * extractors' last call (frames `left_frame` / `right_frame` of their last batch; `left` and `right` may be the same
 * handle after a 2-frame batch); the level-0 images are the buffers that call was given and must still be alive.  The
 * call is ordered after both extractors' streams.  Outputs mvuRight[nl], mvDepth[nl] (-1 where there is no match), on
 * the device.  With no accepted match the reference reads past an empty vector (:627); here nothing is gated. */
int orbfe_stereo_match(orbfe_matcher *m, orbfe_extractor *left, int left_frame, orbfe_extractor *right, int right_frame,
                       const orbfe_keypoint *dev_kl, const uint8_t *dev_dl, int nl, const orbfe_keypoint *dev_kr,
                       const uint8_t *dev_dr, int nr, float mbf, float mb, float *dev_uright, float *dev_depth, int sync);
/* The tree descent of DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform (reference
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259, called per feature from :1127-1200 by Frame::ComputeBoW,
 * src/Frame.cc:395-402, and KeyFrame::ComputeBoW).  The vocabulary is handed over as flat host arrays: node 0 is the
 * root, the children of node i are child_idx[child_off[i] .. child_off[i+1]) in m_nodes[i].children order (ids larger
 * than i), node_desc holds 32 bytes per node, a node without children is a leaf; depth_L = m_L. */
typedef struct orbfe_vocab orbfe_vocab;
int orbfe_vocab_create(int device, int n_nodes, const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc,
                       int depth_L, orbfe_vocab **out);
int orbfe_vocab_destroy(orbfe_vocab *v);
/* per feature i: dev_leaf_node[i] = the leaf the descent ends in (the host maps it to word id and weight),
 * dev_nid[i] = the node passed at level depth_L - levelsup (0 when that level is <= 0; also 0 when the walk ends above
 * that level, where the reference leaves the value unset).  FORB::distance, strict '<': the first minimum wins. */
int orbfe_bow_transform(orbfe_matcher *m, const orbfe_vocab *v, const uint8_t *dev_desc, int n, int levelsup,
                        int32_t *dev_leaf_node, int32_t *dev_nid, int sync);
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307) for many map points in one call: point p owns the
 * descriptor rows dev_off[p] .. dev_off[p+1] of dev_desc (the rows of its observations, in the reference's
 * std::map<KeyFrame*, size_t> iteration order); dev_best[p] = index inside the point of the descriptor with the least
 * median Hamming distance to the others (median = sorted row [(int)(0.5*(N-1))], the self distance 0 included; the first
 * minimum wins), -1 for a point without descriptors.  max_per_point = the largest row count (<= 65535). */
int orbfe_distinctive_descriptors(orbfe_matcher *m, const uint8_t *dev_desc, const int32_t *dev_off, int n_points,
                                  int max_per_point, int32_t *dev_best, int sync);
/* The frame grid and the windowed search on it: Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:230-246, 382-392),
 * Frame::GetFeaturesInArea (:327-380) and the candidate loop of ORBmatcher::SearchByProjection(Frame&,
 * const vector<MapPoint*>&, th) (src/ORBmatcher.cc:71-115), for a batch of projected map points.  The grid is built on
 * the host from the Frame members (undistorted keypoint coordinates and octaves, descriptors, mvuRight or NULL,
 * mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv) exactly as the reference builds mGrid, and kept on the
 * device.  Host pointers throughout: these calls sit inside host loops of the reference.  A grid is read-only once built:
 * any number of matcher contexts (threads) may search the same grid at the same time; the staging of a search belongs to
 * the matcher context. */
typedef struct orbfe_frame_grid orbfe_frame_grid;
typedef struct orbfe_area_query {
    float x, y, r;                /* GetFeaturesInArea(x, y, r, min_level, max_level) */
    int32_t min_level, max_level;
    float proj_xr, er_max;        /* :87-92: skip idx when mvuRight[idx] > 0 and |proj_xr - mvuRight[idx]| > er_max; off if er_max < 0 */
    int32_t pad;
} orbfe_area_query;
typedef struct orbfe_area_result {
    int32_t best_dist, best_idx, best_level, best_dist2, best_level2;   /* 256, -1, -1, 256, -1 when nothing was compared */
    int32_t n_candidates;                                               /* vIndices.size() */
    int32_t best_idx2;   /* the keypoint that holds best_dist2 / best_level2 (-1 if none): the reference keeps no such local; it
                          * lets the caller tell whether an association made later in its loop touches this result */
    int32_t pad1;
} orbfe_area_result;
int orbfe_frame_grid_create(int device, const float *kx, const float *ky, const int32_t *koct, const uint8_t *kdesc,
                            const float *uright, int n, float min_x, float min_y, float width_inv, float height_inv,
                            orbfe_frame_grid **out);
int orbfe_frame_grid_destroy(orbfe_frame_grid *g);
/* Refill an existing grid with another frame's keypoints (same arguments as create): the device block is kept and only grows,
 * the upload is one copy from the grid's pinned block on the matcher's stream, complete on return.  For a pool of grids that is
 * recycled frame after frame (adapters/ORBmatcher_orbfe.cc keeps one keyed by frame id) instead of a hipMalloc / hipFree pair
 * per Frame.  The caller must make sure no search on `g` is in flight.  No counterpart in the reference (AssignFeaturesToGrid
 * refills the Frame's own std::vector cells, src/Frame.cc:230-246). */
int orbfe_frame_grid_update(orbfe_matcher *m, orbfe_frame_grid *g, const float *kx, const float *ky, const int32_t *koct,
                            const uint8_t *kdesc, const float *uright, int n, float min_x, float min_y, float width_inv,
                            float height_inv);
/* blocked (n bytes or NULL): keypoints the loop skips (:83-85: already associated with an observed MapPoint when the call
 * is made; associations made DURING the reference's loop are the caller's to replay).  Synchronous. */
int orbfe_search_area(orbfe_matcher *m, orbfe_frame_grid *g, const orbfe_area_query *queries, const uint8_t *query_desc, int nq,
                      const uint8_t *blocked, orbfe_area_result *results);
/* The same windowed search with the per-candidate tests of the other guided searches (KeyFrame::GetFeaturesInArea,
 * src/KeyFrame.cc:569-608, is Frame::GetFeaturesInArea with min_level = max_level = -1 on the same 64 x 48 grid, so one grid
 * type serves Frame and KeyFrame):
 *   blocked           keypoints the loop skips: vpMatched[idx] (src/ORBmatcher.cc:375), CurrentFrame.mvpMapPoints[i2] (:1541)
 *   dist_limit        SearchForInitialization (:444): skip idx when dist_limit[idx] <= dist (vMatchedDistance), n ints or NULL
 *   inv_level_sigma2  Fuse(KeyFrame*, vpMapPoints, th) (:914-938): with u = query x, v = query y, ur = proj_xr, skip idx when
 *                     (ex^2 + ey^2 [+ er^2 if mvuRight[idx] >= 0]) * inv_level_sigma2[octave] > chi2_mono [chi2_stereo]
 *                     (5.99 / 7.8); nlevels floats or NULL = no gate.  float products and sums individually rounded.
 * The level test of those loops (kpLevel in [nPredictedLevel-1, nPredictedLevel], :380, :911, :1067, :1207, :1287) is the
 * query's min_level / max_level; set er_max < 0 where the loop has no right-coordinate test.  Results as orbfe_search_area
 * (loops that start from INT_MAX instead of 256, :432, :1060, :1199, see 256 where nothing was compared: best_idx = -1). */
typedef struct orbfe_area_options {
    const uint8_t *blocked;
    const int32_t *dist_limit;
    const float *inv_level_sigma2;
    int32_t nlevels;
    int32_t has_query_origin;          /* non-zero: the cell range of a window is computed from (query_min_x, query_min_y) */
    double chi2_mono, chi2_stereo;
    float query_min_x, query_min_y;    /* KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:574-586) subtracts the KeyFrame's own
                                        * int mnMinX / mnMinY (include/KeyFrame.h:185-186) while the cells it walks were
                                        * assigned with the Frame's float bounds (src/Frame.cc:384-385) */
} orbfe_area_options;
int orbfe_search_area_ex(orbfe_matcher *m, orbfe_frame_grid *g, const orbfe_area_query *queries, const uint8_t *query_desc, int nq,
                         const orbfe_area_options *opt, orbfe_area_result *results);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-288) and SearchByBoW(KeyFrame*, KeyFrame*, ...)
 * (:522-655) without the rotation histogram: for every vocabulary node present in both feature vectors (the caller's
 * merge walk, :180-264, gives node_off / node_idx for both sides in FeatureVector order), each side-1 feature with
 * valid1 != 0 (MapPoint present and not bad; NULL = all valid) takes the best side-2 feature of the node by the loop at
 * :205-226 -- side-2 features matched earlier in the call or listed in blocked2 (:574-580; NULL = none) are skipped --
 * and is matched if bestDist1 <= max_dist (TH_LOW, or TH_LOW - 1 for the strict test at :598) and
 * (float)bestDist1 < nn_ratio * (float)bestDist2.  match1[i] = matched side-2 index or -1.  The sequential dependency
 * (:209, :576) stays inside a node because a feature belongs to one node; the device walks each node in order, so the
 * result is the reference loop's.  Host pointers, synchronous.  include/ORBmatcher_orbfe.h adds the histogram. */
int orbfe_search_by_bow(orbfe_matcher *m, const uint8_t *desc1, const uint8_t *valid1, int n1, const uint8_t *desc2,
                        const uint8_t *blocked2, int n2, const int32_t *node_off1, const int32_t *node_idx1,
                        const int32_t *node_off2, const int32_t *node_idx2, int n_nodes, int max_dist, float nn_ratio,
                        int32_t *match1);
/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823) without the rotation histogram: shared nodes as in
 * orbfe_search_by_bow; has_mp1 / has_mp2: GetMapPoint(idx) != NULL (:699-703, :722-726); uright1 / uright2: mvuRight (a value
 * >= 0 makes the keypoint a stereo one, :705, :728); xy1 / xy2: mvKeysUn[*].pt as (x, y) pairs; oct2: mvKeysUn[*].octave of
 * side 2; F12: the fundamental matrix, row-major 3 x 3 floats; ex, ey: the epipole of :664-670; scale_factors /
 * level_sigma2: pKF2->mvScaleFactors / mvLevelSigma2 (nlevels <= 16 entries).  A side-1 feature takes the side-2 feature of its
 * node with the least distance <= TH_LOW among those passing the epipole test (:743-749) and CheckDistEpipolarLine
 * (:140-157); of equal distances the last in list order wins (`dist > bestDist` at :738).  match1[idx1] = idx2 or -1.
 * Host pointers, synchronous.  include/ORBmatcher_orbfe.h adds the histogram and vMatchedPairs. */
int orbfe_search_for_triangulation(orbfe_matcher *m, const uint8_t *desc1, const uint8_t *has_mp1, const float *uright1,
                                   const float *xy1, int n1, const uint8_t *desc2, const uint8_t *has_mp2, const float *uright2,
                                   const float *xy2, const int32_t *oct2, int n2, const int32_t *node_off1,
                                   const int32_t *node_idx1, const int32_t *node_off2, const int32_t *node_idx2, int n_nodes,
                                   const float *F12, float ex, float ey, const float *scale_factors, const float *level_sigma2,
                                   int nlevels, int only_stereo, int32_t *match1);
/* Host-pointer forms of three device-pointer calls above, for use from inside the reference's host code (Frame::ComputeBoW,
 * Frame::ComputeStereoMatches): descriptors are rows of a cv::Mat (`*_stride` bytes apart, cv::Mat::step), keypoints are the
 * std::vector<cv::KeyPoint> storage.  Inputs travel through the matcher context's pinned block in one copy, outputs in one
 * copy back; synchronous; nothing is allocated once the blocks have grown to the frame size. */
int orbfe_bow_transform_host(orbfe_matcher *m, const orbfe_vocab *v, const uint8_t *desc, size_t desc_stride, int n, int levelsup,
                             int32_t *leaf_node, int32_t *nid);
int orbfe_stereo_rowband_host(orbfe_matcher *m, const orbfe_keypoint *kl, const uint8_t *dl, size_t dl_stride, int nl,
                              const orbfe_keypoint *kr, const uint8_t *dr, size_t dr_stride, int nr, const float *scale_factors,
                              int nlevels, int nrows, float min_d, float max_d, int32_t *best_dist, int32_t *best_idx);
int orbfe_stereo_match_host(orbfe_matcher *m, orbfe_extractor *left, int left_frame, orbfe_extractor *right, int right_frame,
                            const orbfe_keypoint *kl, const uint8_t *dl, size_t dl_stride, int nl, const orbfe_keypoint *kr,
                            const uint8_t *dr, size_t dr_stride, int nr, float mbf, float mb, float *uright, float *depth);
int orbfe_matcher_synchronize(orbfe_matcher *m);
/* the HIP stream of a context / extractor as a void* (hipStream_t), so callers can order their own work after it */
void *orbfe_matcher_stream(orbfe_matcher *m);
void *orbfe_extractor_stream(orbfe_extractor *e);

/* Merge of per-shard kNN-2 results for a database sharded over several GPUs: given S shards'
 * (best_dist, best_idx, second_dist) for nq queries laid out [S][nq], produce the result the unsharded scan would
 * give: best = lowest distance then lowest global index; second = second smallest of the multiset of all shards'
 * best and second.  Host pointers (the gathered buffers are 12 bytes per query and shard). */
int orbfe_merge_knn2(const int32_t *best_dist, const int32_t *best_idx, const int32_t *second_dist, int n_shards, int nq,
                     int32_t *out_best_dist, int32_t *out_best_idx, int32_t *out_second_dist);

/* The same search with the exchange inside the library, for a C++ host with one process per GPU (SURVEY.md section 8b / 8e;
 * BASELINE.json configs[4]: 2000 queries against 1 M map-point descriptors sharded over 8 GPUs).  No counterpart in the
 * reference (single process, no collectives).  RCCL over xGMI: ncclBroadcast of the query batch from rank 0 (nq x 32 bytes),
 * the local scan (orbfe_match_knn2 on this rank's shard), ncclAllGather of the per-shard triples (12 bytes per query and
 * rank -- latency-bound, far below a link's bandwidth) and the fold of orbfe_merge_knn2 as a kernel, all on the matcher's
 * stream; every rank ends up with the unsharded answer in device memory.  librccl.so.1 is bound at run time (the copy the
 * process already holds, else the system's); without it these calls fail with ORBFE_ERR_DEVICE and nothing else is affected.
 *   unique_id  rank 0 creates the 128-byte id and ships it to the other ranks by its own means (MPI, a file, a socket, torch)
 *   create     collective over all ranks of the job: ncclCommInitRank on the matcher's device; max_queries sizes the buffers
 *   run        collective; dev_q holds the queries on rank 0 and receives them elsewhere; dev_shard / n_shard / idx_base as in
 *              orbfe_match_knn2 (idx_base = global index of the shard's first descriptor)
 * The exchange has run on hardware with one rank only (tests/test_gpu_match.py); more ranks need more GPUs than the
 * development box has -- see DESIGN.md section 6. */
#define ORBFE_SHARDED_ID_BYTES 128
typedef struct orbfe_knn2_sharded orbfe_knn2_sharded;
/*   available  binds librccl (ORBFE_RCCL_LIB names another library) and does nothing else: ORBFE_OK, or ORBFE_ERR_DEVICE with the
 *              loader's message.  Not collective, touches no GPU: every rank calls it BEFORE any call that can block on the
 *              others, so that a job agrees on a fallback instead of hanging in ncclCommInitRank (bench.py). */
int orbfe_knn2_sharded_available(void);
int orbfe_knn2_sharded_unique_id(uint8_t id[ORBFE_SHARDED_ID_BYTES]);
int orbfe_knn2_sharded_create(orbfe_matcher *m, const uint8_t id[ORBFE_SHARDED_ID_BYTES], int rank, int world, int max_queries,
                              orbfe_knn2_sharded **out);
int orbfe_knn2_sharded_run(orbfe_knn2_sharded *s, uint8_t *dev_q, int nq, const uint8_t *dev_shard, int n_shard, int32_t idx_base,
                           int32_t *dev_best_dist, int32_t *dev_best_idx, int32_t *dev_second_dist, int sync);
int orbfe_knn2_sharded_destroy(orbfe_knn2_sharded *s);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H_ */
