/*
 * orb_oracle.h -- C interface of the CPU ORACLE for the ORB front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under orb_slam2_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (abhineet123/ORB_SLAM2) ships no tests,
 * fixtures or golden vectors and its arithmetic lives in
 * OpenCV, which is neither vendored nor installed here, so the reference
 * cannot be compiled (oracle/_ref is therefore absent).  This oracle restates
 * src/ORBextractor.cc, src/ORBmatcher.cc:1647-1663 and src/Frame.cc:466-640
 * line by line together with the generic C++ path of OpenCV 2.4.11/3.2 for
 * resize / FAST / GaussianBlur / fastAtan2, and is
 * checked against hand-derivable known answers in tests/test_oracle_kat.py.
 */
#ifndef ORB_ORACLE_H_
#define ORB_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout of cv::KeyPoint (7 x 4 bytes) */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* ---- ORBextractor (src/ORBextractor.cc:410-470, 1043-1105) ---- */
orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast);
void orc_destroy(orc_extractor *e);
/* operator(): returns 0, *n_out = number of keypoints (<= cap or -1 if cap too small) */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                orc_keypoint *kps, int cap, uint8_t *desc32, int *n_out);

/* constructor tables */
int orc_nlevels(const orc_extractor *e);
void orc_scale_tables(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2);
void orc_features_per_level(const orc_extractor *e, int *out);
void orc_umax(const orc_extractor *e, int *out16);

/* state left behind by the last orc_extract (introspection for stage-wise parity) */
int orc_level_size(const orc_extractor *e, int level, int *w, int *h);
const uint8_t *orc_level_ptr(const orc_extractor *e, int level);   /* w*h, contiguous (the ROI, no border) */
const uint8_t *orc_blur_ptr(const orc_extractor *e, int level);    /* w*h or NULL if level had 0 keypoints */
/* FAST candidates of a level in reference order; x,y relative to minBorder (16); 3 floats each (x,y,response) */
int orc_level_candidates(const orc_extractor *e, int level, float *xyr, int cap);
/* keypoints kept by DistributeOctTree for a level, list order, level coordinates incl. border; 4 floats (x,y,response,angle) */
int orc_level_selected(const orc_extractor *e, int level, float *xyra, int cap);

/* ---- stand-alone primitives ---- */
/* cv::resize(..., INTER_LINEAR) 8UC1, OpenCV 2.4/3.2 generic path */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride);
/* coefficient tables of that resize: xofs[dw], ialpha[2*dw] (also for y with dh) */
void orc_resize_tables(int ssize, int dsize, int *ofs, int16_t *coef);
/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) 8UC1 pre-3.4.1 fixed-point path */
void orc_gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
void orc_gauss7_kernel(int *k7);
/* cv::FAST(win, kps, threshold, true) (TYPE_9_16); returns count; xyr = (x,y,response) window coords */
int orc_fast9(const uint8_t *win, int w, int h, int stride, int threshold, float *xyr, int cap);
/* the per-level cell loop of ComputeKeyPointsOctTree (src/ORBextractor.cc:771-829) */
int orc_fast_cells(const uint8_t *img, int w, int h, int stride, int ini_th, int min_th, float *xyr, int cap);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t *img, int stride, int cx, int cy, const int *umax16);
void orc_descriptor(const uint8_t *blurred, int stride, int cx, int cy, float angle_deg, uint8_t *desc32);
/* DistributeOctTree (src/ORBextractor.cc:539-763); xyr in/out 3 floats each; returns kept count */
int orc_distribute_octree(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, float *out_xyr, int cap);
int orc_cv_round(double v);

/* ---- ORBmatcher (src/ORBmatcher.cc:1647-1663) ---- */
int orc_hamming(const uint8_t *a32, const uint8_t *b32);
/* brute-force best / second-best (update rule of src/ORBmatcher.cc:216-225); idx = -1 if nr == 0 */
void orc_knn2(const uint8_t *q, int nq, const uint8_t *r, int nr, int32_t *best_dist, int32_t *best_idx, int32_t *second_dist);
/* candidate-list form: for query i the candidates are cand_idx[cand_off[i] .. cand_off[i+1]) */
void orc_match_csr(const uint8_t *q, int nq, const uint8_t *r, const int32_t *cand_off, const int32_t *cand_idx,
                   int32_t *best_dist, int32_t *best_idx, int32_t *second_dist);

/* ---- Frame::ComputeStereoMatches (src/Frame.cc:466-640) ---- */
/* Hamming row-band search only (src/Frame.cc:476-549): best_dist[iL] (TH_HIGH=100 if none), best_idx[iL] */
void orc_stereo_rowband(const orc_keypoint *kl, const uint8_t *dl, int nl, const orc_keypoint *kr, const uint8_t *dr, int nr,
                        const float *scale_factors, int nrows, float min_d, float max_d, int32_t *best_dist, int32_t *best_idx);
/* the whole routine incl. SAD refinement, parabola, median gate: u_right[nl], depth[nl] */
void orc_stereo_full(const orc_extractor *el, const orc_extractor *er, const orc_keypoint *kl, const uint8_t *dl, int nl,
                     const orc_keypoint *kr, const uint8_t *dr, int nr, float bf, float b, float *u_right, float *depth);

/* ---- DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) ----
 * The vocabulary tree as flat arrays: node 0 is the root, the children of node i are child_idx[child_off[i] .. child_off[i+1])
 * in m_nodes[i].children order, node_desc holds 32 bytes per node, a node without children is a leaf (isLeaf()). */
/* tree descent of one feature (:1218-1259): leaf[i] = final node id, nid[i] = node at level L - levelsup (0 when that level
 * is <= 0; the reference leaves *nid unset when the descent ends above that level -- pinned to 0 here) */
void orc_bow_descend(const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc, int L, const uint8_t *desc, int n,
                     int levelsup, int32_t *leaf, int32_t *nid);
/* the whole transform(features, BowVector, FeatureVector, levelsup) (:1127-1200): weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY;
 * norm 0 none, 1 L1, 2 L2 (ScoringObject::mustNormalize).  bow_ids/bow_vals: the BowVector in map order (capacity n);
 * fv_node/fv_feat: the FeatureVector flattened in map order, features of a node in insertion order (capacity n). */
void orc_bow_transform(const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc, const int32_t *word_id,
                       const double *weight, int L, int weighting, int norm, const uint8_t *desc, int n, int levelsup,
                       int32_t *bow_ids, double *bow_vals, int32_t *n_bow, int32_t *fv_node, int32_t *fv_feat, int32_t *n_fv);

/* ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched over map points ----
 * point p owns descriptors desc[off[p] .. off[p+1]); best[p] = index (inside the point) of the descriptor with the least
 * median distance to the others, median = sorted row[(int)(0.5*(N-1))] with the self distance 0 in the row; first minimum
 * wins (strict '<' from INT_MAX); -1 for a point without descriptors (the reference returns early). */
void orc_distinctive(const uint8_t *desc, const int32_t *off, int n_points, int32_t *best);

/* ---- Frame grid (src/Frame.cc:230-246, 327-392) and the windowed best/second search of SearchByProjection ----
 * keypoints: kx, ky (undistorted coordinates), koct (octave).  Grid: 64 x 48 cells (include/Frame.h:37-38). */
typedef struct orc_grid orc_grid;
orc_grid *orc_grid_create(const float *kx, const float *ky, const int32_t *koct, int n, float min_x, float min_y,
                          float width_inv, float height_inv);
void orc_grid_destroy(orc_grid *g);
/* KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:569-608) subtracts the KeyFrame's own int mnMinX / mnMinY from the window while
 * the cells were assigned with the Frame's float bounds: set the window origin apart from the cell origin */
void orc_grid_set_query_origin(orc_grid *g, float min_x, float min_y);
/* Frame::GetFeaturesInArea (:327-380): indices in the reference's order; returns the count (out has room for n) */
int orc_features_in_area(const orc_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out);
/* the candidate loop of ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:71-115) for one
 * query: candidates from GetFeaturesInArea, skip blocked[idx] != 0 (:83-85, a static snapshot), the right-coordinate gate
 * (:87-92) when er_max >= 0, strict '<' best / second with their levels.  res = {bestDist, bestIdx, bestLevel, bestDist2,
 * bestLevel2, number of candidates returned by GetFeaturesInArea} */
void orc_search_area(const orc_grid *g, const uint8_t *kdesc, const float *uright, const uint8_t *blocked, float x, float y,
                     float r, int min_level, int max_level, float proj_xr, float er_max, const uint8_t *qdesc, int32_t res[7]);
/* SearchByBoW loops, ComputeThreeMaxima and the rotation-consistency filter (src/ORBmatcher.cc:159-288, :522-655, :1601-1642) */
void orc_search_by_bow(const uint8_t *desc1, const uint8_t *valid1, int n1, const uint8_t *desc2, const uint8_t *blocked2, int n2,
                       const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2, int n_nodes,
                       int max_dist, float nn_ratio, int32_t *match1);
void orc_three_maxima(const int32_t *sizes, int L, int32_t ind[3]);
int orc_rotation_filter(const float *angle1, const float *angle2, int32_t *match1, int n1);

/* the remaining guided searches from their window search on; see orb_oracle.cpp for the reference lines */
int orc_search_by_projection_sim3(const orc_grid *g, const uint8_t *kdesc, int nq, const float *u, const float *v,
                                  const float *radius, const int32_t *level, const uint8_t *qdesc, uint8_t *matched,
                                  int32_t *match_idx);
int orc_search_by_projection_reloc(const orc_grid *g, const uint8_t *kdesc, const float *kangle, int nq, const float *u,
                                   const float *v, const float *radius, const int32_t *level, const float *qangle,
                                   const uint8_t *qdesc, int orb_dist, int check_orientation, uint8_t *occupied,
                                   int32_t *assigned);
int orc_fuse_select(const orc_grid *g, const uint8_t *kdesc, const float *uright, const float *inv_level_sigma2, int use_chi2,
                    int nq, const float *u, const float *v, const float *ur, const float *radius, const int32_t *level,
                    const uint8_t *qdesc, int32_t *best_idx);
int orc_search_by_sim3(const orc_grid *g1, const uint8_t *kdesc1, const orc_grid *g2, const uint8_t *kdesc2, int nq1,
                       const int32_t *q1_index, const float *u1, const float *v1, const float *radius1, const int32_t *level1,
                       const uint8_t *qdesc1, int nq2, const int32_t *q2_index, const float *u2, const float *v2,
                       const float *radius2, const int32_t *level2, const uint8_t *qdesc2, int32_t *match12);
int orc_search_for_initialization(const orc_grid *g2, const uint8_t *kdesc2, const float *kangle2, int n1, const int32_t *oct1,
                                  const float *angle1, const uint8_t *desc1, float *prev_xy, int window_size, float nn_ratio,
                                  int check_orientation, int32_t *vnMatches12);
int orc_search_for_triangulation(const uint8_t *desc1, const uint8_t *has_mp1, const float *uright1, const float *xy1,
                                 const float *angle1, int n1, const uint8_t *desc2, const uint8_t *has_mp2, const float *uright2,
                                 const float *xy2, const int32_t *oct2, const float *angle2, int n2, const int32_t *off1,
                                 const int32_t *idx1, const int32_t *off2, const int32_t *idx2, int n_nodes, const float *F12,
                                 float ex, float ey, const float *scale_factors, const float *level_sigma2, int only_stereo,
                                 int check_orientation, int32_t *vMatches12);

/* time n_frames calls of orc_extract over `threads` threads (one frame per thread at a time); returns seconds */
double orc_time_extract(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, const uint8_t *imgs, int n_frames,
                        int w, int h, int stride, long frame_stride, int threads, int reps, long *total_kps);
double orc_time_knn2(const uint8_t *q, int nq, const uint8_t *r, int nr, int threads, int swar);

#ifdef __cplusplus
}
#endif
#endif
