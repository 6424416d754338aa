/*
 * mpsfm_hip.h — C ABI of libmpsfm_hip.so, the MI355X (gfx950) replacement for the
 * pyceres / pycolmap native boundary that MP-SfM's bundle adjustment and triangulation
 * numerics cross (reference call sites cited per entry point below; paths are relative
 * to the reference checkout).
 *
 * Conventions
 *   - Plain C, no exceptions, no torch types.  Every entry point returns 0 on success or a
 *     negative MPSFM_E* code; mpsfm_last_error() gives a thread-local message.
 *   - The caller owns every buffer.  The library never retains a host pointer after a call
 *     returns (handles own device copies only).
 *   - All floating point is IEEE double.  Indices are int32 (counts int64).
 *   - Quaternions are stored (x, y, z, w) — Eigen order — as in
 *     mpsfm/sfm/mapper/bundle_adjustment.py:113-122 (pose.rotation.quat).
 *   - The library fails loudly (MPSFM_ENODEVICE) when no gfx950 device is present; there is
 *     no CPU fallback inside it.
 */
#ifndef MPSFM_HIP_H
#define MPSFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPSFM_ABI_VERSION 2

/* ---- error codes -------------------------------------------------------------------- */
#define MPSFM_OK 0
#define MPSFM_EINVAL (-1)     /* malformed problem (index out of range, NULL pointer, ...) */
#define MPSFM_ENODEVICE (-2)  /* no HIP device / not gfx950 */
#define MPSFM_ENOMEM (-3)     /* host or device allocation failed */
#define MPSFM_EHIP (-4)       /* a HIP runtime call failed */
#define MPSFM_ENUMERIC (-5)   /* the initial point could not be evaluated (NaN / z<=0 in a log) */
#define MPSFM_EUNSUPPORTED (-6)
#define MPSFM_ECOMM (-7)      /* the all-reduce hook reported a failure */

/* ---- loss functions ------------------------------------------------------------------
 * pycolmap.LossFunctionType as mapped in mpsfm/sfm/mapper/bundle_adjustment.py:44-48.
 * rho(s), s = squared residual norm (Ceres semantics):
 *   TRIVIAL  rho = s
 *   SOFT_L1  rho = 2 a^2 (sqrt(1 + s/a^2) - 1)
 *   CAUCHY   rho = a^2 log(1 + s/a^2)
 * Every loss is wrapped as Scaled(rho, magnitude): cost = 1/2 * magnitude * rho(s).      */
enum mpsfm_loss_type { MPSFM_LOSS_TRIVIAL = 0, MPSFM_LOSS_SOFT_L1 = 1, MPSFM_LOSS_CAUCHY = 2 };

/* ---- the flat BA problem -------------------------------------------------------------
 * Exactly what Optimizer.__build_problem (bundle_adjustment.py:67-185) gathers before it
 * calls pyceres.solve: the images of the bundle (with gauge flags), the 3-D points their
 * observations reference, one 2-residual PINHOLE reprojection block per observation
 * (pycolmap.create_default_bundle_adjuster, :85-104) and one 1-residual log-depth block per
 * valid depth prior (pycolmap.create_depth_bundle_adjuster, :163-176).                   */
typedef struct mpsfm_ba_problem {
  int32_t n_cams;  /* images (poses) referenced by the observation lists            */
  int32_t n_pts;   /* 3-D points                                                      */
  int32_t n_intr;  /* distinct PINHOLE intrinsics                                     */

  const double* cam_intr;        /* [n_intr][4]  fx fy cx cy; constant (:92-94)      */
  const int32_t* cam_intr_idx;   /* [n_cams]                                          */
  const uint8_t* pose_const;     /* [n_cams] 1: quat and translation constant
                                    (first bundle image :114-116, fix_pose, or an image
                                    outside the bundle that sees a variable point)     */
  int32_t gauge_axis_cam;        /* camera whose translation x is held fixed
                                    (SubsetManifold(3,[0]), :117-121), or -1           */
  const uint8_t* pt_const;       /* [n_pts] 1: point constant                          */

  int64_t n_obs;                 /* reprojection residual blocks                        */
  const int32_t* obs_cam;        /* [n_obs]                                             */
  const int32_t* obs_pt;         /* [n_obs]                                             */
  const double* obs_xy;          /* [n_obs][2] pixel measurement (Point2D.xy)           */
  int32_t reproj_loss_type;      /* mpsfm_loss_type (reference default SOFT_L1, :24)    */
  double reproj_loss_scale;      /* a  = reproj_loss_scale * kp_std (:101)              */
  double reproj_loss_magnitude;  /* k  = 1 / kp_std^2 (:99)                             */

  int64_t n_dobs;                /* log-depth residual blocks (0: reprojection only)    */
  const int32_t* dobs_cam;       /* [n_dobs]                                            */
  const int32_t* dobs_pt;        /* [n_dobs]                                            */
  const double* dobs_depth;      /* [n_dobs] prior depth d sampled at the keypoint      */
  const double* dobs_magnitude;  /* [n_dobs] m = d^2 / clip(var,1e-6) (:161)            */
  const double* dobs_param;      /* [n_dobs] a = mult * rob_std * sqrt(var) / d (:160)  */
  int32_t depth_loss_type;       /* CAUCHY in ba(), TRIVIAL in refine_3d_points()       */
  const double* shift_logscale;  /* [n_cams][2] constant per-image (shift b, log-scale s)
                                    of the prior, NULL = zeros (:83, :178-182)          */
} mpsfm_ba_problem;

/* Parameters the solver updates in place, like Ceres does through the pybind11 views. */
typedef struct mpsfm_ba_state {
  double* cam_quat_xyzw; /* [n_cams][4] */
  double* cam_t;         /* [n_cams][3] */
  double* pts;           /* [n_pts][3]  */
} mpsfm_ba_state;

/* Sum-all-reduce hook for landmark-sharded BA: called with a buffer of `count` doubles that
 * must be replaced by its element-wise sum over all ranks before the hook returns (or, for a
 * device buffer, before later work on `stream`).  `on_device` is 1 when `buf` is device
 * memory of the handle's device.  Return 0 on success. */
typedef int (*mpsfm_allreduce_fn)(void* user, double* buf, int64_t count, int on_device,
                                  void* stream);

/* Solver options.  mpsfm_ba_default_options() fills the pyceres.SolverOptions() defaults
 * that Optimizer.solve leaves untouched (bundle_adjustment.py:285-293). */
typedef struct mpsfm_ba_options {
  int32_t max_num_iterations;               /* 50    */
  double function_tolerance;                /* 1e-6  */
  double gradient_tolerance;                /* 1e-10 */
  double parameter_tolerance;               /* 1e-8  */
  double initial_trust_region_radius;       /* 1e4   */
  double max_trust_region_radius;           /* 1e16  */
  double min_trust_region_radius;           /* 1e-32 */
  double min_relative_decrease;             /* 1e-3  */
  double min_lm_diagonal;                   /* 1e-6  */
  double max_lm_diagonal;                   /* 1e32  */
  int32_t max_num_consecutive_invalid_steps;/* 5     */
  int32_t jacobi_scaling;                   /* 1     */
  int32_t device;                           /* HIP device ordinal                      */
  void* stream;                             /* hipStream_t to run on, NULL = own stream */
  int32_t verbose;                          /* >0: per-iteration line on stderr         */
  mpsfm_allreduce_fn allreduce;             /* NULL: single shard (or the native RCCL communicator below) */
  void* allreduce_user;
  /* landmark sharding over the GPUs of a node: this rank's position ... */
  int32_t world_size;                       /* 0 / 1: single shard; > 1 with the hook: lets the per-rank maxima (gradient
                                               tolerance test) travel in per-rank slots of the summed buffer           */
  int32_t rank;
  /* ... and, with use_rccl = 1, the library's OWN communicator: ncclCommInitRank(world_size, comm_id, rank) at
     mpsfm_ba_create, ncclAllReduce(sum, fp64) on the handle's stream for every exchange (no host callback per LM
     iteration).  comm_id comes from mpsfm_comm_unique_id() on one rank; the caller hands it to the others. */
  int32_t use_rccl;
  uint8_t comm_id[128];
} mpsfm_ba_options;

#define MPSFM_MAX_TRACE 64

enum mpsfm_termination {
  MPSFM_TERM_FUNCTION_TOLERANCE = 0,
  MPSFM_TERM_GRADIENT_TOLERANCE = 1,
  MPSFM_TERM_PARAMETER_TOLERANCE = 2,
  MPSFM_TERM_MAX_ITERATIONS = 3,
  MPSFM_TERM_MIN_RADIUS = 4,
  MPSFM_TERM_INVALID_STEPS = 5, /* Ceres FAILURE: too many consecutive invalid steps */
  MPSFM_TERM_NO_VARIABLES = 6
};

typedef struct mpsfm_ba_summary {
  double initial_cost;       /* includes fixed_cost */
  double final_cost;         /* includes fixed_cost */
  double fixed_cost;         /* blocks whose camera and point are both constant */
  int32_t num_iterations;    /* LM iterations performed (iteration 0 not counted) */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t termination;       /* mpsfm_termination */
  int64_t num_residual_blocks;   /* reprojection + depth blocks in the problem   */
  int64_t num_residual_evals;    /* residual blocks x (cost or Jacobian) evaluations */
  int64_t num_jacobian_evals;    /* Jacobian sweeps */
  int32_t reduced_dim;       /* order of the reduced camera system */
  double final_radius;
  double time_total_s;       /* wall time of the solve, device-synchronised */
  double time_linearize_s;   /* device time: track sweep (residual/Jacobian/Schur reduce) */
  double time_dense_s;       /* device time: reduced camera system factor + solve          */
  double time_update_s;      /* device time: back-substitution + candidate cost sweep      */
  int32_t trace_len;
  double trace_cost[MPSFM_MAX_TRACE];     /* cost after each iteration (index 0 = initial)  */
  double trace_radius[MPSFM_MAX_TRACE];
  uint8_t trace_accepted[MPSFM_MAX_TRACE];
} mpsfm_ba_summary;

typedef struct mpsfm_ba_handle mpsfm_ba_handle;

/* -- library ------------------------------------------------------------------------- */
int mpsfm_abi_version(void);
const char* mpsfm_last_error(void);
/* number of visible gfx950 devices (0 on a CPU-only host; does not initialise a context) */
int mpsfm_device_count(void);
void mpsfm_ba_default_options(mpsfm_ba_options* opt);
/* ncclGetUniqueId of the RCCL library found in the process (dlopen): 128 bytes for mpsfm_ba_options.comm_id */
int mpsfm_comm_unique_id(uint8_t id[128]);

/* -- bundle adjustment: replaces pyceres.solve(options, bundler.problem, summary)
 *    (bundle_adjustment.py:184, 285-293) ---------------------------------------------- */

/* One-shot: upload, solve, write the refined poses/points back into `state`. */
int mpsfm_ba_solve(const mpsfm_ba_problem* problem, mpsfm_ba_state* state,
                   const mpsfm_ba_options* options, mpsfm_ba_summary* summary);

/* Resident form: the problem lives in HBM between calls (what bench.py times). */
int mpsfm_ba_create(const mpsfm_ba_problem* problem, const mpsfm_ba_state* initial,
                    const mpsfm_ba_options* options, mpsfm_ba_handle** out);
int mpsfm_ba_set_state(mpsfm_ba_handle* h, const mpsfm_ba_state* state); /* H2D            */
int mpsfm_ba_reset_state(mpsfm_ba_handle* h);  /* D2D: back to the state given at create   */
int mpsfm_ba_solve_resident(mpsfm_ba_handle* h, mpsfm_ba_summary* summary);
int mpsfm_ba_get_state(mpsfm_ba_handle* h, mpsfm_ba_state* state);       /* D2H            */
void mpsfm_ba_destroy(mpsfm_ba_handle* h);

/* Cost of the current resident state: 1/2 sum rho, split by block kind. */
int mpsfm_ba_eval_cost(mpsfm_ba_handle* h, double* cost_reproj, double* cost_depth);

/* One Jacobian/Schur track sweep at the current state with trust-region radius `radius`
 * (no parameter update): the launches bench.py prices against the HBM roofline.
 * `elapsed_ms` receives the HIP-event time of the whole sweep (dense chunks, reduction of their slabs,
 * general chunks and long tracks). */
int mpsfm_ba_sweep_once(mpsfm_ba_handle* h, double radius, float* elapsed_ms);
/* The parts of the last mpsfm_ba_sweep_once: ms = {k_track_sweep_dense, k_reduce_slabs, general + long-track kernels},
 * info = {dense chunks, general chunks, long tracks, destination parts of the reduction}.  Either pointer may be NULL. */
int mpsfm_ba_sweep_parts(mpsfm_ba_handle* h, float ms[3], int64_t info[4]);
/* Download the reduced camera system built by the last sweep: S (n x n, row-major, symmetric)
 * and rhs (n); n = summary.reduced_dim = 6 x variable cameras, rows in the CALLER's camera order
 * (the handle keeps its own slot order, see mpsfm_ba_dense_plan).  Test/diagnostic entry point. */
int mpsfm_ba_get_reduced_system(mpsfm_ba_handle* h, double* S, double* rhs, int32_t n);
int mpsfm_ba_reduced_dim(mpsfm_ba_handle* h);
/* Solution y (scaled coordinates, length n) of the last dense solve.  Test/diagnostic. */
int mpsfm_ba_get_dense_solution(mpsfm_ba_handle* h, double* y, int32_t n);
/* Factor + solve only, on the last assembled system (prices the MFMA dense solve). */
int mpsfm_ba_dense_solve_once(mpsfm_ba_handle* h, float* elapsed_ms);
/* How the handle factors the reduced camera system (what Ceres' fill-reducing ordering and sparse
 * Cholesky do behind bundle_adjustment.py:288).  info[0..9] = camera slots (= variable cameras),
 * 32-column tile columns of the system incl. the alignment padding, levels of the tile elimination tree (= factorisation launches),
 * nested-dissection depth (-1: caller's camera order), 1 if the back substitution uses the
 * inverse accumulators, work items, tile products, inverse roles, 6x6 blocks of S stored,
 * back-substitution launches. */
int mpsfm_ba_dense_plan(mpsfm_ba_handle* h, int64_t info[10]);

/* -- point covariances: replaces pycolmap.estimate_ba_covariance(POINTS)
 *    (bundle_adjustment.py:244-261).  cov[j] = (sum_i magnitude * Jp_i^T Jp_i)^-1 over the
 *    reprojection blocks of point j with every other variable held constant. ------------- */
int mpsfm_point_covs(const mpsfm_ba_problem* problem, const mpsfm_ba_state* state,
                     int32_t device, double* covs /* [n_pts][3][3] */);

/* -- per-track triangulation numerics: the arithmetic inside
 *    pycolmap.IncrementalTriangulator / ObservationManager used at
 *    mpsfm/sfm/mapper/triangulator.py:48,53-55,123-128 and mapper/base.py:686-797 -------- */
typedef struct mpsfm_tracks {
  int32_t n_cams, n_tracks, n_intr;
  const double* cam_quat_xyzw; /* [n_cams][4] */
  const double* cam_t;         /* [n_cams][3] */
  const double* cam_intr;      /* [n_intr][4] */
  const int32_t* cam_intr_idx; /* [n_cams]    */
  const int64_t* track_start;  /* [n_tracks+1] CSR offsets into the element arrays */
  const int32_t* el_cam;       /* [n_el] */
  const double* el_xy;         /* [n_el][2] */
} mpsfm_tracks;

/* Linear multi-view triangulation of every track (COLMAP TriangulateMultiViewPoint):
 * xyz[t] = smallest eigenvector of sum_i (P_i - x_i x_i^T P_i)^T (...), dehomogenised. */
int mpsfm_triangulate_tracks(const mpsfm_tracks* tracks, int32_t device,
                             double* xyz /* [n_tracks][3] out */);

/* Per-track quality numbers at given points: max pairwise triangulation angle (radians),
 * per-element squared reprojection error and cheirality (depth > 0) flags.
 * Any output pointer may be NULL. */
int mpsfm_filter_tracks(const mpsfm_tracks* tracks, const double* xyz /* [n_tracks][3] */,
                        int32_t device, double* max_tri_angle /* [n_tracks] */,
                        double* el_sq_err /* [n_el] */, uint8_t* el_front /* [n_el] */);

/* -- row f2: track-graph logic of pycolmap.IncrementalTriangulator as MpsfmTriangulator drives it
 *    (mpsfm/sfm/mapper/triangulator.py:32-48, 88-100, 123, 165-175): Find / Create / Continue, Complete, Merge,
 *    Retriangulate with the fork's ignore_image_ids.  COLMAP 3.11 semantics (the fork's source is not in the reference
 *    tree: parity unpinned).  The scene stays with the caller: it hands over the keypoints + correspondence graph once
 *    and the mutable state before a call; every call leaves an operation log the caller replays on its
 *    ObservationManager (add_point3D / add_observation / delete_point3D).  Keypoints are addressed by their global
 *    index kp_start[image] + point2D_idx; points by their index in the state's arrays (new points continue that range). */
typedef struct mpsfm_tri_options {       /* pycolmap.IncrementalTriangulatorOptions fields used by the calls below */
  int32_t max_transitivity;              /* 1   (only the direct correspondences are implemented)                 */
  double create_max_angle_error;         /* 2.0 degrees                                                            */
  double continue_max_angle_error;       /* 2.0                                                                    */
  double merge_max_reproj_error;         /* 4.0 px                                                                 */
  double complete_max_reproj_error;      /* 4.0 px                                                                 */
  int32_t complete_max_transitivity;     /* 5                                                                      */
  double re_max_angle_error;             /* 5.0                                                                    */
  double re_min_ratio;                   /* 0.2                                                                    */
  int32_t re_max_trials;                 /* 1                                                                      */
  double min_angle;                      /* 1.5 (the mapper overrides it to 0.001, mapper/base.py:35-40)          */
  int32_t ignore_two_view_tracks;        /* 1   (the mapper overrides it to 0)                                     */
} mpsfm_tri_options;
typedef struct mpsfm_tri_graph {
  int32_t n_images;
  const int64_t* kp_start;   /* [n_images+1] */
  const double* kp_xy;       /* [n_kp][2] Point2D.xy */
  const double* cam_intr;    /* [n_images][4] PINHOLE fx fy cx cy */
  const int64_t* corr_start; /* [n_kp+1] CSR of the correspondence graph */
  const int64_t* corr_kp;    /* [n_corr] matched keypoint (global index) */
} mpsfm_tri_graph;
typedef struct mpsfm_tri_state {
  const uint8_t* registered;      /* [n_images] image.has_pose */
  const double* cam_quat_xyzw;    /* [n_images][4] */
  const double* cam_t;            /* [n_images][3] */
  const int64_t* kp_point;        /* [n_kp] index of the keypoint's 3-D point in xyz, or -1 */
  int64_t n_points;
  const double* xyz;              /* [n_points][3] */
} mpsfm_tri_state;
typedef struct mpsfm_triangulator mpsfm_triangulator;
enum mpsfm_tri_op { MPSFM_TRI_ADD_POINT = 0 /* a = point, b = track length, xyz */, MPSFM_TRI_ADD_OBS = 1 /* a = point, b = keypoint */,
                    MPSFM_TRI_DELETE_POINT = 2 /* a = point */ };

void mpsfm_tri_default_options(mpsfm_tri_options* o);
int mpsfm_triangulator_create(const mpsfm_tri_graph* graph, int32_t device, mpsfm_triangulator** out);
void mpsfm_triangulator_destroy(mpsfm_triangulator* h);
int mpsfm_triangulator_set_state(mpsfm_triangulator* h, const mpsfm_tri_state* state);
int mpsfm_triangulator_triangulate_image(mpsfm_triangulator* h, const mpsfm_tri_options* o, int32_t image, int64_t* count);
int mpsfm_triangulator_complete_image(mpsfm_triangulator* h, const mpsfm_tri_options* o, int32_t image, int64_t* count);
/* n < 0: every track (complete_all_tracks / merge_all_tracks) */
int mpsfm_triangulator_complete_tracks(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int64_t* points, int64_t n, int64_t* count);
int mpsfm_triangulator_merge_tracks(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int64_t* points, int64_t n, int64_t* count);
int mpsfm_triangulator_retriangulate(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int32_t* ignore_images, int32_t n_ignore,
                                     int64_t* count);
/* operation log of the last call, in the order the operations happened */
int64_t mpsfm_triangulator_num_ops(mpsfm_triangulator* h);
int64_t mpsfm_triangulator_num_points(mpsfm_triangulator* h);
int mpsfm_triangulator_get_ops(mpsfm_triangulator* h, int32_t* type, int64_t* a, int64_t* b, double* xyz /* [n_ops][3] */);
/* track elements (global keypoint indices) of the log's ADD_POINT operations, concatenated in log order */
int64_t mpsfm_triangulator_num_op_elements(mpsfm_triangulator* h);
int mpsfm_triangulator_get_op_elements(mpsfm_triangulator* h, int64_t* els);
int mpsfm_triangulator_stats(mpsfm_triangulator* h, int64_t* batch, int64_t* batch_hits, int64_t* host_estimates);

/* COLMAP's EstimateTriangulation (LORANSAC<TriangulationEstimator, ..., InlierSupportMeasurer, CombinationSampler>, what
 * IncrementalTriangulator::Create / CompleteImage run per candidate track; reference call sites
 * mpsfm/sfm/mapper/triangulator.py:88-100, 123) for many independent candidate tracks in ONE launch: one thread per
 * candidate, at most 64 views each (MPSFM_EUNSUPPORTED beyond).  The engine above uses the same kernel for its batches. */
typedef struct mpsfm_tri_candidates {
  int64_t n_candidates;
  const int64_t* cand_start;          /* [n_candidates+1] CSR into the view arrays                                  */
  const double* view_cam_from_world;  /* [n_views][3][4] row-major                                                  */
  const double* view_intr;            /* [n_views][4] PINHOLE fx fy cx cy                                           */
  const double* view_xy;              /* [n_views][2] pixel measurement                                             */
  double min_tri_angle;               /* radians (options.min_angle)                                                */
  double max_error;                   /* radians when residual_type = 0, pixels when 1                              */
  int32_t residual_type;              /* 0 = ANGULAR_ERROR (Create), 1 = REPROJECTION_ERROR (CompleteImage)         */
  const int64_t* min_num_trials;      /* [n_candidates] or NULL = C(n,2) up to 15 views, else 0 (Create's rule)     */
} mpsfm_tri_candidates;
int mpsfm_tri_estimate_batch(const mpsfm_tri_candidates* c, int32_t device, double* xyz /* [n][3] */, uint8_t* ok /* [n] */,
                             uint8_t* inlier /* [n_views] */);

/* -- row f3: depth-block selection of Optimizer.__build_problem for a whole bundle in one launch
 *    (mpsfm/sfm/mapper/bundle_adjustment.py:124-161) and the whitened log-depth errors of
 *    update_truncation_multiplier (:295-333).  Per keypoint that has a 3-D point: bilinear samples of the validity
 *    mask and the depth map (PriorUtils._data_at_kps, image/mixins/priorutils.py:49-62: grid_sample, zero padding,
 *    align_corners=True, keypoints scaled by camera.sx / sy), the camera-frame depth of the point
 *    (points3D_utils.py:9-25), the masks and the loss weights.
 *    flags bit 0: valid_at_kps (sample == 1), bit 1: depth > 0, bit 2: 1/f < depth/depth3d < f,
 *          bit 3: |log d.clip - log z.clip| / sqrt(var) < 3 (gross_outliers test, :145-147)
 *    magnitude = d^2 / clip(var, 1e-6) (:161), param = multiplier * sqrt(var) / d (:160),
 *    whitened  = (log d - log z) / clip(sqrt(var) / d, 1e-6) (:323-329). ------------------------------------- */
typedef struct mpsfm_depth_gather {
  int32_t n_images;
  const int32_t* map_h; const int32_t* map_w;    /* [n_images] */
  const double* const* depth_map;                 /* [n_images] -> H*W, depth.data ("update") or depth.data_prior */
  const uint8_t* const* valid_map;                /* [n_images] -> H*W, depth.valid */
  const double* sx; const double* sy;             /* [n_images] camera.sx, camera.sy */
  const double* cam_quat_xyzw; const double* cam_t; /* [n_images][4], [n_images][3] */
  int64_t n_obs;                                  /* keypoints with a 3-D point, all images */
  const int32_t* obs_img;                         /* [n_obs] image index */
  const double* obs_xy;                           /* [n_obs][2] keypoint, original image scale */
  const double* obs_var;                          /* [n_obs] depth.uncertainty_update[point2D_idx] */
  const int32_t* obs_pt;                          /* [n_obs] index into pts */
  int32_t n_pts;
  const double* pts;                              /* [n_pts][3] */
  int32_t scale_filter; double scale_filter_factor; int32_t gross_outliers;
  double multiplier;                              /* param_multiplier * truncation_multiplier * rob_std (:109,159) */
} mpsfm_depth_gather;

int mpsfm_depth_blocks(const mpsfm_depth_gather* g, int32_t device, uint8_t* flags /* [n_obs] */, double* depth,
                       double* depth3d, double* magnitude, double* param, double* whitened /* each [n_obs] */);

/* -- depth-from-normals integration: replaces the per-image solve of Image.integrate()
 *    (mpsfm/sfm/scene/image/integration.py:133-137 -> _integrate :383-520): IRLS over a 5-point SPD
 *    system on the H*W log-depths with Jacobi-preconditioned CG (scipy/cupy `cg` semantics), bilateral
 *    weights sigmoid(k ((A2 z)^2 - (A1 z)^2)), depth-prior and sparse-depth terms.  Maps are row-major
 *    [H][W]; `normals` is [H][W][3] in the reference's channel order (nx = ch 1, ny = ch 0, nz = -ch 2,
 *    :273-275); `normals_var` holds the diagonal (00, 11, 22) of the per-pixel normal covariance. ---- */
#define MPSFM_INT_MAX_IRLS 16
typedef struct mpsfm_int_problem {
  int32_t H, W;
  const double* depth_prior;        /* depth.data_prior                                   */
  const double* depth_uncertainty;  /* depth.uncertainty (variance)                       */
  const uint8_t* valid;             /* depth.valid                                        */
  const double* normals;            /* normals.data                                       */
  const double* normals_var;        /* diag of normals.uncertainty                        */
  const double* depth_init;         /* depth.data: the map being refined (checkpoint)     */
  double K[4];                      /* (K[1,1] sy, K[0,0] sx, K[1,2] sy, K[0,2] sx), :118-124 */
  int32_t n_sparse;                 /* sparse 3-D points projected into the map (:99-116) */
  const int32_t* sparse_x; const int32_t* sparse_y;
  const double* sparse_depth3d; const double* sparse_zvar;
  /* Image.default_conf (mpsfm/sfm/scene/image/base.py:30-55) */
  double large_number, tol, step_size, cg_tol, lambda1, lambda2, k;
  double depth_magnitude_multiplier, normals_magnitude_multiplier, scale_filter_factor;
  int32_t max_iter, cg_max_iter, scale_filter;
  /* state the reference caches on the image between calls (IntVars, :18-29) */
  int32_t init;                     /* _integrate(init=...)                                */
  int32_t integrated;               /* in                                                  */
  double energy_old;                /* in                                                  */
  double* wu; double* wv;           /* [H*W] in (when init && integrated) / out, may be NULL */
} mpsfm_int_problem;

typedef struct mpsfm_int_summary {
  int32_t changed;                  /* 1: depth_out holds the new map; 0: frame skipped     */
  int32_t irls_iterations, cg_iterations_total, integrated_out;
  double energy_initial, energy_final, energy_old_out;
  int32_t cg_iters[MPSFM_INT_MAX_IRLS];
  double energies[MPSFM_INT_MAX_IRLS + 1];
  float ms;                         /* device time of the solve (HIP events)                */
} mpsfm_int_summary;

int mpsfm_integrate_depth(const mpsfm_int_problem* problem, int32_t device, double* depth_out /* [H*W] */,
                          mpsfm_int_summary* summary);

/* The same solve for a batch of images in ONE sequence of launches (what MpsfmMapper.integrate_bundle loops
 * over, reference mapper/base.py:619-631): kernels run on a (pixels, image) grid, every image keeps its own
 * IRLS / CG state and stops on its own tests, so the results equal those of n single calls (bit for bit while
 * the batch has fewer than 400 000 pixels in total; larger batches use two pixels per thread in the CG
 * kernels, which regroups the partial sums: last-bit differences) while the launch and synchronisation latency
 * is paid once.  All images must share H, W and the configuration
 * scalars (MPSFM_EINVAL otherwise); summaries[i].ms is the device time of the whole batch. */
int mpsfm_integrate_depth_batch(int32_t n_images, const mpsfm_int_problem* problems /* [n] */, int32_t device,
                                double* const* depth_out /* [n] pointers to H*W */, mpsfm_int_summary* summaries /* [n] */);

/* ---- row f4: uncertainty propagation through the integration (reference
 *    mpsfm/sfm/scene/image/integration.py:51-79 `IntegrationUncertainty`, :522-574 `calculate_hessian`,
 *    :576-616 `calculate_int_covs_at_points / _at_kps`).  The "Hessian" is the matrix of calc_Amat built at
 *    the depth checkpoint `depth_init` with weights recomputed from it (init=False) and, when use_sparse==0
 *    (conf.ignore_depths, the default), without the sparse-point term; no scale filter.  The reference solves
 *    H x = e_k per query pixel with cholespy (float32) and returns x.sum(0): the column sum of H^-1, which by
 *    symmetry is (H^-1 1)[k] — computed here by ONE preconditioned-CG solve in float64 to `rtol`.
 *    var_out[i] = value at pixel (qx[i], qy[i]); field_out (may be NULL) receives the whole H*W field.
 *    summary: cg_iters[0] iterations, changed = 1 when the tolerance was met, ms = device time. ---- */
int mpsfm_integration_variances(const mpsfm_int_problem* problem, int32_t device, int32_t use_sparse, int32_t n_query,
                                const int32_t* qx, const int32_t* qy, double rtol, int32_t max_iter,
                                double* var_out /* [n_query] */, double* field_out /* [H*W] or NULL */,
                                mpsfm_int_summary* summary);

#ifdef __cplusplus
}
#endif
#endif /* MPSFM_HIP_H */
