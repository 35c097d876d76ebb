/*
 * include/nbk.h -- C-ABI of libnbk (numbotics_amd/csrc), the MI355X (gfx950) batched kinematics +
 * collision-validity engine.
 *
 * The reference (landonclark97/numbotics) has no FFI layer: its de-facto kernel ABI is the numba
 * signatures of the two batched kernels plus the Python methods that wrap PyBullet (SURVEY.md
 * section 8b).  Each entry point below names the reference interface it replaces; paths are relative
 * to the reference repository root.
 *
 * Conventions
 *   - plain pointers and sizes, no torch types; every array pointer that is not marked "host" is a
 *     DEVICE pointer (hipMalloc / torch.cuda tensor.data_ptr()) on the current device;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *     with respect to the host and capturable into a hipGraph (no allocation, no synchronisation).
 *     Two qualifications, both about the library's own scratch memory: nbk_validity_batch and
 *     nbk_edge_validity_batch keep one scratch set per (descriptor, stream), allocated on that
 *     stream's FIRST call (and regrown when a later batch needs more): such a call allocates and may
 *     wait for that stream's earlier work.  Captured before the scratch exists they return
 *     NBK_ERR_UNSUPPORTED (run the call once outside the capture, or use nbk_validity_batch_ws, whose
 *     scratch is the caller's); captured after, they are self-contained graph nodes: each replay
 *     prepares its own tables and clears its own counters in the stream's scratch, and direct calls
 *     made on that stream afterwards (in any order with the replays) do the same -- they no longer
 *     reuse tables between calls.  A graph holds the scratch's address: do not replay it after a
 *     direct call on the same stream with a LARGER batch has regrown the scratch (re-capture).  The *_host
 *     conveniences synchronise by definition.  Batches of 2^21 configurations or more (and edge batches of that many samples)
 *     run every other 2^20-configuration tile on a second, library-owned stream forked from and joined to `stream` with
 *     events -- the call still begins after, and completes before, its neighbours in `stream`'s order;
 *   - every compute call must be made with the descriptor's device current (hipSetDevice):
 *     NBK_ERR_INVALID otherwise;
 *   - float64 everywhere (the reference computes in float64); q is row-major (B, n_q);
 *   - return value: NBK_OK or a negative status; no exceptions cross the boundary;
 *   - a descriptor is immutable after creation and may be shared by streams and threads.
 */
#ifndef NBK_H
#define NBK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBK_ABI_VERSION 2

enum {
    NBK_OK = 0,
    NBK_ERR_INVALID = -1,      /* bad argument (null pointer, negative size, bad index) */
    NBK_ERR_NO_DEVICE = -2,    /* no HIP device / not a gfx950 code object for this device */
    NBK_ERR_HIP = -3,          /* a HIP runtime call failed; see nbk_last_error() */
    NBK_ERR_UNSUPPORTED = -4,  /* descriptor exceeds a compiled-in limit (32 joints / DoF, 2^20 pairs, LDS capacity), or this
                                  entry point cannot serve this descriptor (per-pair distances of robots with 30+ primitives) */
    NBK_ERR_ALLOC = -5
};

/* shape / joint codes used inside descriptors */
enum { NBK_SPHERE = 0, NBK_CAPSULE = 1, NBK_BOX = 2, NBK_CYLINDER = 3, NBK_PLANE = 4, NBK_HULL = 5 };
enum { NBK_REVOLUTE = 0, NBK_PRISMATIC = 1 };
enum { NBK_CONNECT = 0, NBK_STEER = 1 };

#define NBK_MAX_JOINTS 32
#define NBK_MAX_DOF 32

/*
 * Flat robot + scene description ("compile the robot": replaces the per-frame flattening of
 * numbotics/robots/arm.py:17-71 and the per-query shape/pair bookkeeping of arm.py:190-250,555-580).
 * All pointers are HOST pointers; nbk_model_create copies everything to the device.
 */
typedef struct {
    int32_t n_q;                  /* degrees of freedom (columns of q) */
    int32_t n_joints;             /* movable joints = moving frames, parents first */
    const int32_t *joint_parent;  /* [J] parent moving frame, -1 = base */
    const int32_t *joint_type;    /* [J] NBK_REVOLUTE / NBK_PRISMATIC */
    const int32_t *joint_qidx;    /* [J] column of q */
    const double *joint_rot;      /* [J][27] M0 = R K, M1 = R (K - I), M2 = R [a]x (3x3 row-major each);
                                     local rotation = M0 - cos(q) M1 + sin(q) M2 (robots/helpers.py:43-55) */
    const double *joint_trans;    /* [J][3] translation of the (fixed-merged) joint offset */
    const double *joint_slide;    /* [J][3] R a for prismatic joints, 0 otherwise */
    const double *joint_axis;     /* [J][3] joint axis in the joint frame */
    const double *base_pose;      /* [12] 3x4 row-major */
    int32_t n_rshapes;            /* robot collision primitives */
    const int32_t *rshape_frame;  /* [S] moving frame carrying the shape, -1 = base */
    const int32_t *rshape_type;   /* [S] NBK_SPHERE / CAPSULE / BOX / CYLINDER / HULL */
    const double *rshape_local;   /* [S][12] pose of the primitive in its moving frame */
    const double *rshape_param;   /* [S][4] sphere r | capsule r,hl | cylinder r,hl | box hx,hy,hz | hull index ; [3] = margin */
    int32_t n_wshapes;            /* static world primitives (obstacles) */
    const int32_t *wshape_type;   /* [W] the above or NBK_PLANE (param[0..2] = unit normal) */
    const double *wshape_pose;    /* [W][12] world pose */
    const double *wshape_param;   /* [W][4] */
    int32_t n_pairs;              /* allowed (shape, shape) pairs, sorted by pair_a */
    const int32_t *pair_a;        /* [P] robot shape */
    const int32_t *pair_b;        /* [P] robot shape (< S) or S + world shape */
    /* Convex hulls of MESH collision shapes (numbotics/utils/shape.py:81-94 hands the file of numbotics/utils/mesh.py:18-37
     * to pybullet GEOM_MESH, i.e. one convex hull per mesh object; numbotics/physics/helpers.py:252-255 for URDF <mesh>).
     * A shape of type NBK_HULL names its hull in param[0]; vertices and face planes are in the primitive's local frame,
     * whose origin must be a point of the hull (the host uses the mean of the hull's vertices); margin inflates the hull. */
    int32_t n_hulls;
    const int32_t *hull_vert_begin;  /* [H+1] */
    const double *hull_verts;        /* [NV][3] */
    const int32_t *hull_face_begin;  /* [H+1]; a hull may have no planes (flat point set): distances stay exact; the
                                        penetration depth is EPA's either way, without planes its fallback (when EPA gives no
                                        answer) has only the other shape's axes and the centre line */
    const double *hull_planes;       /* [NF][4] unit outward normal n and offset d: inside n.x <= d.  nbk_model_create checks
                                        ||n|^2 - 1| <= 1e-9 and n.v <= d (+1e-9 relative) for every vertex of the hull and
                                        returns NBK_ERR_INVALID otherwise: the broadphase certifies collisions from the ball
                                        these planes inscribe */
} nbk_model_desc;

typedef struct nbk_model nbk_model;

int32_t nbk_abi_version(void);
const char *nbk_status_string(int32_t status);
const char *nbk_last_error(void);          /* text of the last HIP failure on this thread */
int32_t nbk_device_count(void);            /* 0 when no GPU is visible; never fails */

int32_t nbk_model_create(const nbk_model_desc *desc, nbk_model **out);
void nbk_model_destroy(nbk_model *m);
int32_t nbk_model_num_pairs(const nbk_model *m);

/*
 * Batched forward kinematics of one frame.
 * Replaces nb_compute_transformation + nb_joint_transform (numbotics/robots/helpers.py:33-113) as
 * called by Arm.forward_kinematics (numbotics/robots/arm.py:369-410).
 *   path  (host) [path_len] joint indices root -> frame;  local (host) [12] constant pose after the last
 *   joint (trailing fixed joints, COM offset, a single local_pose);  local_pose (device, optional)
 *   [B][16] per-configuration right factor;  T_out (device) [B][16] row-major 4x4.
 */
int32_t nbk_fk_batch(const nbk_model *m, const double *q, int64_t B, const int32_t *path, int32_t path_len,
                     const double *local, const double *local_pose, double *T_out, void *stream);

/*
 * FK of many frames of every configuration in one sweep (additive: the reference computes one frame per call,
 * numbotics/robots/arm.py:369-410; callers that need every link pose -- visualisation, proximity bookkeeping -- loop over
 * the link names).  A frame set names each frame by the moving frame (joint index, -1 = base) it hangs off and its constant
 * 3x4 local pose (trailing fixed joints, COM offset); host arrays, copied to the device once.
 *   T_out (device) [B][n_frames][16], frame order as given.  Poses are bit-identical to nbk_fk_batch on the same frame.
 */
typedef struct nbk_frameset nbk_frameset;
int32_t nbk_frameset_create(const nbk_model *m, int32_t n_frames, const int32_t *frame_joint, const double *frame_local,
                            nbk_frameset **out);
void nbk_frameset_destroy(nbk_frameset *fs);
int32_t nbk_fk_frames_batch(const nbk_model *m, const nbk_frameset *fs, const double *q, int64_t B, double *T_out,
                            void *stream);

/*
 * Batched geometric Jacobian [v; w] of one frame.
 * Replaces nb_compute_jacobian (numbotics/robots/helpers.py:117-187) as called by Arm.jacobian
 * (numbotics/robots/arm.py:413-461).  mode 0: end pose = T*local; 1: T*local*pose[b] (local_pose);
 * 2: end position = translation of pose[b] (global_pose).  J_out (device) [B][6][n_q].
 */
int32_t nbk_jacobian_batch(const nbk_model *m, const double *q, int64_t B, const int32_t *path,
                           int32_t path_len, const double *local, int32_t mode, const double *pose,
                           double *J_out, void *stream);

/*
 * Batched Arm.inverse_kinematics (numbotics/robots/arm.py:464-552): damped least squares
 *   q <- q + J^T (J J^T + lambda I)^-1 diff,  diff = [p* - p ; vee(0.5 (R - R^T))], R = R* R_ee^T
 * (numbotics/math/spatial.py:207-212), lambda_0 = 0.1, x1.2 / failures+1 when |diff| grew, x0.5 / failures = 0 otherwise;
 * an element iterates while |diff| > tol and failures < max_failures, at most max_iter times.
 *   pose [B][16] target poses, q0 [B][n_q] starts; path / local as for nbk_fk_batch (host);
 *   limits (host, optional) [n_q][2]: clip q to [lower, upper] after every step (use_limits=True);
 *   q_out [B][n_q]; success [B] uint8 = |diff| < tol; diff_norm (optional) [B]; iters (optional) [B] int32 steps taken.
 * The 6x6 damped system is solved by an unpivoted Cholesky (the reference calls LAPACK LU): equal to rounding.
 */
int32_t nbk_ik_batch(const nbk_model *m, const double *pose, const double *q0, int64_t B, const int32_t *path,
                     int32_t path_len, const double *local, const double *limits, double tol, int32_t max_iter,
                     int32_t max_failures, double *q_out, uint8_t *success, double *diff_norm, int32_t *iters,
                     void *stream);

/*
 * Batched Arm.in_collision (numbotics/robots/arm.py:603-604): bit b of mask_bits / mask_bytes[b] is 1
 * iff min over the allowed pairs of the signed distance is < threshold (strict).  Replaces the
 * per-configuration PyBullet round trip Arm.collisions -> Chain.distance_to -> getClosestPoints
 * (arm.py:555-580, numbotics/physics/chain.py:944-969).
 * A configuration with a NaN or infinite joint value is reported as colliding.
 *   mask_bits  (device, optional) [ceil(B/64)] uint64, bit (b % 64) of word (b / 64);
 *   mask_bytes (device, optional) [B] uint8.  At least one must be given.
 */
int32_t nbk_validity_batch(const nbk_model *m, const double *q, int64_t B, double threshold,
                           uint64_t *mask_bits, uint8_t *mask_bytes, void *stream);
/*
 * Same, with caller-owned scratch.  Every batch runs as a broadphase kernel that appends the surviving
 * (configuration, pair) items to a queue in `workspace`, followed by a dense narrowphase kernel.
 * nbk_validity_workspace_bytes(m, B) gives the size needed (0 only for B = 0 or a descriptor without pairs; a call that passes
 * no workspace runs the slower fused single-kernel path).  The queue is sized
 * for the worst case (every pair of every configuration of a tile survives): up to 1 GiB, up to 8 GiB for descriptors with
 * more than 512 pairs; larger batches are processed in tiles of that size.
 * nbk_validity_batch itself keeps one internal workspace per (descriptor, stream), grown with hipMalloc on demand: calls on
 * different streams share nothing and overlap; use this variant when the memory must be the caller's (allocator pools,
 * graphs that outlive a regrowth of the internal scratch).  `workspace` must be 64-byte aligned (hipMalloc / torch allocations
 * are): NBK_ERR_INVALID otherwise.
 */
int64_t nbk_validity_workspace_bytes(const nbk_model *m, int64_t B);
int32_t nbk_validity_batch_ws(const nbk_model *m, const double *q, int64_t B, double threshold,
                              uint64_t *mask_bits, uint8_t *mask_bytes, void *workspace,
                              int64_t workspace_bytes, void *stream);

/*
 * Batched Arm.closest_to (arm.py:599-600): min signed distance and the index of the pair attaining it
 * (first minimum in pair order; -1 / +inf when there are no pairs).
 */
int32_t nbk_closest_batch(const nbk_model *m, const double *q, int64_t B, double *min_dist,
                          int32_t *argmin, void *stream);

/*
 * Batched Arm.collisions (arm.py:555-580): signed distance of every allowed pair, dist [B][P], and
 * optionally the Proximity fields (numbotics/physics/collision.py:25-32) witness [B][P][9] =
 * position on subject, position on target, unit normal from target to subject.
 */
int32_t nbk_pair_distances_batch(const nbk_model *m, const double *q, int64_t B, double *dist,
                                 double *witness, void *stream);

/*
 * Batched Arm.jacobian_proximity (numbotics/robots/arm.py:620-632) over every allowed pair: besides dist [B][P] and
 * witness [B][P][9] (as above), jrows [B][P][n_q] with
 *   jrows[b][p] = n . Jv_subject(q_b; position_on_subject) - n . Jv_target(q_b; position_on_target),
 * n = normal_target_to_subject, Jv = the linear rows of Arm.jacobian(..., global_pose=trans_mat(pos=point))
 * (numbotics/robots/helpers.py:117-187); the target term is dropped for world targets (arm.py:628).
 * These are the gradient rows of the pair distances that IrisSolver's counter-example search consumes
 * (numbotics/planning/safe_sets.py:86-121).
 */
int32_t nbk_proximity_jacobian_batch(const nbk_model *m, const double *q, int64_t B, double *dist,
                                     double *witness, double *jrows, void *stream);

/*
 * Batched DiscreteConnector.connect / steer (numbotics/planning/sampling_based/connectors.py:57-100)
 * with the default linear trajectory (numbotics/planning/trajectories.py:6-22) and
 * validity_checker = not in_collision(q, threshold).
 *   starts, goals [E][n_q]; dist (optional) [E] = distance_func(start, goal), NULL = Euclidean norm;
 *   valid [E] uint8: 1 iff every sample T = arange(0, T_f, resolution/d) U {T_f} is collision free,
 *   0 also for d <= float32 eps (the reference returns None);
 *   end (optional) [E][n_q]: goal (connect) or traj(T_f) (steer), NaN for the degenerate edge;
 *   n_samples (optional) [E] int32 = len(T).
 * Asynchronous: the sample count is only known on the device, so the launches cover the CAPACITY of the stream's edge scratch
 * (at least E * (ceil(max_distance / resolution) + 2) samples, and 1.25 x what the previous call on the stream needed, which
 * the device reports through pinned memory) and blocks beyond the true count exit at once; edges that do not fit are walked
 * by one wave each -- same results.  (Robots whose primitives exceed the LDS-parked layout -- some 30+ -- size the scratch
 * with one read-back instead and cannot be captured.)
 */
int32_t nbk_edge_validity_batch(const nbk_model *m, const double *starts, const double *goals,
                                const double *dist, int64_t E, double resolution, double max_distance,
                                int32_t mode, double threshold, uint8_t *valid, double *end,
                                int32_t *n_samples, void *stream);

/*
 * Exact k nearest neighbours of every point among the points inserted before it (itself included): the neighbour lists
 * an insert-then-query loop over the reference's flat L2 index yields (numbotics/math/geometry/nearest_neighbors.py:6-85,
 * numbotics/planning/sampling_based/graph.py:165-178; faiss.IndexFlatL2 is a third-party dependency: tie-breaking and
 * its float32 summation order are unpinned).  points (device) [N][dim] float32, k <= 64;
 * out_idx (device) [N][k] int32, ascending by (distance, index), -1 padded for the first rows.
 * distance = sum_c (x_c - y_c)^2 accumulated in dimension order with separate float32 roundings.
 */
int32_t nbk_knn_prefix(const float *points, int32_t n_points, int32_t dim, int32_t k, int32_t *out_idx, void *stream);

/* Arithmetic-contract self test: elementwise sincos(a), sqrt(a), a/b computed by the device routines
 * the kernels use (all arrays device, length n). */
int32_t nbk_selftest_math(const double *a, const double *b, int64_t n, double *sin_out, double *cos_out,
                          double *sqrt_out, double *div_out, void *stream);

/*
 * The reference's scalar contracts from HOST memory, latency-optimised: Arm.in_collision(q) on one configuration
 * (numbotics/robots/arm.py:603-604) and DiscreteConnector.connect / steer on one edge (connectors.py:57-100), as the
 * sequential planners call them (numbotics/planning/sampling_based/planners/prm.py:40, rrt.py:36).  Inputs and results go through pinned,
 * device-mapped memory owned by the descriptor (no staging copies), on a private stream; the call returns when the result
 * is there.  q / start / goal / end: host [n_q].  dist < 0 = Euclidean norm.  Calls on one descriptor serialise.
 */
int32_t nbk_validity_scalar_host(const nbk_model *m, const double *q, double threshold, int32_t *in_collision);
int32_t nbk_edge_validity_scalar_host(const nbk_model *m, const double *start, const double *goal, double dist,
                                      double resolution, double max_distance, int32_t mode, double threshold,
                                      int32_t *valid, double *end, int32_t *n_samples);

/* Host-buffer conveniences for callers without a device allocator (PCIe inclusive; they allocate,
 * copy, run, copy back and synchronise). */
int32_t nbk_fk_batch_host(const nbk_model *m, const double *q, int64_t B, const int32_t *path,
                          int32_t path_len, const double *local, double *T_out);
int32_t nbk_validity_batch_host(const nbk_model *m, const double *q, int64_t B, double threshold,
                                uint8_t *mask_bytes);

#ifdef __cplusplus
}
#endif
#endif
