/* multiply_hip.h — C ABI of libmultiply_hip.so (gfx950 / MI355X).
 *
 * The reference (eth-ait/MultiPly) has no FFI: its hot path is PyTorch code under
 * code/lib/model/ that is entered through Multiply.forward (code/lib/model/multiply.py:174).
 * This library is the boundary introduced BELOW that Python signature: every entry point
 * takes raw device pointers, sizes and a hipStream_t (as void*), allocates nothing, never
 * throws, and returns 0 on success or a hipError_t / negative argument-error code.
 * multiply_amd/hip.py is the ctypes binding a maintainer would add to the reference
 * (see INTEGRATION.md); each entry point names the reference code it replaces.
 *
 * All float tensors are fp32, row-major, contiguous.  "count pointers" are device ints so that
 * data-dependent sizes never need a host round trip: kernels are launched for the upper bound
 * and read the real count on the device.
 */
#ifndef MULTIPLY_HIP_H
#define MULTIPLY_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MP_MAX_LAYERS 10
#define MP_MAX_CHUNKS 9
#define MP_BIAS_STRIDE 288 /* floats per layer in a packed bias table */
#define MP_SMPL_V 6890
#define MP_SMPL_J 24
#ifndef MP_KNN_CLUSTER      /* (overridable for experiments: -DMP_KNN_CLUSTER=32 -DMP_KNN_NC=216 + the same two environment variables for hip.py) */
#define MP_KNN_CLUSTER 32   /* vertices per nearest-neighbour cluster (a power of two <= 64).  Round 6: 32 x 216 instead of 64 x 108 -- the */
#define MP_KNN_NC 216       /* clusters (216*32 = 6912 >= 6890, padded); <= 511.  cluster scan is wave-uniform, so finer clusters prune   */
                            /* better: sampler warp -19 %, Jacobian -18 %, frame -1.4 ms; 16 x 431: another -0.4 ms, training +0.8 ms    */
#endif

enum { MP_ACT_NONE = 0, MP_ACT_SOFTPLUS = 1, MP_ACT_RELU = 2, MP_ACT_SIGMUL = 3 };

typedef struct {
    int n_chunk;   /* chunks of 32 output rows */
    int use_reg;   /* layer consumes the 256 register-fed K slots (previous layer output) */
    int use_in;    /* layer consumes the 64/96 input-fed K slots (encoded network input) */
    int act;       /* MP_ACT_* */
    int out_chunk; /* chunk returned in fp32 instead of feeding the next layer, -1 = none */
    int aux;       /* reverse sweep only (MP_ACT_SIGMUL): bits 0..7 = 1 + stored-sigmoid layer, bits 8..15 = capture id */
} MpLayer;

typedef struct {
    int n_layers;
    int total_chunks;
    MpLayer layer[MP_MAX_LAYERS];
} MpNet;

/* ---- weights -------------------------------------------------------------------------------
 * Packs one linear layer (optionally weight-normalised: w = g * v / ||v||_row,
 * networks.py:82-83) into the half (f16) MFMA-fragment layout of csrc/mlp_core.hpp and writes its
 * fp32 bias row, with an optional hoisted contribution  bias[r] += sum_c W[r][hoist_col0+c] *
 * hoist_vec[c]  (the pose / frame conditioning that the reference concatenates to every point,
 * networks.py:164-165, 279-281, 275-276, is the same for all points of a call).
 *   v [out_dim][in_dim], g [out_dim] or NULL, b [out_dim]
 *   rowmap [n_rows] : source row of packed row r, -1 = zero row   (n_rows multiple of 32)
 *   colmap [(8+ks_in)*32], colscale [...] : source column of K slot s (-1 = zero) and a factor
 *   wpack_layer : destination (n_rows/32 chunks of (8+ks_in)*2 KiB) or NULL to only write the bias
 *   bias_layer  : destination, MP_BIAS_STRIDE floats (rows >= n_rows are zeroed); bias_scale multiplies (b + hoist).
 * Softplus networks run in scaled units (csrc/mlp_core.hpp): the host passes colscale = K for input-fed slots and
 * bias_scale = K = 100 log2(e) on hidden layers, and colscale = 1/K on the last (linear) layer. */
int mp_pack_layer(const float* v, const float* g, const float* b, int out_dim, int in_dim, const int* rowmap,
                  int n_rows, const int* colmap, const float* colscale, int ks_in, int hoist_col0, int hoist_n,
                  const float* hoist_vec, float bias_scale, void* wpack_layer, float* bias_layer, void* stream);
/* mp_pack_layer for ALL layers of one network in ONE launch (a training forward repacks the sampler's half-precision SDF net of
 * every person: 9 layers x 2 launches each before).  table: n_layers (<= MP_MAX_LAYERS) DEVICE-resident records, one per layer,
 * fields as mp_pack_layer's arguments (device pointers; hoist_vec NULL or hoist_n 0 = no hoisting; wpack_layer NULL = bias
 * only); every layer's bias row is written for all MP_BIAS_STRIDE entries (zeros past n_rows). */
typedef struct {
    const void *v, *g, *b, *rowmap, *colmap, *colscale, *hoist_vec;
    void *wpack_layer, *bias_layer;
    int out_dim, in_dim, n_rows, hoist_col0, hoist_n;
    float bias_scale;
} MpPackLayer;
int mp_pack_layers(const MpPackLayer* table, int n_layers, int ks_in, void* stream);

/* ---- fused MLP evaluation -------------------------------------------------------------------
 * mp_mlp_sdf: ImplicitNet.forward restricted to the sdf column (networks.py:126-181; caller:
 * multiply.py:140-141 inside the sampler, ray_sampler.py:85-88).
 *   xc [*][3] canonical points, worklist[i] = point id (NULL: id = i), *count work items
 *   sdf_out[point id] <- sdf.  multires = 6 Fourier encoding of 3-D input. */
int mp_mlp_sdf(const MpNet* net, const void* wpack, const float* bias, const float* xc, const int* worklist,
               const int* count, int max_count, float* sdf_out, void* stream);

/* mp_mlp_sdf_x2: the same queries with SPLIT ACTIVATIONS (round 6; the sampler's default, `Multiply.sampler_sdf_mode = 'f16x2'`):
 * same arguments, same packed half-precision weights and bias table, but every activation travels as two halves x = hi + lo
 * (22 mantissa bits) and a product is two MFMAs, W_h x_h + W_h x_l, with fp32 accumulation and an fp32 softplus.  The error
 * against the fp32 reference network drops from ~1e-3 to ~1e-4 (what is left is the rounding of the weights), the depths the
 * reference's sampler derives from these values (ray_sampler.py:85-94, multiply.py:137-151) agree accordingly. */
int mp_mlp_sdf_x2(const MpNet* net, const void* wpack, const float* bias, const float* xc, const int* worklist,
                  const int* count, int max_count, float* sdf_out, void* stream);

/* mp_mlp_full: ImplicitNet.forward, all 1+256 outputs, for callers outside the fused renderer
 * (multiply_model.py:941-945 query_oc).  d_in = 3 (multires 6) or 4 (multires 10). out [n][257]. */
int mp_mlp_full(const MpNet* net, const void* wpack, const float* bias, const float* x, int d_in, int n,
                float* out, void* stream);

/* mp_mlp_shade: value + forward-mode tangents of the foreground ImplicitNet, i.e. sdf, d sdf/d x_c and the
 * 256 features in one pass (replaces the second forward + autograd.grad of multiply.py:643-659), then
 * normal = normalize(normalize(grad . Jinv), eps=1e-6) (multiply.py:661, :606).
 *   worklist[i] = point id; xc [*][3]; jinv [*][9] (row-major inverse of d x_d / d x_c)
 *   sdf_out[id], normal_out[id][3]; feat_frag: f16 B-fragments, 512 B per WORK INDEX (consumed by mp_mlp_color) */
int mp_mlp_shade(const MpNet* net, const void* wpack, const float* bias, const float* xc, const float* jinv,
                 const int* worklist, const int* count, int max_count, float* sdf_out, float* normal_out,
                 void* feat_frag, void* stream);

/* mp_mlp_shade_rev: same outputs as mp_mlp_shade, computed in reverse mode, segment by segment of the worklist:
 * a plain forward sweep that writes sigmoid(100 z) of every hidden unit as UNORM8 (mp_sig_bytes_per_point() = 2048 B per work
 * index) into `sig` (seg_points * mp_sig_bytes_per_point() B, seg_points a multiple of 256, ZERO-INITIALISED once by the
 * caller), then a reverse sweep through
 * the TRANSPOSED layers (gnet / gpack: the 'grad' plan of multiply_amd/hip.py; w8_slots: the sdf row of the last layer,
 * 256 halves in K-slot order).  Two network columns per point instead of the four of the forward-mode kernel. */
int mp_sig_bytes_per_point(void);
int mp_mlp_shade_rev(const MpNet* net, const void* wpack, const float* bias, const MpNet* gnet, const void* gpack,
                     const void* w8_slots, const float* xc, const float* jinv, const int* worklist, const int* count,
                     int max_count, float* sdf_out, float* normal_out, void* feat_frag, void* sig, int seg_points,
                     void* stream);

/* mp_mlp_color: RenderingNet.forward mode 'pose_no_view' (networks.py:277-281, 305-311):
 * sigmoid(MLP([x_c, n, lin_pose(pose) (hoisted), feat])) -> rgb_out[id][3]. */
int mp_mlp_color(const MpNet* net, const void* wpack, const float* bias, const float* xc, const float* normal,
                 const void* feat_frag, const int* worklist, const int* count, int max_count, float* rgb_out,
                 void* stream);

/* mp_background: NeRF++ inverted-sphere background (multiply.py:514-539, 682-726): for every ray, n_bg depths
 * -> depth2pts_outside -> bg ImplicitNet (frame code hoisted) -> bg RenderingNet ('nerf_frame_encoding') ->
 * bg_volume_rendering with AbsDensity -> bg_rgb[R][3].  z_bg [n_bg] are the (shared) inverse depths in
 * DESCENDING order as after the flip of multiply.py:516, or per ray [R][n_bg] if z_per_ray. */
int mp_background(const MpNet* net_imp, const void* wpack_imp, const float* bias_imp, const MpNet* net_ren,
                  const void* wpack_ren, const float* bias_ren, const float* dirs, const float* cam, const float* z_bg,
                  int z_per_ray, int n_rays, float radius, float* bg_rgb, void* stream);

/* ---- SMPL ------------------------------------------------------------------------------------
 * SMPLServer.forward (lib/model/smpl.py:50-94) -> lbs (lib/smpl/lbs.py:136-229): posed vertices, bone
 * transforms relative to the canonical pose (tfs_c_inv, may be NULL for absolute), posed joints.
 *   params [86] = scale, transl3, thetas72, betas10 (device).  work: >= 3*V + 24*16 + 207 + 64 floats. */
int mp_smpl_pose(const float* v_template, const float* shapedirs, const float* posedirs, const float* j_regressor,
                 const float* lbs_weights, const int* parents, const float* params, const float* tfs_c_inv,
                 float* verts, float* tfs, float* joints, float* work, void* stream);

/* Nearest-neighbour acceleration structure over one vertex set (exact K=1 search, replaces
 * pytorch3d.ops.knn_points, deformer.py:39): vertices gathered in cluster order + bounding spheres.
 *   perm [NC*CL] vertex ids in cluster order (-1 = padding); vsorted [NC*CL][4] = x,y,z,id-as-int-bits;
 *   cbound [NC + NC / 2][4] = centre, radius of every cluster, then of every PAIR of clusters (2c, 2c + 1): the coarse granularity the
 *   training-mode searches of mp_warp_inverse / mp_warp_inverse_shade run on; the other consumers read the first NC rows. */
int mp_knn_build(const float* verts, const int* perm, float* vsorted, float* cbound, void* stream);

/* Oriented box of the posed vertices inflated by `inflate` (multiply.py:208-214; PCA axes instead of trimesh's
 * minimum-volume box).  obb [15] = centre3, axes (3 rows), half extents3. */
int mp_obb(const float* verts, float inflate, float* obb, void* stream);

/* The MINIMUM-VOLUME oriented box the reference asks trimesh for (multiply.py:208-214: smpl_mesh.bounding_box_oriented,
 * primitives.Box(extents * 1.2, transform)) from the convex hull of the posed vertices.  The hull is computed by the caller
 * (multiply_amd/obb.py hull_search_inputs: Qhull); the search over (hull-facet normal, silhouette edge) candidates and the box
 * itself are computed here in fp64.  All inputs fp64, device resident:
 *   hull_verts [H][3]; normals [N][3] the distinct facet normals in the caller's order of preference; per hull edge
 *   edge_vec [E][3] and the normals of its two facets edge_na / edge_nb [E][3]; work [2 N] scratch.
 *   obb [16] as mp_obb: centre3, axes (3 rows: facet normal, rectangle sides), half extents3 x inflate, pad. */
int mp_obb_hull(const double* hull_verts, int n_hull_verts, const double* normals, int n_normals, const double* edge_vec,
                const double* edge_na, const double* edge_nb, int n_edges, float inflate, double* work, float* obb, void* stream);

/* The same box with the convex hull ALSO built on the device (gift wrapping: 8 workgroups on one XCD share the pivots of a
 * level-synchronous front, one of them keeps the edge table; fp64 predicates; ~0.35 ms for a 6 890-vertex body): no device -> host
 * copy of the posed vertices, no host hull.  A BATCH of n_bodies (<= 64) bodies in one call, their wraps side by side:
 * verts [n_bodies][n_verts][3] fp32 (n_verts < 6912); work [n_bodies][mp_obb_hull_device_work_bytes()] bytes, 8-byte aligned;
 * obb [n_bodies][16]; status [n_bodies][8] (device ints): {hull vertices, facets, edges, failed, rounds, pivot / insert / total
 * clocks x 16}: failed != 0 (a non-manifold patch from exactly coplanar vertices, or more than 2048 facets) leaves obb all-zero -- the caller
 * reads it with its other device counts and falls back to mp_obb_hull with a host-side hull. */
int mp_obb_hull_device_work_bytes(void);
int mp_obb_hull_device(const float* verts, int n_verts, int n_bodies, float inflate, void* work, float* obb, int* status, void* stream);
/* Test hook: on != 0 makes every following mp_obb_hull_device call take its give-up path (the wrap's workgroups spin on each other
 * and abandon the wrap after ~30 ms when they are not all resident: status[..][3] = 1, the caller falls back to the host hull).
 * Returns the previous setting. */
int mp_debug_hull_abandon(int on);

/* ---- training objective ---------------------------------------------------------------------
 * The per-ray / per-point terms of Loss.forward (reference code/lib/model/loss.py:6-177) and the gradient of their weighted sum in
 * ONE launch (csrc/loss.hip).  All pointers device fp32 unless noted:
 *   rgb [R][3] (NaN rows are left out), rgb_gt [R][3], acc [R], accp [R][P], gth [N][3] (eikonal gradients),
 *   sam [R][P] SAM logits or NULL (term off), in_mask [R] bytes or NULL (in-shape term off)
 *   -> terms [8] = {total, rgb_loss, eikonal, bce, in_shape, sam_mask, 0, 0} with
 *      total = rgb_loss + w_eik eikonal + w_bce bce + w_in in_shape + w_sam sam_mask   (the caller resolves the epoch schedules)
 *      d_rgb [R][3], d_acc [R], d_accp [R][P], d_gth [N][3] = d total / d (rgb, acc, accp, gth). */
typedef struct {
    const float *rgb, *rgb_gt, *acc, *accp, *gth, *sam;
    const unsigned char* in_mask;
    float *d_rgb, *d_acc, *d_accp, *d_gth, *terms;
    int n_rays, n_persons, n_eik;
    float w_eik, w_bce, w_in, w_sam, eps;
} MpLossArgs;
int mp_loss_fused(const MpLossArgs* args, void* stream);

/* ---- rays -------------------------------------------------------------------------------------
 * rend_util.get_camera_params (lib/utils/rend_util.py:45-87) + far sphere root (:131-147).
 *   uv [R][2], intrinsics [16], pose [16] -> dirs [R][3], far [R]; cam is pose[:3,3]. */
int mp_ray_setup(const float* uv, const float* intrinsics, const float* pose, int n_rays, float radius, float* dirs,
                 float* far, void* stream);
/* Ray / box test and ordered compaction (multiply.py:256-266): hit_index [<=R] ascending ray ids, *hit_count;
 * inv_index [R] = position in hit_index or -1.  group_size: rays of a convergence group without a hit get their
 * first ray as fallback (multiply.py:262-263 applied per chunk). scan_tmp: >= R+1 ints. */
int mp_ray_cull(const float* dirs, const float* pose, const float* obb, int n_rays, int group_size, int* hit_index,
                int* hit_count, int* inv_index, int* scan_tmp, void* stream);
/* The same, refined for eval-mode rendering without changing a pixel: a ray that passes the box but stays further than the
 * outlier radius 0.1 (deformer.py:49) from every vertex on [near, far[r]] carries only sdf = 4 samples (multiply.py:142-143);
 * when 1 - exp(-sigma(4) (far - near)) is exactly 0 in fp32 its pixel is the background's, exactly like a ray outside the
 * box, and it is dropped here.  cbound [MP_KNN_NC][4] = the bounding spheres of the posed vertex clusters (mp_knn_build), far [R]
 * from mp_ray_setup, beta = device scalar (density), near_ = the sampler's near bound. */
int mp_ray_cull_near(const float* dirs, const float* pose, const float* obb, const float* cbound, const float* far,
                     const float* beta, float near_, int n_rays, int group_size, int* hit_index, int* hit_count,
                     int* inv_index, int* scan_tmp, void* stream);
/* Explicit hit set (parity tests): fills hit_count / inv_index from a given ascending hit_index [n_hit]. */
int mp_ray_hits_from_index(const int* hit_index, int n_hit, int n_rays, int* hit_count, int* inv_index, void* stream);

/* ---- canonical warp ---------------------------------------------------------------------------
 * SMPLDeformer.forward(inverse=True) (deformer.py:19-50, 72-88): nearest posed vertex -> its skinning weights ->
 * x_c = (sum_j w_j T_j)^-1 x ; outlier = dist > 0.1.
 * The blended transform only depends on the nearest vertex: mp_blend_table evaluates it once per pose for every vertex,
 * table [n_verts][3][4] fp32, row r = (I[r][0..2], T[r][3] / T[3][3]) with T = sum_j skin_w[v][j] tfs[j], I = T[:3,:3]^-1
 * (skin_w [n_verts][24], tfs [24][16]); the warp kernels read x_c = I (x - c) and jinv = I from it (`blend_table`).
 * Points are either explicit (pts [max_rays][3], dirs, pose, hit_index, hit_count and z unused) or implicit samples of hit rays:
 * point id = k*n_s + s, x = cam + z[k*z_stride + s] * dirs[hit_index[k]], cam = pose[:3,3].
 *   mode 0 (training): every point is written to xc and appended to worklist.
 *   mode 1 (eval): outliers get sdf_out = 4 (multiply.py:142-143) and are NOT appended.
 *   ray_active [max_rays] per-ray flag or NULL; launch_active: device int, 0 = nothing to do (kernel exits), or NULL.  Outputs: xc [n][3], worklist, *work_count (atomic, must be 0). */
int mp_blend_table(const float* skin_w, const float* tfs, int n_verts, float* table, void* stream);
int mp_warp_inverse(const float* pts, const float* dirs, const float* pose, const int* hit_index, const int* hit_count,
                    const float* z, int z_stride, int n_s, int max_rays, const float* vsorted, const float* cbound,
                    const float* blend_table, int mode, const int* ray_active, const int* launch_active,
                    float* xc, unsigned char* outlier, float* sdf_out, int* worklist, int* work_count, void* bin_work,
                    void* stream);
/* bin_work (optional, mode 0 with implicit samples only; mp_warp_bin_work_bytes(max_rays * n_s) bytes, 16-byte aligned): the
 * points are first grouped by their nearest vertex cluster, so that the 64 points of a wave open the same few clusters -- a
 * training batch's random pixels otherwise scatter every wave over the whole body (50 of the then 108 clusters opened per wave).  Results
 * are identical (they go out by point id); only the order of the worklist changes. */
int mp_warp_bin_work_bytes(int n_points);
/* The same for the final samples of the shading pass (z rows hold n_s+1 depths).  eval_mode: outliers get sdf 4 and are
 * dropped from the worklist only when their compositing alpha 1-exp(-sigma(4) dt) is exactly 0 in fp32.
 * need_flag [id] (optional) = 1 for every point that was appended; nn_index [id] (optional) = the nearest posed vertex
 * whose skinning weights were used (the training backward needs it: deformer.py:47 detaches the weights). */
int mp_warp_inverse_shade(const float* dirs, const float* pose, const int* hit_index, const int* hit_count,
                          const float* z, int z_stride, int n_s, int max_rays, const float* vsorted, const float* cbound,
                          const float* blend_table, int eval_mode, const float* beta, float* xc,
                          unsigned char* outlier, unsigned char* need_flag, float* sdf_out, int* worklist,
                          int* work_count, int* nn_index, void* bin_work, void* stream);
/* Jacobian of forward skinning at canonical points (deformer.py:31-35 + multiply.py:625-641): nearest CANONICAL
 * vertex -> weights -> J = (sum_j w_j T_j)[:3,:3] -> jinv [id][9].
 * n_s > 0: points are the samples of the hit rays (id = k*n_s + s, as in mp_warp_inverse_shade) and only ids with
 * need[id] != 0 are processed; n_s == 0: explicit list of n_pts points (need, hit_count ignored).
 * seed [id] (optional, with verts_c [V][3] = the canonical vertices in original order): a vertex id per point whose
 * canonical distance bounds the search (the posed nearest vertex from mp_warp_inverse_shade); the result stays exact. */
int mp_warp_jacobian(const float* xc, const unsigned char* need, const int* hit_count, int max_rays, int n_s, int n_pts,
                     const float* vsorted_c, const float* cbound_c, const float* blend_table, float* jinv,
                     int* nn_index, const int* seed, const float* verts_c, void* stream);

/* ---- VolSDF error-bound sampler (ray_sampler.py:66-220), split at the SDF queries ---------------
 * State per hit ray k (row stride zmax = 640): zs/sdfs sorted samples and their sdf, nz count, znew/sdfnew [128]
 * the samples whose sdf is being queried, beta, done flag.  Per group: not_converged flags.
 * cfg = {N_samples, N_samples_eval, N_samples_extra, beta_iters, max_total_iters} ints, {eps, add_tiny, near} floats.
 * A convergence group = the hit rays whose ray id falls in one block of `group_size` consecutive rays of the call
 * (n_rays_total rays): the reference's `beta.max() > beta0` vote (ray_sampler.py:137) is taken per group, so that a
 * whole-frame call with group_size = pixel_per_batch reproduces the reference's chunked rendering loop exactly. */
typedef struct {
    int n_samples, n_samples_eval, n_samples_extra, beta_iters, max_total_iters;
    float eps, add_tiny, near_;
} MpSamplerCfg;
typedef struct {
    float* zs; float* sdfs; int* nz; float* znew; float* sdfnew; float* beta; int* ray_active;
    int* group_flag;   /* [max_total_iters+1][n_groups] */
    float* zfinal;     /* [max_rays][n_samples + n_samples_extra + 2] */
    int* iters;        /* [n_groups] iterations run (diagnostics) */
    int* any_active;   /* [max_total_iters+1] any_active[i] != 0: some ray still needs SDF queries in iteration i */
} MpSamplerState;
/* uniform start (ray_sampler.py:21-42, 70-76); t_rand [max_rays][n_eval] or NULL (eval: no jitter) */
int mp_sampler_init(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* far, const int* hit_index,
                    const int* hit_count, int max_rays, int group_size, int n_rays_total, const float* t_rand,
                    void* stream);
/* merge the queried samples, d*, error bound, beta bisection, group convergence vote (ray_sampler.py:89-137) */
int mp_sampler_bound(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* beta0, const int* hit_index,
                     const int* hit_count, int max_rays, int group_size, int n_rays_total, int iter, void* stream);
/* up-sample from the error-bound pdf, or draw the final samples and assemble zfinal (ray_sampler.py:139-209);
 * u_final [max_rays][n_samples] / extra_idx [max_total_iters][n_extra] supply the training randomness (row k-1 =
 * randperm(n_eval*k)[:n_extra], used when the sorted list holds n_eval*k depths), NULL = eval linspace */
int mp_sampler_resample(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* beta0, const float* far,
                        const int* hit_index, const int* hit_count, int max_rays, int group_size, int n_rays_total,
                        int iter, const float* u_final, const int* extra_idx, void* stream);

/* ---- compositing (multiply.py:425-480, 544-545, 590) ---------------------------------------------
 * Per ray: merge the persons' samples by t_end (ties: lower person first), Laplace density (density.py:20-29),
 * alpha = 1-exp(-sigma dt), T = exp(-exclusive cumsum), sums; bg transmittance = exclusive T of the last sample.
 *   per person p: inv_index[p] [R], z[p] [*][n_z] (n_z = S+1 depths, last = z_max), sdf[p] [*][S], rgb[p] [*][S][3],
 *   normal[p] [*][S][3]  (pointer tables live in device memory, P entries each)
 * Outputs [R]: fg_rgb[3], normal[3], acc, acc_person[P], bg_T; rgb = fg + bg_T*bg_rgb; fg_out = fg + bg_T. */
int mp_composite(int n_rays, int n_person, int n_z, const int* const* inv_index, const float* const* z,
                 const float* const* sdf, const float* const* rgb, const float* const* normal, const float* beta,
                 const float* bg_rgb, float* rgb_values, float* fg_rgb_values, float* normal_values, float* acc_map,
                 float* acc_person, float* bg_T, void* stream);


/* ---- fp32 training path (layer-wise forward with stash + hand-written backward) ----------------------------------
 * The reference trains in fp32 through torch autograd (multiply_model.py:192-217).  These entry points are the pieces of
 * the same computation as explicit forward / adjoint kernels; multiply_amd/train.py chains them inside one
 * torch.autograd.Function so that the reference's Loss and optimisers work unchanged.
 * Forward-mode row convention: a tensor of a network evaluated for P points with spatial tangents has 4P rows:
 * [0,P) values, [P,2P) d/dx, [2P,3P) d/dy, [3P,4P) d/dz.  "ld*" are row strides in floats. */
/* C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N] on rows < bias_rows) (optional ReLU); exact-fp32 MFMA */
int mp_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
               const float* bias, int bias_rows, int accumulate, int relu, void* stream);
/* The same product with every fp32 operand split into two bfloat16 halves (x = hi + lo) on its way into LDS and three
 * v_mfma_f32_16x16x32_bf16 per block product (hi.hi + hi.lo + lo.hi), fp32 accumulate: relative error ~2^-16 per product,
 * fp32 range; same arguments and epilogue. */
int mp_gemm_nt_bf16x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
               const float* bias, int bias_rows, int accumulate, int relu, void* stream);
/* C[M,N] += A[K,M]^T . B[K,N]  (C must be initialised; fp32 atomics) */
int mp_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* colsum,
               int colsum_rows, void* stream);
/* The same contraction on the split-bf16 path (see mp_gemm_nt_bf16x3); the 4 x 4 register transposition that turns the row-major
 * operands into k-contiguous MFMA fragments happens on the way into LDS.  colsum stays an exact fp32 sum. */
int mp_gemm_tn_bf16x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* colsum,
               int colsum_rows, void* stream);
/* Up to MP_TN_MAX_GROUPS such contractions in ONE launch (the weight gradients of all layers of a network): groups is a HOST
 * array; every group needs 16-byte aligned A, B, lda / ldb multiples of 4 and M, N multiples of 128.  Far fewer row slices per
 * output than one launch per contraction needs to fill the chip, hence far fewer fp32 atomics. */
#define MP_TN_MAX_GROUPS 24
typedef struct {
    const float* A;
    const float* B;
    float* C;
    float* colsum;       /* may be NULL */
    int lda, ldb, ldc, M, N, K, colsum_rows, pad_;
} MpTnGroup;
int mp_gemm_tn_bf16x3_grouped(const MpTnGroup* groups, int n_groups, void* stream);
/* Fourier features (embedders.py) of x [P][d_in] (d_in 3|4, L octaves) times `scale` into out[.][ld] at col0; fwd != 0 also
 * writes the three (d_in = 3) tangent row blocks */
int mp_tr_pe(const float* x, int d_in, int P, int L, int fwd, float scale, float* out, int ld, int col0, void* stream);
/* nn.Softplus(beta=100) layer on Z [rows][C] -> H at col0 (times scale); P > 0: forward mode (rows = 4P) */
int mp_tr_softplus_fwd(const float* Z, int ldz, int rows, int C, int P, float scale, float* H, int ldh, int col0,
                       void* stream);
int mp_tr_softplus_bwd(const float* Z, int ldz, int rows, int C, int P, float scale, const float* dH, int ldh, int col0,
                       float* dZ, int lddz, void* stream);
int mp_tr_relu_bwd(const float* H, int ldh, int rows, int C, const float* dH, int lddh, float* dZ, int lddz, void* stream);
/* last ImplicitNet layer Z8 [4P][257] -> sdf, normals (multiply.py:606,661) and the colour-net input XA [n][6] = [x_c, n];
 * the adjoint writes column 0 of a zero-initialised dZ8 (d sdf on value rows, d grad on tangent rows).
 * grad / dgrad (optional, [P][3]): d sdf / d x given separately instead of as tangent rows (reverse-over-reverse net); with them
 * Z8 / dZ8 may be NULL (the fused SDF kernels hand sdf / d sdf over as vectors: sdf_out is then not written, dsdf not copied) */
int mp_tr_shade_in_fwd(const float* Z8, int P, int n_pts, const float* xc, const float* jinv, float* XR, float* nrm,
                       float* sdf, const float* grad, void* stream);
int mp_tr_shade_in_bwd(const float* Z8, int P, int n_pts, const float* jinv, const float* dXR, const float* dsdf,
                       const float* dnrm_extra, float* dZ8, float* djinv, const float* grad, float* dgrad, void* stream);
/* eikonal samples (multiply.py:322-331): grad_theta [E][3] = d sdf/dx of points e0..e0+E of the batch */
int mp_tr_eik_fwd(const float* Z8, int P, int e0, int E, float* grad_theta, const float* grad, void* stream);
int mp_tr_eik_bwd(int P, int e0, int E, const float* dgrad_theta, float* dZ8, float* dgrad, void* stream);
int mp_tr_sigmoid_fwd(const float* Z, long long n, float* Y, void* stream);
int mp_tr_sigmoid_bwd(const float* Y, const float* dY, long long n, float* dZ, void* stream);
/* weight norm w = g v/|v| (networks.py:82-83): W [out][in] and its transpose WT [in][out]; adjoint dW -> dv, dg */
int mp_tr_wn_fwd(const float* v, const float* g, int out_dim, int in_dim, float* W, float* WT, void* stream);
int mp_tr_wn_bwd(const float* v, const float* g, int out_dim, int in_dim, const float* dW, float* dv, float* dg,
                 void* stream);
/* The same for many layers in ONE launch each: descs = DEVICE array of n_desc records, rows of all layers concatenated (row0 =
 * first row of a layer in that range, ascending; total_rows = their sum).  fwd reads v, g (NULL: plain weight) and writes W, WT
 * (WT may be NULL); bwd reads v, g and dW = acc_base + dW_off and writes dv = grad_base + dv_off, dg = grad_base + dg_off
 * (offsets in floats: the per-iteration accumulator / gradient buffers move, the table does not). */
typedef struct {
    const float* v;
    const float* g;
    float* W;
    float* WT;
    long long dW_off, dv_off, dg_off;
    int out_dim, in_dim, row0, pad_;
} MpWnDesc;
int mp_tr_wn_fwd_multi(const MpWnDesc* descs, int n_desc, int total_rows, void* stream);
int mp_tr_wn_bwd_multi(const MpWnDesc* descs, int n_desc, int total_rows, const float* acc_base, float* grad_base, void* stream);
/* per-call constant conditioning folded into the bias: b2 = b + W[:, c0:c0+n] vec ; adjoint dW[:, c0:c0+n] += db2 vec^T */
int mp_tr_hoist_fwd(const float* W, int out_dim, int in_dim, const float* b, int c0, int n, const float* vec, float* b2,
                    void* stream);
int mp_tr_hoist_bwd(const float* db2, int out_dim, int in_dim, int c0, int n, const float* vec, float* dW, void* stream);
int mp_tr_colsum(const float* dZ, int ld, int rows, int C, float* db, void* stream);
/* adjoint of mp_composite: d rgb_values / d acc_map / d acc_person -> d sdf[p], d rgb[p], d bg_rgb, d beta (+=) */
int mp_tr_composite_bwd(int n_rays, int n_person, int n_z, const int* const* inv_index, const float* const* z,
                        const float* const* sdf, const float* const* rgb, const float* beta, const float* bg_rgb,
                        const float* d_rgb_values, const float* d_acc, const float* d_acc_person, float* const* d_sdf,
                        float* const* d_rgb, float* d_bg_rgb, float* d_beta, void* stream);
/* background branch pieces (multiply.py:682-726) with per-ray depths zbg [R][NBG] */
int mp_tr_bg_points(const float* dirs, const float* cam, const float* zbg, int R, int NBG, float radius, float* pts,
                    void* stream);
int mp_tr_bg_comp_fwd(const float* sdf, const float* rgb, const float* zbg, int R, int NBG, float* out, void* stream);
int mp_tr_bg_comp_bwd(const float* sdf, const float* rgb, const float* zbg, int R, int NBG, const float* dout, float* dsdf,
                      float* drgb, void* stream);
/* pose optimisation (the reference back-propagates into BodyModelParams: smpl_pose / smpl_trans / smpl_shape):
 *   mp_tr_pe_bwd    : adjoint of the Fourier features (value + tangent rows) w.r.t. the point, dx [P][d_in] +=
 *   mp_tr_warp_bwd  : adjoint of x_c = (sum_j w_j tfs_j)^-1 x and of Jinv w.r.t. tfs (deformer.py:19-50, 72-88; the
 *                     skinning weights of the nearest posed / canonical vertex are constants), dtfs [24][16] +=
 *   mp_smpl_pose_bwd: adjoint of SMPLServer.forward's smpl_tfs (smpl.py:50-94, lbs.py:276-377) w.r.t. the 86 parameters
 *                     [scale, transl 3, thetas 72, betas 10] (betas through the rest joints only: j_shapedirs
 *                     [24][3][10] = J_regressor . shapedirs, or NULL); rest_joints [24][3] of the forward call */
int mp_tr_pe_bwd(const float* x, int d_in, int P, int L, int fwd, const float* dIN, int ld, float* dx, void* stream);
int mp_tr_warp_bwd(const float* xc, const float* dxc, const float* jinv, const float* djinv, const int* nn_posed,
                   const int* nn_cano, int n, const float* skin_w, const float* tfs, float* dtfs, void* stream);
int mp_smpl_pose_bwd(const int* parents, const float* params, const float* tfs_c_inv, const float* rest_joints,
                     const float* j_shapedirs, const float* dtfs, float* dparams, void* stream);
/* reverse-over-reverse SDF net (train.py ImplicitTrainRev): V = sigma'(Z) (.) U (U NULL: the row vector wrow) ;
 * its adjoint (dU, d sigma') ; dZ = sigma' dX scale + sigma'' dS ; Fourier-feature Jacobian products (3-D points). */
int mp_tr_sigmul(const float* Z, int ldz, long long rows, int C, const float* U, int ldu, const float* wrow, float scale,
                 float* V, int ldv, void* stream);
int mp_tr_rev_adj(const float* Z, int ldz, long long rows, int C, const float* U, int ldu, const float* wrow, float uscale,
                  const float* dV, int lddv, float* dU, int lddu, float* dS, int ldds, void* stream);
int mp_tr_dz(const float* Z, int ldz, long long rows, int C, const float* dX, int lddx, float scale, const float* dS, int ldds,
             float* dZ, int lddz, void* stream);
int mp_tr_pe_grad_fwd(const float* x, int P, int L, const float* G, int ldg, float* grad, void* stream);
int mp_tr_pe_grad_bwd(const float* x, int P, int L, const float* dgrad, const float* G, int ldg, float* dG, int lddg,
                      float* dx, void* stream);
int mp_tr_copy_cols(const float* src, int lds, int c0s, float* dst, int ldd, int c0d, long long rows, int C, float scale,
                    int accumulate, void* stream);

/* ---- layer-FUSED training kernels of the foreground ImplicitNet (csrc/tfuse.hip) -----------------------------------------
 * Replace, for the shipped network shape (8 x 256 softplus layers, skip connection into layer 4, 39 Fourier features, 257
 * outputs), the per-layer GEMM + element-wise chain of the reverse-over-reverse SDF net (networks.py:160-181 under
 * autograd with create_graph, multiply.py:620-661): a workgroup keeps 128 points on chip across the whole layer chain
 * (activations = MFMA B operands in registers, split-bf16 weights through an LDS ring by DMA) and writes only what the
 * weight-gradient contractions (mp_gemm_tn_bf16x3) and the adjoint sweep read back.
 *   mp_tf_sdf_pack : W[9] (effective fp32 weights, row-major [out][in]: in = 108 for layer 0, 256 otherwise), B[9] (biases;
 *                    B[0] = layer 0's bias with the conditioning hoisted in) -> wpack (pack_bytes of mp_tf_sdf_sizes: split
 *                    bf16 fragment tiles of W_l, W_l^T in consumption order) and bias_all [9][288] (pack-row order).
 *                    W and B are device arrays of 9 device pointers.
 *   arena          : floats as reported by mp_tf_sdf_sizes(P, &arena_floats, &pack_bytes).  Every [P][256] tensor has P + 1 rows
 *                    (row P: where the lanes of the last tile's missing points store); R1 = (P + 1) 256:
 *                      dZ(l) l=0..7 at l R1 (bwd)          V(l)  l=0..7 at (8+l) R1 (fwd)
 *                      X(l)  l=1..8 at (15+l) R1 (fwd: layer l's input)   dT(l) l=1..7 at (23+l) R1 (bwd)
 *                      U(l)  l=0..6 at (31+l) R1 ;  dS(l) l=0..7 at (38+l) R1
 *                      IN [P][39] at 46 R1 (Fourier features: caller, before fwd), dG [P][39] at 46 R1 + 39 P (caller, before bwd),
 *                      G [P][39] at 46 R1 + 78 P: d sdf / d Fourier features (fwd); then 256 floats the kernels scribble on
 *                    columns >= 217 of X_4 and dT_4 (the re-injected Fourier features of the skip connection, times 1/sqrt 2)
 *                    are NOT written by the kernels: the caller copies them from IN / dG (mp_tr_copy_cols).
 *                    Weight gradients (mp_gemm_tn_bf16x3[_grouped]): dW_l += dZ_l^T X_l + V_l^T dT_l.
 *   mp_tf_sdf_fwd  : feat [P+1][256] and sdf [P+1] (one pad row each) = the last layer's outputs (the reference's column 0 = sdf, columns 1.. = features),
 *                    G, and the stashes
 *   mp_tf_sdf_bwd  : dfeat [P][256], dsdf [P] (adjoints of feat / sdf), dG in BB0 -> dZ_l, dT_l stashes;
 *                    dw8 [256] += gradient of the last layer's sdf ROW (the gradient sweep's V_7 = sigma'_7 (.) w8 and the value
 *                    sweep's sum of d sdf . X_8), db8 [1] += gradient of its sdf BIAS */
int mp_tf_sdf_sizes(int P, long long* arena_floats, long long* pack_bytes);
int mp_tf_sdf_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream);
int mp_tf_sdf_fwd(const void* wpack, const float* bias_all, const float* w8, float* arena, int P, float* feat, float* sdf,
                  void* stream);
int mp_tf_sdf_bwd(const void* wpack, const float* w8, float* arena, int P, const float* dfeat, const float* dsdf, float* dw8,
                  float* db8, void* stream);
/* mp_tf_sdf_val: the VALUE sweep alone, sdf column only, nothing stashed -- mp_mlp_sdf's interface (worklist of point ids or NULL,
 * device-side count or NULL, sdf_out written at the point ids) at the training path's arithmetic: split-bfloat16 products, fp32
 * accumulation, fp32 softplus, Fourier features from sinf / cosf.  The sampler's queries with `Multiply.sampler_sdf_mode = 'bf16x3'`
 * (ray_sampler.py:85-88): the depths then agree with the fp32 reference to 1e-3 instead of 2e-2 (profiles/r05_sampler_precision.txt).
 * wpack / bias_all as written by mp_tf_sdf_pack (bias row 0 with the call's conditioning hoisted in). */
int mp_tf_sdf_val(const void* wpack, const float* bias_all, const float* xc, const int* worklist, const int* count, int max_count,
                  float* sdf_out, void* stream);

/* The NeRF++ BACKGROUND ImplicitNet (networks.py:126-208 as configured by confs/model: d_in 4, multires 10 -> 84 Fourier features,
 * the frame code hoisted into layer 0's bias, 8 x 256 softplus, skip at layer 4 = [172 | 84] / sqrt 2, 257 outputs, no weight norm;
 * caller multiply.py:514-541) on the same layer-fused skeleton, VALUE ONLY (its outputs feed the density and the colour net; no
 * spatial gradient is taken): mp_tf_bg_fwd = the value sweep, mp_tf_bg_bwd = its adjoint w.r.t. the activations.
 *   mp_tf_bg_pack : W[9] effective weights ([256][116], [256][256] x 2, [172][256], [256][256] x 4, [257][256]), B[9] (B[0] with the
 *                   frame code hoisted in) -> wpack (mp_tf_bg_sizes' pack_bytes), bias_all [9][288]
 *   arena         : R1 = (P + 1) 256 floats per [P][256] tensor (one pad row):  dZ(l) l = 0..7 at l R1 (bwd) | X(l) l = 1..8 at
 *                   (7 + l) R1 (fwd: layer l's input; columns 172.. of X(4) are the caller's: the re-injected features times
 *                   1/sqrt 2, mp_tr_copy_cols) | IN [P][84] at 16 R1 (the Fourier features: caller, before fwd)
 *   mp_tf_bg_fwd  : feat [P+1][256], sdf [P+1] (pad rows) = the last layer's outputs, and the X stashes
 *   mp_tf_bg_bwd  : dfeat [P][256], dsdf [P] -> the dZ stashes; dw8 [256] += the sdf row's weight gradient, db8 [1] += its bias'
 *   weight gradients (caller, mp_gemm_tn_bf16x3[_grouped]): dW_l += dZ_l^T X_l  (l = 0: X = IN; biases = column sums of dZ_l). */
int mp_tf_bg_sizes(int P, long long* arena_floats, long long* pack_bytes);
int mp_tf_bg_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream);
int mp_tf_bg_fwd(const void* wpack, const float* bias_all, float* arena, int P, float* feat, float* sdf, void* stream);
int mp_tf_bg_bwd(const void* wpack, const float* w8, float* arena, int P, const float* dfeat, const float* dsdf, float* dw8,
                 float* db8, void* stream);

/* The foreground RenderingNet ('pose_no_view', networks.py:263-312: 270 -> 4 x 256 ReLU -> 3, sigmoid) on the same skeleton:
 *   mp_tf_col_pack : W[5] (effective weights, layer 0 = [256][270]: columns 0..5 x_c / normal, 6..13 pose embedding, 14.. features),
 *                    B[5] (B[0] with the pose embedding hoisted in) -> wpack, bias_all [5][288]
 *   stash          : mp_tf_col_sizes floats; N1 = (n + 1) 256 (pad rows as above): H(l) l=0..3 at l N1 (ReLU outputs), dZ(l) l=0..3 at
 *                    (4+l) N1 (their adjoints)
 *   mp_tf_col_fwd  : feat [n][256] (row stride 256), xa [n][6] -> rgb [n][3], H stashes
 *   mp_tf_col_bwd  : drgb, rgb [n][3], w4 = W_4 [3][256] -> dfeat [n][256], dxa [n][6], dz4 [n][3] (adjoint of the last layer's
 *                    pre-activations), dZ stashes.  Weight gradients: dW_l = dZ_l^T H_{l-1} etc. by mp_gemm_tn_bf16x3. */
int mp_tf_col_sizes(int n, long long* stash_floats, long long* pack_bytes);
int mp_tf_col_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream);
int mp_tf_col_fwd(const void* wpack, const float* bias_all, float* stash, const float* feat, const float* xa, int n, float* rgb,
                  void* stream);
int mp_tf_col_bwd(const void* wpack, float* stash, const float* w4, const float* rgb, const float* drgb, int n, float* dfeat,
                  float* dxa, float* dz4, void* stream);

/* ---- in / off-surface flags (multiply.py:153-167; training, current_epoch < 250) -------------------------------
 * signed distance of canonical points to a triangle mesh given as face_verts [F][3][3] (= mesh_face_vertices_list[p]):
 * |d| = distance to the closest triangle, negative inside (ray-casting parity), then per ray (n_s consecutive points)
 * off = min > threshold, in = min <= 0.  kaolin 0.13 in the reference (third party): restated, see csrc/mesh.hip. */
int mp_mesh_signed_distance(const float* pts, int n, const float* face_verts, int n_faces, float* sdist, void* stream);
int mp_mesh_ray_flags(const float* sdist, int n_rays, int n_s, float threshold, unsigned char* off, unsigned char* in,
                      void* stream);

/* ---- general deformer queries (deformer.py:19-50, 72-88) for K <= 8 nearest vertices; the render / training path uses
 * the fused K = 1 kernels above.  weights [n][24] = sum_k conf_k skin_w[idx_k], conf = exp(-min(d^2,4)) normalised;
 * outlier (optional) = sqrt(min(d_0^2, 4)) > 0.1.  mp_skinning: x' = (sum_j w_j tfs_j) x, or its inverse. */
int mp_query_weights(const float* pts, int n, const float* verts, int n_verts, const float* skin_w, int K, float* weights,
                     unsigned char* outlier, void* stream);
int mp_skinning(const float* pts, const float* weights, int n, const float* tfs, int inverse, float* out, void* stream);

/* ---- input producer (code/lib/datasets/Hi4D.py:8-20 bilinear_interpolation, :59-88 weighted_sampling) --------------
 * Sub-pixel samples of a frame that is RESIDENT in device memory: img [H][W][3] bytes (RGB), mask [H][W] bytes (the sum of
 * the per-person masks, Hi4D.py:236-245), extra [H][W][n_extra] fp32 (optional, e.g. the SAM mask), pos [n][2] doubles
 * (row, col) as the reference draws them (row < H-1, col < W-1).  Outputs (each optional): rgb [n][3] = interpolated
 * img/255, uv [n][2] = (col, row), mask_out [n], extra_out [n][n_extra]; double arithmetic, rounded once to fp32. */
int mp_sample_pixels(const unsigned char* img, const unsigned char* mask, const float* extra, int n_extra,
                     const double* pos, int n, int H, int W, float* rgb, float* uv, float* mask_out, float* extra_out,
                     void* stream);

/* ---- canonical-mesh extraction (code/lib/libmise/mise.pyx: MISE.__cinit__ :33-78, update :80-97, query :99-120,
 * to_dense :122-154, subdivide_voxels :172-222; driver code/lib/utils/mesh.py:78-131) on a DENSE lattice:
 * resolution R = res0 << depth, n = R + 1 points per axis.  state [n^3] bytes (0 no grid point, 1 unknown, 2 known),
 * val [n^3] fp32, vox / pos / neg [sum_l (res0<<l)^3, l = 0..depth] bytes (voxel tree: 0 absent, 1 leaf, 2 split; and
 * the "touches a value >= / <= threshold" flags, which the caller zeroes before every mp_mise_refine).
 *   mp_mise_init    : level-0 voxels are leaves, their corners are unknown grid points
 *   mp_mise_collect : packs the unknown grid points (x, y, z ints, unspecified order) -> out_xyz, *count (device int,
 *                     zeroed by the caller; may exceed max_out, then only max_out were written)
 *   mp_mise_scatter : stores the values of `count` queried points and marks them known
 *   mp_mise_refine  : one update step: mark leaves from ALL known points, split the active ones (*n_split += splits)
 *   mp_mise_fill    : to_dense: holes inherit the previous value along x, then y, then z
 * Marching cubes over the dense values (skimage.measure.marching_cubes in the reference, third party): tri_table
 * [256][16] ints (edge triples, -1 terminated; corner / edge numbering of csrc/mise.hip), case bit k = corner k below
 * `level`; mp_mc_count -> triangles per cube [(n-1)^3]; mp_mc_emit with exclusive-scan offsets -> verts [3T][3] in
 * lattice units and edge_id [3T] (equal ids = same mesh vertex). */
int mp_mise_init(int res0, int depth, unsigned char* state, unsigned char* vox, void* stream);
int mp_mise_collect(int n, const unsigned char* state, int* count, int max_out, int* out_xyz, void* stream);
int mp_mise_scatter(int n, const int* xyz, const float* values, int count, unsigned char* state, float* val, void* stream);
int mp_mise_refine(int res0, int depth, float threshold, unsigned char* state, const float* val, unsigned char* vox,
                   unsigned char* pos, unsigned char* neg, int* n_split, void* stream);
int mp_mise_fill(int n, unsigned char* state, float* val, void* stream);
int mp_mc_count(const float* val, int n, float level, const int* tri_table, int* counts, void* stream);
int mp_mc_emit(const float* val, int n, float level, const int* tri_table, const long long* offsets, float* verts,
               long long* edge_id, void* stream);

/* ---- per-person mesh z-buffer (code/lib/model/render.py:64-66, 134-157 render_multiple_depth_map over pytorch3d's
 * MeshRasterizer with blur_radius 0; callers multiply_model.py:396, :634, :875).  verts [n_verts][3] world space, faces
 * [n_faces][3] ints; cam_host = 16 floats IN HOST MEMORY: R row-major (OpenCV world -> camera), T, fx, fy, cx, cy; a pixel
 * (row, col) is sampled at (col + 0.5, row + 0.5); faces with a vertex nearer than z_clip are dropped.  keys [H*W] and big
 * [n_faces + 1] are scratch.  zbuf [H][W] = camera-space depth of the nearest covering face (-1: none), pix_to_face [H][W]
 * (optional, -1: none), bary [H][W][3] (optional) perspective-correct barycentrics of the winner.  pytorch3d is third
 * party and unpinned: restated, see csrc/raster.hip. */
int mp_raster_zbuf(const float* verts, int n_verts, const int* faces, int n_faces, const float* cam_host, float z_clip,
                   int H, int W, unsigned long long* keys, int* big, float* zbuf, int* pix_to_face, float* bary,
                   void* stream);

/* ---- soft silhouette render (code/lib/model/render.py:79-105, 121-133 softrender_multiple_meshes: pytorch3d's blurred
 * MeshRasterizer + SoftPhongShader under white ambient light = softmax_rgb_blend of the interpolated vertex colours; consumer
 * multiply_model.py:636-637, :721).  Two calls, because the length of the tile lists is only known after the count:
 *   mp_raster_soft_bins  faces -> 8 x 8-pixel tiles: tile_n [T] scratch (T = ceil(H/8) ceil(W/8)), offsets [T + 1] exclusive
 *                        offsets of the tile lists, offsets[T] = total entries (-1: more than INT_MAX)
 *   mp_raster_soft       tile_n / offsets as mp_raster_soft_bins left them for the SAME mesh, camera, image size and
 *                        blur_radius (tile_n all zero again: it counts the list fill); list [offsets[T]] scratch, colors
 *                        [n_verts][3]; image [H][W][4] = RGB over the background + the
 *                        silhouette 1 - prod(1 - prob); sel [H][W][faces_per_pixel] (optional): the selected faces of every
 *                        pixel in no particular order, -1 padded.  faces_per_pixel <= 100.  cam_host / background_host
 *                        (3 floats) IN HOST MEMORY; camera and pixel conventions of mp_raster_zbuf; distances are pytorch3d
 *                        NDC units (the shorter image side spans [-1, 1]), blur_radius a SQUARED distance.
 * pytorch3d is third party and unpinned: restated, see csrc/raster.hip. */
int mp_raster_soft_bins(const float* verts, int n_verts, const int* faces, int n_faces, const float* cam_host, float z_clip,
                        int H, int W, float blur_radius, int* tile_n, int* offsets, void* stream);
int mp_raster_soft(const float* verts, int n_verts, const int* faces, int n_faces, const float* colors, const float* cam_host,
                   float z_clip, int H, int W, float sigma, float gamma, float blur_radius, int faces_per_pixel, float znear,
                   float zfar, const float* background_host, int* tile_n, const int* offsets, int* list, float* image, int* sel,
                   void* stream);

/* library / device info: returns the gfx arch string compiled in, and checks the current device */
const char* mp_arch(void);
int mp_device_ok(void);

#ifdef __cplusplus
}
#endif
#endif
