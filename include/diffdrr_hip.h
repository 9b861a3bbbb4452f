/*
 * diffdrr_hip.h -- C ABI of libdiffdrr_hip.so: the MI355X (gfx950) renderers
 * behind DiffDRR's renderer seam.
 *
 * The reference has no FFI: its seam is the Python attribute `DRR.renderer`
 * (an nn.Module chosen at /root/reference/diffdrr/drr.py:94-101 and called at
 * drr.py:209-224 as `renderer(density, source, target, img, **kwargs)`).  The
 * entry points below are what a ctypes binding inside
 * diffdrr/renderers.py would call in place of the tensor programs at
 *   renderers.py:34-91   Siddon.forward      -> ddrr_siddon_forward (+ _channels)
 *   renderers.py:205-254 Trilinear.forward   -> ddrr_trilinear_forward
 * and in place of torch-autograd's backward of those programs
 *   (grid_sampler_3d_backward, SortBackward, ~30 elementwise backward ops)
 *                                            -> ddrr_siddon_backward_rays,
 *                                               ddrr_siddon_backward_volume,
 *                                               ddrr_trilinear_backward.
 * INTEGRATION.md shows that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (HIP, gfx950), dense and
 *    C-contiguous; nothing is copied, pointers are borrowed for the call only;
 *  - volume V[dx][dy][dz]  (z fastest: reference drr.py:81-85 `density`;
 *    the reference only *views* it permuted, renderers.py:160);
 *  - source (B, src_n, 3) with src_n == 1 (one source per pose, the DRR case:
 *    detector.py:152) or src_n == N; target (B, N, 3); both in voxel-index
 *    coordinates (after drr.py:204-205);
 *  - img (B, N): ray length in world units (drr.py:201), NULL == all ones;
 *  - out (B, N) (the reference's (B, 1, N) without the singleton);
 *  - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous
 *    and never synchronise the device; the library keeps no state on the device
 *    and, on the host, only a thread-local error string, each device's CU count
 *    and "kernel attribute set" flags;
 *  - `launch_ws` of the *_bricks entry points: a caller-owned device buffer of
 *    ddrr_brick_launch_workspace_bytes(dx, dy, dz) bytes, 16-byte aligned, that
 *    belongs to THIS call until the call's work on `stream` is done (its brick
 *    counter and the hand-out order of the bricks live there; contents need not
 *    be initialised).  Calls that may run concurrently -- other streams, a
 *    captured graph next to eager launches -- must be given different buffers;
 *  - return value: 0 on success, otherwise a hipError_t (or -1 for an
 *    argument error); ddrr_last_error() describes the last failure.
 *
 * det_h/det_w/tile_h/tile_w are a pure performance hint: when the N rays of a
 * pose are a row-major det_h x det_w detector grid (detector.py:126) each
 * 64-lane wavefront renders a tile_h x tile_w tile of pixels (tile_h * tile_w
 * == 64); pass zeros for an arbitrary ray list.  Results do not depend on it.
 */
#ifndef DIFFDRR_HIP_H
#define DIFFDRR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DDRR_ABI_VERSION 33

#define DDRR_REDUCE_SUM 0 /* reducefn="sum"  renderers.py:176-177 */
#define DDRR_REDUCE_MAX 1 /* reducefn="max"  renderers.py:178-179 */

#define DDRR_LOOKUP_STEP 0          /* mode="nearest", align_corners=False: exact voxel stepping */
#define DDRR_LOOKUP_MID_NEAREST 1   /* nearest lookup at each segment midpoint (any align_corners) */
#define DDRR_LOOKUP_MID_TRILINEAR 2 /* Siddon(mode="bilinear"): trilinear lookup at midpoints */

#define DDRR_SIDDON_AUX 8 /* floats per ray in the forward record used by the backward */

/* layouts of the forward record handed to ddrr_siddon_backward_rays */
#define DDRR_AUX_INTERLEAVED 0 /* (B, N, 8): ddrr_siddon_forward */
#define DDRR_AUX_BLOCKED 1     /* (ceil(B N / 16), 80): I, S0x, S0z, S1x, S1z of 16 consecutive rays per
                                * block, one or two planes of 8 / 16 rays per 64-byte line
                                * (csrc/record_layout.h): ddrr_siddon_forward_bricks */
#define DDRR_REC_BLOCK_RAYS 16
#define DDRR_REC_BLOCK_FLOATS 80
#define DDRR_AUX_PACKED 2      /* (7, B, N): fixed-point record, csrc/record_pack.h: ddrr_siddon_forward_bricks(record_vmax > 0) */
#define DDRR_PACKED_AUX_PLANES 7
#define DDRR_TRI_AUX_PLANES 7   /* sum T, sum dT_xyz, sum alpha dT_xyz: ddrr_trilinear_forward_bricks */

int ddrr_abi_version(void);
const char *ddrr_last_error(void);

/* Siddon.forward, mask=None (renderers.py:34-76).  aux: NULL, or (B, N, 8)
 * record for ddrr_siddon_backward_rays (only with DDRR_LOOKUP_STEP).
 * n_vox: NULL, or (B, N) int32 receiving the number of positive-length
 * in-volume segments of each ray (the algorithmic voxel count). */
int ddrr_siddon_forward(const float *volume, int dx, int dy, int dz, const float *source,
                        int src_n, const float *target, const float *img, int B, int N,
                        float voxel_shift, float eps, int reduce_mode, int lookup_mode,
                        int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                        float *out, float *aux, int *n_vox, void *stream);

/* Volume-stationary form of ddrr_siddon_forward for the DRR case (sum, nearest,
 * align_corners=False, one source per pose, the N = det_h * det_w rays of a pose being a
 * row-major target grid that is an affine image of the pixel lattice, detector.py:126-153): one
 * workgroup per 32^3 brick staged in LDS traces every ray of every pose through it and
 * adds the partial integrals to `out` (zero-filled by the call) with fp32 atomics.  The
 * volume is read from HBM once per call, whatever B.  The image equals
 * ddrr_siddon_forward's up to fp32 summation order (which is not deterministic here).
 * out: (B, N); may be NULL when aux is given (the record alone: ddrr_siddon_ncc_forward forms the image).
 * aux: NULL, or the blocked backward record, DDRR_REC_BLOCK_FLOATS * ceil(B N / DDRR_REC_BLOCK_RAYS)
 * floats (zero-filled and accumulated by the call) for ddrr_siddon_backward_rays /
 * ddrr_siddon_backward_pose (aux_layout = DDRR_AUX_BLOCKED): a run of 8 adjacent pixels adds whole
 * 64-byte lines to it (the record's atomics are executed at the memory side, per line).
 * record_vmax: 0, or max |volume| (> 0): aux is then (DDRR_PACKED_AUX_PLANES, B, N) and receives
 * the record in 32-bit fixed point, two fields per 64-bit integer atomic (3 atomics per ray and
 * brick instead of 5; exact, order-independent sums; resolution 2 vmax (dx+dy+dz+3) / 2^30 per
 * brick piece) for aux_layout = DDRR_AUX_PACKED.
 * brick_storage: how a brick is held in LDS.  DDRR_BRICKS_F32: the volume's own fp32 values,
 * 32^3 voxels per brick.  DDRR_BRICKS_Q16: 16-bit block quantisation, one (min, step) pair per
 * brick (V ~ min + q step, q = 0 .. 65535; all arithmetic stays fp32), which lets a brick of
 * twice the volume (32 x 32 x 64) fit a CU's LDS: fewer (ray, brick) pairs, 6-7 % faster.
 * ONLY bricks that keep their values are quantised: the error of a stored voxel is at most
 * (max - min of the brick) / 131070 in absolute terms, so a brick qualifies if that range is at
 * most 12x its LEVEL, the smallest mean |V| of any of its 4 x 4 x 4 blocks (over the non-zero
 * voxels when the brick's minimum is 0, which is stored exactly) -- every voxel is then within
 * 9.2e-5 of the mean of the dimmest block a ray can cross.  Every other brick (one bright voxel
 * among dim ones: a metal marker, contrast agent, un-normalised HU; inf / NaN; |values| or a
 * range outside 2^-60 .. 2^40) is rendered from the volume's own fp32 values, as two
 * 32 x 32 x 32 halves, in the same launch: results of DDRR_BRICKS_Q16 are within the 1e-4 of
 * DDRR_BRICKS_F32's whatever the volume holds (tests/test_brick_storage_guard.py).
 * DDRR_BRICKS_Q16_PACKED: the same bricks, additionally kept in the workspace as they lie in
 * LDS (padding included, brick after brick; + 52 % of the volume's bytes): a later call stages
 * a brick with a straight 16-byte copy of half the bytes instead of converting the fp32 volume
 * again -- what a volume that is rendered many times wants, most of all with few poses per
 * launch, where staging is most of the launch.
 * (The 16-bit walk carries alpha pre-scaled by 2^64: rays with |alpha| >= 2^63 inside the volume
 * -- |t - s + eps| < ~1e-16 on an axis, which eps = 1e-8 excludes -- overflow to inf / NaN there,
 * where fp32 bricks would return a finite value.)
 * brick_ranges: NULL for DDRR_BRICKS_F32; otherwise a caller-owned workspace of
 * ddrr_brick_workspace_bytes(dx, dy, dz, brick_storage) bytes, 16-byte aligned: a 256-byte
 * header (int32 word 0: bricks on the fp32 path, word 1: bricks), the bricks' (min, max) and
 * fp32 flags (and the packed bricks).  With ranges_valid = 0 the call fills it first (one pass
 * over the volume, ~0.1 ms at 512^3; ~0.35 ms with the packed bricks), with ranges_valid = 1 it
 * trusts what an earlier call for the SAME volume contents and the same brick_storage left
 * there (a registration or a pose sweep renders one volume thousands of times) -- a caller
 * must only pass 1 after a call with 0 and B > 0 has returned 0 for these contents.
 * ranges_valid | DDRR_BRICKS_CLEARED (ABI 30): the caller has already zeroed what the launch's atomics
 * add to -- `aux` if given, else `out` -- AND the first 16 bytes of launch_ws (the brick counter),
 * on this stream (ddrr_pose_raygen_forward can do both in its own launch): the call then clears
 * nothing, one launch less in front of a render of a pose or two.
 * (A volume with only a few double bricks per CU balances badly: 256^3 is 6 % faster on fp32
 * bricks; the Python layer chooses, diffdrr_amd/renderers.py.)
 * Any volume shape and any float-aligned volume pointer take every brick_storage: the kernels
 * stage quads of four voxels along z with one 16-byte load from a dword-aligned address (the
 * reference's example CT has 133 slices).
 * launch_ws: see the conventions at the top of this file. */
#define DDRR_BRICKS_F32 0
#define DDRR_BRICKS_Q16 1
#define DDRR_BRICKS_Q16_PACKED 2
#define DDRR_BRICKS_CLEARED 2 /* (a bit of ranges_valid) */
long ddrr_brick_workspace_bytes(int dx, int dy, int dz, int brick_storage);
long ddrr_brick_launch_workspace_bytes(int dx, int dy, int dz);
int ddrr_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                               const float *target, const float *img, int B, int det_h,
                               int det_w, float voxel_shift, float eps, float *out, float *aux,
                               float record_vmax, int brick_storage, float *brick_ranges,
                               int ranges_valid, void *launch_ws, void *stream);
/* The same render of a SUBSAMPLE of the detector grid (ABI 31; reference drr.py:36-39, 142-147 and
 * detector.py:134-137: `p_subsample` keeps a random subset of the pixels and `reshape_subsampled_drr`
 * scatters the rendered values into zeros): pixel_mask holds one bit per pixel of the det_h x det_w
 * grid -- bit n % 32 of word n / 32 set = pixel n is rendered -- shared by the B poses, ceil(det_h det_w
 * / 32) words, 4-byte aligned; NULL = every pixel (= ddrr_siddon_forward_bricks).  source / target /
 * img are still the whole grid's (the kernels cull candidates with the grid's affine model, then
 * drop the pixels whose bit is clear before any ray is loaded or walked); out (B, N) / aux hold zeros
 * at the other pixels: `out` IS the scattered image the reference builds. */
int ddrr_siddon_forward_bricks_masked(const float *volume, int dx, int dy, int dz, const float *source,
                                      const float *target, const float *img, int B, int det_h,
                                      int det_w, float voxel_shift, float eps, float *out, float *aux,
                                      float record_vmax, int brick_storage, float *brick_ranges,
                                      int ranges_valid, void *launch_ws, const unsigned *pixel_mask,
                                      void *stream);

/* Volume gradient for the DRR case of ddrr_siddon_forward_bricks (reduce sum), also
 * volume-stationary: each 32^3 brick of g_volume is accumulated in LDS (ds_add_f32) from
 * every ray of every pose that crosses it and STORED once -- no global atomics, no
 * zero-fill by the caller; g_volume (dx, dy, dz) is fully written.  Replaces
 * grid_sampler_3d_backward (nearest) behind renderers.py:159-164 like
 * ddrr_siddon_backward_volume, which remains the entry for arbitrary ray lists. */
int ddrr_siddon_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                       const float *target, const float *img,
                                       const float *grad_out, int B, int det_h, int det_w,
                                       float voxel_shift, float eps, float *g_volume,
                                       void *launch_ws, void *stream);

/* Pose/ray gradients of ddrr_siddon_forward from its aux record (aux_layout says which
 * forward wrote it): what autograd of renderers.py:94-113 + :70-71 returns.  g_source is per ray (B, N, 3) (sum
 * over rays for a broadcast source); g_target (B, N, 3); g_img (B, N), the
 * gradient w.r.t. `img` (NULL to skip, e.g. stop_gradients_through_grid_sample). */
int ddrr_siddon_backward_rays(const float *aux, int aux_layout, const float *grad_out,
                              const float *source, int src_n, const float *target,
                              const float *img, int B, int N, float eps, int reduce_mode,
                              float *g_source, float *g_target, float *g_img, void *stream);

/* Volume gradient of ddrr_siddon_forward (replaces grid_sampler_3d_backward,
 * nearest): ACCUMULATES grad_out * img * dalpha into g_volume[dx][dy][dz] with
 * hardware fp32 atomics; the caller zero-fills g_volume. */
int ddrr_siddon_backward_volume(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int reduce_mode, int det_h, int det_w, int tile_h, int tile_w,
                                float *g_volume, void *stream);

/* Backward of ddrr_siddon_forward for the midpoint lookups (DDRR_LOOKUP_MID_NEAREST: mode
 * "nearest" with align_corners = 1; DDRR_LOOKUP_MID_TRILINEAR: Siddon(mode="bilinear")), reduce
 * sum: autograd of renderers.py:57-71 incl. the path through the midpoint positions
 * (grid_sampler_3d_backward).  Outputs per ray, any may be NULL; g_volume ACCUMULATED with fp32
 * atomics (the caller zero-fills). */
int ddrr_siddon_backward_midpoint(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_out, int B, int N, float voxel_shift,
                                  float eps, int lookup_mode, int align_corners, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *stream);

/* Siddon.forward with a mask (mask_to_channels, renderers.py:77-89): labels is
 * the (dx, dy, dz) uint8 label map, out is (B, C, N) and is fully written. */
int ddrr_siddon_forward_channels(const float *volume, const unsigned char *labels, int dx, int dy,
                                 int dz, const float *source, int src_n, const float *target,
                                 const float *img, int B, int N, int C, float voxel_shift,
                                 float eps, int det_h, int det_w, int tile_h, int tile_w,
                                 float *out, void *stream);

/* The same render for the DRR case (one source per pose, row-major det_h x det_w target grid)
 * on the volume-stationary brick kernel: the label travels in the low 8 bits of the staged
 * voxel word, the value keeps a 16-bit mantissa (rounded to nearest, 2^-17 relative per voxel:
 * channel sums agree with the plain render to ~1e-5 of the image scale).  out (B, C, N) is
 * fully written; B * C * N < 2^30, N < 2^22.  The backward is ddrr_siddon_backward_channels. */
int ddrr_siddon_forward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                        int dy, int dz, const float *source, const float *target,
                                        const float *img, int B, int det_h, int det_w, int C,
                                        float voxel_shift, float eps, float *out, void *launch_ws,
                                        void *stream);
/* The same render from a volume of READY-PACKED words (ABI 31): ddrr_channel_words writes, for every
 * voxel, the word the channel kernel stages -- the value rounded to a 16-bit mantissa in the upper 24
 * bits, the label in the low byte, labels >= C as the value 0 under label 0 -- into `words` (n_voxels
 * floats); ddrr_siddon_forward_channels_bricks_words then stages a brick with straight 16-byte copies
 * (no label loads, no packing): for a (volume, label map) pair that is rendered many times.
 * Call ddrr_channel_words in front of EVERY such render: it compares a fingerprint of volume and
 * labels (1024 voxels each, kept in `state`) and ends after a few microseconds if nothing changed;
 * if something did -- or with force != 0: the first call, or a change the caller knows of -- it packs
 * the words again.  state: ddrr_channel_words_state_bytes() bytes, 4-byte aligned, caller-owned, ZERO
 * when first handed over (int32 word 1 counts the repacks).  Same result as
 * ddrr_siddon_forward_channels_bricks. */
long ddrr_channel_words_state_bytes(void);
int ddrr_channel_words(const float *volume, const unsigned char *labels, long n_voxels, int C, float *words,
                       void *state, int force, void *stream);
int ddrr_siddon_forward_channels_bricks_words(const float *words, int dx, int dy, int dz, const float *source,
                                              const float *target, const float *img, int B, int det_h,
                                              int det_w, int C, float voxel_shift, float eps, float *out,
                                              void *launch_ws, void *stream);

/* Backward of the channel render w.r.t. the RAYS for the DRR case, on the volume-stationary
 * bricks (replaces ScatterAddBackward . SortBackward of renderers.py:77-89 for source / target /
 * img): for grad_out (B, C, N) it writes the blocked backward record (DDRR_AUX_BLOCKED,
 * DDRR_REC_BLOCK_FLOATS * ceil(B N / DDRR_REC_BLOCK_RAYS) floats, zero-filled by the call) of
 * the volume weighted by each voxel's own incoming gradient, W(x) = V(x) grad_out[b, label(x), n]
 * -- labels >= C weigh 0 -- so that ddrr_siddon_backward_rays(aux, DDRR_AUX_BLOCKED, ones (B, N),
 * ...) returns d/d source, d/d target and d/d img of sum_c grad_out_c out_c.  The weight of a
 * label run is gathered from grad_out when a ray's label changes inside a brick.
 * B * C * N < 2^30, N < 2^22.  (The volume gradient: ddrr_siddon_backward_channels_volume_bricks.) */
int ddrr_siddon_backward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                         int dy, int dz, const float *source, const float *target,
                                         const float *grad_out, int B, int det_h, int det_w, int C,
                                         float voxel_shift, float eps, float *aux, void *launch_ws,
                                         void *stream);

/* The same for the marcher's channel render (ddrr_trilinear_forward_channels_bricks): the planar
 * record of ddrr_trilinear_forward_bricks (DDRR_TRI_AUX_PLANES planes of B N floats, zero-filled
 * by the call) with every sample's T and dT multiplied by grad_out[b, label of its nearest
 * voxel, n] (labels >= C weigh 0), so that ddrr_trilinear_backward_rays(aux, ones (B, N), ...)
 * returns d/d source, d/d target, d/d img and d/d alphamin, alphamax of sum_c grad_out_c out_c.
 * The brick holds the volume's own fp32 values; the labels are read from the label map.
 * B * C * N < 2^30, N < 2^22.  (The volume gradient: ddrr_trilinear_backward_channels_volume_bricks.) */
int ddrr_trilinear_backward_channels_bricks(const float *volume, const unsigned char *labels,
                                            int dx, int dy, int dz, const float *source,
                                            const float *target, const float *grad_out, int B,
                                            int det_h, int det_w, int C, float voxel_shift,
                                            float eps, int n_points, const float *alphamin,
                                            const float *alphamax, float *aux, void *launch_ws,
                                            void *stream);

/* Backward of the channel render w.r.t. the VOLUME for the DRR case, on the volume-stationary
 * bricks (the grid_sampler_3d_backward of renderers.py:77-89's gather, weighted per channel):
 * g_volume[x] = sum over poses and rays of len(ray in voxel x) img grad_out[b, label(x), n], labels
 * >= C weigh 0.  STORED, every voxel exactly once (no zero fill by the caller, unlike
 * ddrr_siddon_backward_channels).  The brick in LDS is the accumulator, 24-bit fixed point over the
 * voxel's label in the word's low byte (the label plane has no room of its own next to it); poses
 * whose source lies in or next to the volume -- no bound on a voxel's sum -- accumulate in fp32
 * and look the labels up in the label map.  B * C * N < 2^30, N < 2^22. */
int ddrr_siddon_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy, int dz,
                                                const float *source, const float *target,
                                                const float *img, const float *grad_out, int B,
                                                int det_h, int det_w, int C, float voxel_shift,
                                                float eps, float *g_volume, void *launch_ws,
                                                void *stream);

/* The same for the marcher's channel render (renderers.py:242-252): g_volume[x] = sum over poses,
 * rays and samples of (trilinear weight of corner x) img step grad_out[b, label of the sample's
 * nearest voxel, n], on the owner bricks of ddrr_trilinear_backward_volume_bricks (31-bit fixed-point
 * accumulators; the labels are read from the label map).  STORED. */
int ddrr_trilinear_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy,
                                                   int dz, const float *source, const float *target,
                                                   const float *img, const float *grad_out, int B,
                                                   int det_h, int det_w, int C, float voxel_shift,
                                                   float eps, int n_points, const float *alphamin,
                                                   const float *alphamax, float *g_volume,
                                                   void *launch_ws, void *stream);

/* Backward of ddrr_siddon_forward_channels: what autograd of renderers.py:77-89 (scatter_add
 * of the weighted segments into channels) returns for grad_out (B, C, N).  Outputs as in
 * ddrr_siddon_backward_rays (per ray; any may be NULL); g_volume: NULL, or ACCUMULATED with
 * fp32 atomics (the caller zero-fills). */
int ddrr_siddon_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                  int dy, int dz, const float *source, int src_n,
                                  const float *target, const float *img, const float *grad_out,
                                  int B, int N, int C, float voxel_shift, float eps, int det_h,
                                  int det_w, int tile_h, int tile_w, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *stream);

/* The materialised per-segment tensor the reference hands to a CALLABLE reducefn
 * (renderers.py:70-71, 175-183; notebooks/tutorials/introduction.ipynb:506-529): terms is
 * (B, M - 1, N) with M = dx + dy + dz + 3 sorted plane crossings -- i.e. the reference's
 * (B, N, M - 1) tensor transposed, so that rays write coalesced; term k = img * V * dalpha of the
 * k-th interval between consecutive crossings, 0 outside the volume (mode "nearest",
 * align_corners = 0).  _backward: autograd of it for grad_terms (B, M - 1, N); outputs as in
 * ddrr_siddon_backward_channels. */
int ddrr_siddon_segments(const float *volume, int dx, int dy, int dz, const float *source,
                         int src_n, const float *target, const float *img, int B, int N,
                         float voxel_shift, float eps, float *terms, void *stream);
int ddrr_siddon_segments_backward(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_terms, int B, int N, float voxel_shift,
                                  float eps, float *g_source, float *g_target, float *g_img,
                                  float *g_volume, void *stream);

/* The batch-global marching range Trilinear.forward computes when alphamin / alphamax are not
 * given (renderers.py:220-223 over _get_alpha_minmax :124-140): range2[0] = min over all rays of
 * the first intersection with the volume (clipped to >= 0), range2[1] = max of the last
 * (clipped to <= 1).  One pass over the rays; for ray lists that need no gradient through the
 * range (the module falls back to tensor ops when they do). */
int ddrr_trilinear_alpha_range(const float *source, int src_n, const float *target, int B, int N,
                               int dx, int dy, int dz, float voxel_shift, float eps,
                               float *range2, void *stream);

/* Trilinear.forward, mask=None (renderers.py:205-241).  alphamin/alphamax are
 * DEVICE scalars (renderers.py:220-223 evaluated by the caller, or the
 * caller's own values); mode_nearest selects Trilinear(mode="nearest"). */
int ddrr_trilinear_forward(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int reduce_mode,
                           int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                           float *out, void *stream);

/* Trilinear.forward with a mask (mask_to_channels, renderers.py:242-252): every sample goes
 * to the channel of the mask label found by a nearest lookup at the sample point; labels is
 * the (dx, dy, dz) uint8 label map, out is (B, C, N) and is fully written. */
int ddrr_trilinear_forward_channels(const float *volume, const unsigned char *labels, int dx,
                                    int dy, int dz, const float *source, int src_n,
                                    const float *target, const float *img, int B, int N, int C,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int align_corners, int det_h, int det_w, int tile_h,
                                    int tile_w, float *out, void *stream);

/* Volume-stationary forms of the marcher for the DRR case (one source per pose, det_h x
 * det_w target grid, mode "bilinear", reducefn "sum", align_corners = 0): bricks of 31^3 base
 * cells (+1 voxel halo, staged as 32^3 in LDS; voxels outside the volume staged as zeros =
 * the zero padding); a sample belongs to the brick holding floor(index coordinate).
 * _forward_bricks: out (B, N) zero-filled by the call and accumulated with atomics; aux: NULL,
 * or a (DDRR_TRI_AUX_PLANES, B, N) planar backward record for ddrr_trilinear_backward_rays
 * (then only the record is accumulated and out = img * step * plane 0 is formed from it).
 * _backward_volume_bricks: g_volume (dx, dy, dz) is fully written, no zero fill needed: on
 * 32^3 voxel bricks that OWN their voxels (a sample is visited by every brick owning one of
 * its 8 corners), accumulated in LDS and stored once -- no global atomics.
 * Replace renderers.py:205-241 and grid_sampler_3d_backward (bilinear) like the two
 * entries above, which remain the path for arbitrary ray lists and other modes. */
int ddrr_trilinear_forward_bricks(const float *volume, int dx, int dy, int dz,
                                  const float *source, const float *target, const float *img,
                                  int B, int det_h, int det_w, float voxel_shift, float eps,
                                  int n_points, const float *alphamin, const float *alphamax,
                                  float *out, float *aux, void *launch_ws, void *stream);

/* mask_to_channels of the marcher (Trilinear.forward mask branch, renderers.py:242-252) for the
 * DRR case on the volume-stationary bricks (mode "bilinear", align_corners = 0): the packed words
 * of ddrr_siddon_forward_channels_bricks (value with a 16-bit mantissa, label in the low byte)
 * in the marcher's halo bricks; every sample goes to the channel of the label of its nearest
 * voxel, found by the reference's own fp32 coordinate chain as in
 * ddrr_trilinear_forward_channels.  out (B, C, N) is fully written; B * C * N < 2^30, N < 2^22.
 * The backward is ddrr_trilinear_backward_channels. */
int ddrr_trilinear_forward_channels_bricks(const float *volume, const unsigned char *labels,
                                           int dx, int dy, int dz, const float *source,
                                           const float *target, const float *img, int B,
                                           int det_h, int det_w, int C, float voxel_shift,
                                           float eps, int n_points, const float *alphamin,
                                           const float *alphamax, float *out, void *launch_ws,
                                           void *stream);
/* Ray / range gradients of the march from the record of ddrr_trilinear_forward_bricks
 * (what ddrr_trilinear_backward computes by marching again); one source per pose.
 * Any output may be NULL; shapes as in ddrr_trilinear_backward. */
int ddrr_trilinear_backward_rays(const float *aux, const float *grad_out, const float *source,
                                 const float *target, const float *img, int B, int N, float eps,
                                 int n_points, const float *alphamin, const float *alphamax,
                                 float *g_source, float *g_target, float *g_img, float *g_alpha,
                                 void *stream);
int ddrr_trilinear_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                          const float *target, const float *img,
                                          const float *grad_out, int B, int det_h, int det_w,
                                          float voxel_shift, float eps, int n_points,
                                          const float *alphamin, const float *alphamax,
                                          float *g_volume, void *launch_ws, void *stream);

/* Backward of ddrr_trilinear_forward (reduce sum).  Any output may be NULL.
 * g_source/g_target: per ray (B, N, 3), through the sample positions;
 * g_img (B, N); g_alpha (B, N, 2): per-ray contributions to d/d alphamin and
 * d/d alphamax (sum them); g_volume: ACCUMULATED with fp32 atomics. */
int ddrr_trilinear_backward(const float *volume, int dx, int dy, int dz, const float *source,
                            int src_n, const float *target, const float *img,
                            const float *grad_out, int B, int N, float voxel_shift, float eps,
                            int n_points, const float *alphamin, const float *alphamax,
                            int mode_nearest, int align_corners, int det_h, int det_w, int tile_h,
                            int tile_w, float *g_source, float *g_target, float *g_img,
                            float *g_alpha, float *g_volume, void *stream);

/* Backward of ddrr_trilinear_forward_channels (autograd of renderers.py:242-252) for grad_out
 * (B, C, N); outputs as in ddrr_trilinear_backward. */
int ddrr_trilinear_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                     int dy, int dz, const float *source, int src_n,
                                     const float *target, const float *img,
                                     const float *grad_out, int B, int N, int C,
                                     float voxel_shift, float eps, int n_points,
                                     const float *alphamin, const float *alphamax,
                                     int align_corners, int det_h, int det_w, int tile_h,
                                     int tile_w, float *g_source, float *g_target, float *g_img,
                                     float *g_alpha, float *g_volume, void *stream);

/* The materialised per-sample tensor the reference hands to a CALLABLE reducefn of the marcher
 * (renderers.py:226-238): samples is (B, n_points, N) -- the reference's (B, N, P) transposed --
 * with samples[m] = img * step * T(V, x(alpha_m)).  _backward: its autograd for grad_samples
 * (B, n_points, N); outputs as in ddrr_trilinear_backward. */
int ddrr_trilinear_samples(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int align_corners,
                           float *samples, void *stream);
int ddrr_trilinear_samples_backward(const float *volume, int dx, int dy, int dz,
                                    const float *source, int src_n, const float *target,
                                    const float *img, const float *grad_samples, int B, int N,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int mode_nearest, int align_corners, float *g_source,
                                    float *g_target, float *g_img, float *g_alpha,
                                    float *g_volume, void *stream);

/* ddrr_trilinear_backward for reducefn = "max" (renderers.py:178-179): the gradient goes to the
 * arg-max sample of every ray alone.  Arguments and outputs as ddrr_trilinear_backward. */
int ddrr_trilinear_backward_max(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int n_points, const float *alphamin, const float *alphamax,
                                int mode_nearest, int align_corners, float *g_source,
                                float *g_target, float *g_img, float *g_alpha, float *g_volume,
                                void *stream);

/* Fused ray generation for the DRR case: replaces the tensor programs between a pose and
 * the renderer call -- detector.py:151-153 (pose = reorient o extrinsic applied to the
 * calibrated detector points), drr.py:201 (img = ||target - source||, world units) and
 * drr.py:204-205 (affine_inverse to voxel coordinates).  Mw (B, 3, 4): world pose of the
 * C-arm per DRR; Ainv (3, 4): world -> voxel; P (N, 3): calibrated detector points
 * (detector.py:147-150).  Writes source_v (B, 3), target_v (B, N, 3), img (B, N). */
int ddrr_raygen_forward(const float *Mw, const float *Ainv, const float *P, int B, int N,
                        float *source_v, float *target_v, float *img, void *stream);

/* Adjoint of ddrr_raygen_forward chained behind ddrr_siddon_backward_rays, reduced per
 * pose in the kernel: gMw (B, 3, 4) = dLoss/dMw (zero-filled by the call) from the forward record `aux`, grad_out
 * (B, N) and the rays the forward used.  Replaces torch autograd of renderers.py:94-113,
 * drr.py:201-205 and detector.py:151-153 (no (B, N, 3) gradient tensor is materialised).
 * with_img_path = 0 drops the gradient through `img`
 * (stop_gradients_through_grid_sample, renderers.py:63-65).  Reduce sum only. */
int ddrr_siddon_backward_pose(const float *aux, int aux_layout, const float *grad_out,
                              const float *source_v, const float *target_v, const float *img,
                              const float *Mw, const float *Ainv, const float *P, int B, int N,
                              float eps, int with_img_path, float *gMw, void *stream);

/* World pose of the C-arm per DRR from Euler angles (radians) + translation in one launch:
 * Mw (B, 3, 4) = [R | R xyz] @ reorient, R = E(a0, rot0) E(a1, rot1) E(a2, rot2), axes a_k in
 * {0: X, 1: Y, 2: Z} -- reference pose.py:444-470 (euler_angles_to_matrix), pose.py:155-157
 * + 108-114 (convert / make_matrix) and detector.py:151 (reorient.compose(pose)), ~35 ATen
 * launches there and ~70 in their backward.  reorient34: top 3 rows of the 4x4 reorient.
 * _backward: g_rot (B, 3), g_xyz (B, 3) from gMw (B, 3, 4). */
int ddrr_pose_euler_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                            const float *reorient34, int B, float *Mw, void *stream);
int ddrr_pose_euler_backward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *gMw, int B, float *g_rot,
                             float *g_xyz, void *stream);

/* The registration / sweep step AROUND the renderer in three launches instead of nine (ABI 29).
 * A registration iteration (reference registration.py:32-42 + metrics.py:21-44 per iteration of
 * notebooks/tutorials/registration.ipynb:240-316) is, at one pose, ~0.19 ms of brick kernel and
 * ~0.06 ms of small launches at ~4.4 us each.  Same arithmetic per element as the entries they fuse:
 *   ddrr_pose_raygen_forward      = ddrr_pose_euler_forward + ddrr_raygen_forward (Mw is still
 *                                   written: the backward reads it); since ABI 30 it can also clear
 *                                   what the render behind it needs zeroed -- `clear`: clear_floats
 *                                   floats (the record or image of ddrr_siddon_forward_bricks),
 *                                   clear_launch_ws: that call's launch workspace (its brick
 *                                   counter), either may be NULL; both 16-byte aligned -- which
 *                                   is then told so (ranges_valid | DDRR_BRICKS_CLEARED): one
 *                                   launch less in front of a one-pose render;
 *   ddrr_siddon_ncc_forward       = the image from the record (out = img * plane I; `out` may be
 *                                   NULL) + ddrr_ncc_forward, many workgroups per pair (moments
 *                                   by double atomics; the pair's last workgroup finishes `stats`);
 *                                   ABI 31: `ncc_sum` (one float, or NULL) receives sum_b ncc[b], put
 *                                   together by the pairs' last workgroups -- the objective of a
 *                                   batched registration step without a reduction launch behind it;
 *   ddrr_siddon_ncc_backward_pose = ddrr_ncc_backward + ddrr_siddon_backward_pose +
 *                                   ddrr_pose_euler_backward: g_rot, g_xyz (B, 3) of
 *                                   sum_b g_out[b] ncc[b], nothing per ray or per pixel is written.
 *                                   `target_v` and `img` are what ddrr_pose_raygen_forward wrote for
 *                                   (Mw, Ainv, P): the kernel regenerates a ray's target and length
 *                                   from those -- the same bits, 16 of 52 bytes per ray not read --
 *                                   and the two pointers are only checked for NULL;
 * aux: the BLOCKED float record of ddrr_siddon_forward_bricks (which may be called with
 * out = NULL when only the record is wanted).  x1: the fixed image(s), (B, N) with
 * x1_stride = N or one image with x1_stride = 0.  ws: ddrr_siddon_ncc_workspace_bytes(B) bytes,
 * 8-byte aligned, caller-owned, ZERO when first handed over and left zero by every call (the
 * accumulators and tickets clean up after themselves: no fills between calls); one workspace
 * per stream.  Reduce sum, float record only. */
long ddrr_siddon_ncc_workspace_bytes(int B);
int ddrr_pose_raygen_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *Ainv, const float *P, int B, int N,
                             float *Mw, float *source_v, float *target_v, float *img, float *clear,
                             long clear_floats, void *clear_launch_ws, void *stream);
int ddrr_siddon_ncc_forward(const float *aux, const float *img, const float *x1, long x1_stride, int B,
                            int N, float eps, void *ws, float *ncc, float *stats, float *out,
                            float *ncc_sum, void *stream);
int ddrr_siddon_ncc_backward_pose(const float *aux, const float *img, const float *x1, long x1_stride,
                                  const float *stats, const float *g_out, int g_stride,
                                  const float *source_v, const float *target_v, const float *Mw,
                                  const float *Ainv, const float *P, const float *rot, const float *xyz,
                                  int a0, int a1, int a2, const float *reorient34, int B, int N,
                                  float eps, int with_img_path, void *ws, float *g_rot, float *g_xyz,
                                  void *stream);

/* ddrr_siddon_backward_pose + ddrr_pose_euler_backward in one launch (ABI 33): (g_rot, g_xyz) (B, 3) of ANY
 * objective of the image, given its per-pixel gradient grad_out (B, N) -- what autograd hands the render of
 * `drr(rot, xyz, parameterization="euler_angles")` (reference drr.py:155-188, registration.py:32-33) when
 * the similarity is computed outside it (MultiscaleNormalizedCrossCorrelation2d, GradientNormalized-
 * CrossCorrelation2d, a user's own loss).  The same kernel as ddrr_siddon_ncc_backward_pose with the
 * gradient read instead of formed from NCC statistics; aux, source_v, Mw and ws as there (rays from
 * ddrr_pose_raygen_forward for (rot, xyz, reorient34, Ainv, P); ws: ddrr_siddon_ncc_workspace_bytes(B)). */
int ddrr_siddon_backward_pose_euler(const float *aux, const float *grad_out, const float *source_v,
                                    const float *Mw, const float *Ainv, const float *P, const float *rot,
                                    const float *xyz, int a0, int a1, int a2, const float *reorient34, int B,
                                    int N, float eps, int with_img_path, void *ws, float *g_rot, float *g_xyz,
                                    void *stream);

/* One Adam step of a registration's two pose parameter groups in ONE launch (ABI 30): rot, xyz
 * (B, 3) updated in place from their gradients, with torch.optim.Adam's update rule (no weight
 * decay, no amsgrad; reference notebooks/tutorials/registration.ipynb:240-316 steps
 * torch.optim.Adam([{rotation, lr_rot}, {translation, lr_xyz}], maximize=True)):
 *   step += 1;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)      (maximize: g = -g)
 * m_*, v_* (B, 3) and step_* (1 float each) are the optimizer's state, zero before the first step.
 * torch's fused Adam takes four launches for the two groups (diffdrr_amd.registration.PoseAdam is
 * the torch.optim.Optimizer over this entry). */
int ddrr_pose_adam_step(float *rot, float *xyz, const float *g_rot, const float *g_xyz, float *m_rot,
                        float *v_rot, float *m_xyz, float *v_xyz, float *step_rot, float *step_xyz, int B,
                        float lr_rot, float lr_xyz, float beta1, float beta2, float eps, int maximize,
                        void *stream);

/* NormalizedCrossCorrelation2d, patch_size = None (reference metrics.py:21-44) for image
 * pairs of N pixels: out (B) = mean(z1 z2), z = (x - mean) / sqrt(var + eps).  x2 (B, N);
 * x1 (B, N) with x1_stride = N, or ONE image shared by the batch with x1_stride = 0 (the
 * fixed image of a registration / sweep).  stats (B, 5) = {mu1, s1, mu2, s2, ncc} for the
 * backward.  One kernel instead of ~10 reductions / elementwise passes. */
int ddrr_ncc_forward(const float *x1, long x1_stride, const float *x2, int B, int N, float eps,
                     float *out, float *stats, void *stream);

/* Gradient of ddrr_ncc_forward w.r.t. x2 (and x1 unless shared); either may be NULL.  g_out: (B)
 * with g_stride = 1, or ONE value for every pair with g_stride = 0 (what autograd hands the
 * backward of `.sum()` / `.mean()`: an expanded scalar, read in place). */
int ddrr_ncc_backward(const float *x1, long x1_stride, const float *x2, const float *stats,
                      const float *g_out, int g_stride, int B, int N, float *g_x1, float *g_x2,
                      void *stream);

/* NormalizedCrossCorrelation2d(patch_size = p) (reference metrics.py:16-44; ABI 31): the mean over all
 * p x p windows (stride 1) of the windows' own NCC -- each window z-scored on its own, biased variance
 * + eps -- for image pairs (H, W): the local similarity of MultiscaleNormalizedCrossCorrelation2d
 * (metrics.py:47-63) and of GradientNormalizedCrossCorrelation2d(patch_size = p) over the 2 B Sobel
 * channel images.  The reference materialises both images as (B, (H-p+1)(W-p+1), p, p) tensors (`to_patches`,
 * 256^2 at p = 13: 40 MB per image and pose) and autograd a dozen more; here a window is two passes over its
 * pixels in LDS.  x2 (B, H, W); x1 (B, H, W) with x1_stride = H W, or ONE image with x1_stride = 0.
 * out (B) (zeroed by the call).  coef (B, H-p+1, W-p+1, 4), 16-byte aligned, or NULL: per window
 * {1 / (s1 s2), mu1 / (s1 s2), ncc / s2^2, mu2 ncc / s2^2} for the backward, which forms
 * d out / d x2 (B, H, W) per pixel from the <= p^2 windows that hold it (g_out, g_stride: as
 * ddrr_ncc_backward).  1 <= p <= min(H, W, 64). */
int ddrr_ncc_patch_forward(const float *x1, long x1_stride, const float *x2, int B, int H, int W, int p,
                           float eps, float *out, float *coef, void *stream);
int ddrr_ncc_patch_backward(const float *x1, long x1_stride, const float *x2, const float *coef,
                            const float *g_out, int g_stride, int B, int H, int W, int p, float *g_x2,
                            void *stream);

/* The Sobel pair in front of GradientNormalizedCrossCorrelation2d (reference metrics.py:69-94:
 * Conv2d(1, 2, 3, padding=1) with Gx = [[1,0,-1],[2,0,-2],[1,0,-1]], Gy = [[1,2,1],[0,0,0],
 * [-1,-2,-1]], zero padding): img (B, H, W) -> out (B, 2, H, W), and its adjoint g_out (B, 2, H, W)
 * -> g_img (B, H, W).  The similarity itself is ddrr_ncc_forward over the 2 B channel images. */
int ddrr_sobel_forward(const float *img, int B, int H, int W, float *out, void *stream);
int ddrr_sobel_backward(const float *g_out, int B, int H, int W, float *g_img, void *stream);

/* The same with the reference's Gaussian blur in front (metrics.py:66, 88-92: Sobel(sigma > 0) calls
 * torchvision's gaussian_blur(img, k = int(6 sigma + 1) | 1, sigma) -- REFLECT padding by k // 2, then the outer
 * product of k normalised taps; the default of GradientNormalizedCrossCorrelation2d is sigma = 1, k = 7) in one
 * launch each way: img (B, H, W) with img_stride = H W, or ONE image with img_stride = 0 -> out (B, 2, H, W),
 * and the adjoint g_out (B, 2, H, W) -> g_img (B, H, W).  taps: k floats ON THE DEVICE (the host computes them
 * as torchvision does: exp(-x^2 / 2 sigma^2) on linspace(-(k-1)/2, (k-1)/2, k), normalised), k odd, <= 31,
 * k // 2 < min(H, W).  The blurred image is zero outside the image for the Sobel's zero padding. */
int ddrr_blur_sobel_forward(const float *img, long img_stride, int B, int H, int W, const float *taps, int k,
                            float *out, void *stream);
int ddrr_blur_sobel_backward(const float *g_out, int B, int H, int W, const float *taps, int k, float *g_img,
                             void *stream);

/* ---- double precision ---------------------------------------------------------------------
 * The reference computes in the dtype its module holds: `DRR(...).to(torch.float64)` renders and
 * differentiates in fp64 (drr.py:71-75; renderers.py:34-76, 205-241 follow their inputs' dtype).
 * Same semantics as the fp32 entry points above, all pointers `double`; per-ray kernels (any
 * ray list), for accuracy rather than speed:
 *   Siddon     mode="nearest", align_corners=False, reducefn sum | max (forward);
 *              aux: NULL or a (B, N, 8) record {I, S0_xyz, S1_xyz, -} of the sum for _backward
 *   _backward  g_source (B, N, 3) per ray, g_target (B, N, 3), g_img (B, N) from the record;
 *              g_volume (Dx, Dy, Dz), accumulated with atomics into a zero-filled array, by a
 *              second walk; any of the four may be NULL
 *   Trilinear  mode="bilinear", align_corners=False, reducefn sum; alphamin / alphamax: device
 *              scalars (renderers.py:220-223); _backward additionally g_alpha (B, N, 2) per ray
 *              (d / d alphamin, d / d alphamax: sum over rays on the caller's side). */
int ddrr_siddon_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                            int src_n, const double *target, const double *img, int B, int N,
                            double voxel_shift, double eps, int reduce_mode, double *out,
                            double *aux, void *stream);
int ddrr_siddon_backward_f64(int dx, int dy, int dz, const double *source, int src_n,
                             const double *target, const double *img, const double *grad_out,
                             const double *aux, int B, int N, double voxel_shift, double eps,
                             double *g_source, double *g_target, double *g_img, double *g_volume,
                             void *stream);
int ddrr_trilinear_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                               int src_n, const double *target, const double *img, int B, int N,
                               double voxel_shift, double eps, int n_points,
                               const double *alphamin, const double *alphamax, double *out,
                               void *stream);
int ddrr_trilinear_backward_f64(const double *volume, int dx, int dy, int dz,
                                const double *source, int src_n, const double *target,
                                const double *img, const double *grad_out, int B, int N,
                                double voxel_shift, double eps, int n_points,
                                const double *alphamin, const double *alphamax, double *g_source,
                                double *g_target, double *g_img, double *g_alpha, double *g_volume,
                                void *stream);

/* ---- the materialising general path -------------------------------------------------------
 * Every keyword combination of the reference's renderers that the fused entry points above do
 * not take goes through the tensors the reference itself materialises just before it reduces
 * (renderers.py:70-71 the (B, N, M - 1) per-segment terms `img * intersection_length`,
 * M = dx + dy + dz + 3; :235-236 the (B, N, P) per-sample terms `img * step_size`); sum / max / a
 * callable reducefn / the mask_to_channels scatter (:77-89, :242-252) are then ordinary tensor
 * operations on the caller's side.  That covers: Siddon with a mask, a callable reducefn, or
 * gradients of reducefn "max" / stop_gradients_through_grid_sample TOGETHER WITH a midpoint
 * lookup (mode="bilinear" / align_corners=True; :57-66), and all of those plus the marcher's
 * mode="nearest" / reducefn="max" / align_corners=True for a module in float64 (drr.py:71-75).
 *   f64            0: every pointer is `float`, 1: `double` (volume, rays, img, outputs alike)
 *   terms          (B, M - 1, N)  -- the reference's tensor transposed, rays write coalesced;
 *   samples        (B, P, N)
 *   raw            1: the looked-up values themselves, without img and the interval / step
 *                  length: the label lookup `_get_voxel(mask, ...)` (:82-84, :246-248; the
 *                  caller truncates `.long()`)
 *   lookup         DDRR_LOOKUP_* (STEP only with align_corners = 0); nearest: marcher mode
 *   _backward      autograd of the (not raw) tensor for grad_terms / grad_samples of its shape:
 *                  g_source (B, N, 3) per ray, g_target (B, N, 3), g_img (B, N), g_alpha
 *                  (B, N, 2) per ray, g_volume (Dx, Dy, Dz) accumulated with atomics into a
 *                  zero-filled array; any may be NULL.  through_lookup = 0: the values were
 *                  looked up under no_grad (stop_gradients_through_grid_sample, :63-65): only
 *                  the interval lengths carry gradient (g_img, g_volume must be NULL). */
int ddrr_siddon_segments_general(const void *volume, int f64, int dx, int dy, int dz,
                                 const void *source, int src_n, const void *target,
                                 const void *img, int B, int N, double voxel_shift, double eps,
                                 int lookup, int align_corners, int raw, void *terms,
                                 void *stream);
int ddrr_siddon_segments_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                          const void *source, int src_n, const void *target,
                                          const void *img, const void *grad_terms, int B, int N,
                                          double voxel_shift, double eps, int lookup,
                                          int align_corners, int through_lookup, void *g_source,
                                          void *g_target, void *g_img, void *g_volume,
                                          void *stream);
int ddrr_trilinear_samples_general(const void *volume, int f64, int dx, int dy, int dz,
                                   const void *source, int src_n, const void *target,
                                   const void *img, int B, int N, double voxel_shift, double eps,
                                   int n_points, const void *alphamin, const void *alphamax,
                                   int nearest, int align_corners, int raw, void *samples,
                                   void *stream);
int ddrr_trilinear_samples_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                            const void *source, int src_n, const void *target,
                                            const void *img, const void *grad_samples, int B,
                                            int N, double voxel_shift, double eps, int n_points,
                                            const void *alphamin, const void *alphamax,
                                            int nearest, int align_corners, void *g_source,
                                            void *g_target, void *g_img, void *g_alpha,
                                            void *g_volume, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFDRR_HIP_H */
