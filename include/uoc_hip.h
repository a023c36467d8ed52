/*
 * uoc_hip.h — C ABI of libuoc_hip.so, the MI355X (gfx950) native library behind the
 * reference's Python call surface for the inference hot path
 * (the reference has no FFI layer — these entry points are what a ctypes
 * binding for that path binds).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer into caller-owned memory (a torch tensor);
 *     the library allocates nothing across the boundary except the opaque net handle.
 *   - every function returns 0 on success or a negative errno-style code; it never throws.
 *     uoc_last_error() returns a thread-local message for the last failure.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises unless documented.
 *   - embeddings ("X") are PIXEL-MAJOR: [batch][n][64] fp32, one 256-byte row per pixel.
 *     This is the layout the fused backbone head writes and the clustering kernels read.
 */
#ifndef UOC_HIP_H
#define UOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UOC_OK 0
#define UOC_EINVAL (-22)
#define UOC_ENOMEM (-12)
#define UOC_EHIP (-5)
#define UOC_ENOENT (-2)
#define UOC_ETIMEDOUT (-110)

#define UOC_EMBED_DIM 64   /* channel count C the kernels are specialised for */
#define UOC_MAX_SEEDS 128  /* num_seeds upper bound (reference default 100)   */

int uoc_version(void);
/* Always 0 since round 6: every kernel choice that affects rounding is a compile-time constant and the library holds one
 * implementation per step (rounds 3-5 had a -DUOC_DEV build with the measured-and-rejected alternates).  Kept for ABI stability. */
int uoc_is_dev_build(void);
/* Hash of everything that could make two processes compute different bits (the library version).  The frame-parallel runner all-reduces it next to its error flag: ranks that disagree
 * fail before the gather instead of silently breaking sharding independence. */
unsigned long long uoc_config_fingerprint(void);
/* Releases process-wide helper objects (the per-device events that order the persistent sampling kernels of
 * different streams).  Optional; call it before process exit while the HIP runtime is still up.  Idempotent. */
int uoc_shutdown(void);
const char *uoc_last_error(void);
/* The library reads its (speed-only) UOC_* environment variables once and caches them; a process that changes one
 * afterwards (tests, micro-benchmarks) calls this to have them read again at their next use. */
int uoc_reload_env(void);

/* ------------------------------------------------------------------------------------------
 * Mean-shift clustering  — replaces lib/utils/mean_shift.py:128-229 (cosine metric)
 * ---------------------------------------------------------------------------------------- */

/* Seed selection runs as ONE persistent cooperative launch with X resident on chip when the batch
 * fits the device (default); 0 forces the streaming one-launch-per-step kernel.  Same results. */
int uoc_ms_set_persistent_fps(int on);
/* A caller that launches clustering from SEVERAL streams of one device (frames in flight) must switch this on: the
 * persistent sampling grids of different streams are then ordered through a per-device event, because two of them
 * partially resident at the same time would each wait for blocks the other keeps off the chip.  Off by default
 * (single-stream callers pay nothing). */
int uoc_ms_set_stream_ordering(int on);
/* Number of seed-selection calls since process start in which fields that the on-chip kernel should have handled
 * went to the streaming kernel instead (does not fit on chip / cooperative launch refused).  The two kernels sum the
 * dot product in different orders; a caller that needs placement-independent results checks that this stays 0. */
int uoc_ms_fps_fallbacks(void);
/* The persistent kernel's grid-wide exchange spins with a bound; a block that gives up raises a sticky device
 * flag and the call's outputs are then meaningless.  uoc_ms_check synchronises `stream`, reads and clears the
 * flag: 0, or UOC_ETIMEDOUT.  The host mirrors call it at the points where they synchronise anyway. */
int uoc_ms_check(void *stream);

/* Scratch bytes the clustering entry points need for (batch, n, m). */
size_t uoc_ms_workspace_bytes(int batch, int n, int m);

/* Farthest-point seed selection — select_smart_seeds, mean_shift.py:128-189.
 * d_first_index [batch] int32: the index the reference draws with np.random.randint (:155).
 * d_seeds [batch][m][64], d_indices [batch][m] int32. */
int uoc_ms_select_seeds(const float *d_X, int batch, int n, int m, const int32_t *d_first_index,
                        float *d_seeds, int32_t *d_indices, void *d_ws, size_t ws_bytes, void *stream);

/* The same with the first num_init rows of every d_seeds[b] already chosen by the caller — the init_seeds /
 * num_init_seeds continuation of select_smart_seeds, mean_shift.py:142-170 (`seeds = init_seeds` there too: the
 * selection is written into the caller's matrix).  The given rows need not be rows of X; their d_indices stay -1
 * like the reference's selected_indices.  num_init = 0 is uoc_ms_select_seeds (d_first_index is then required;
 * otherwise it is not read).  num_init > 0 always runs the one-launch-per-step kernel. */
int uoc_ms_select_seeds_from(const float *d_X, int batch, int n, int m, int num_init, const int32_t *d_first_index,
                             float *d_seeds, int32_t *d_indices, void *d_ws, size_t ws_bytes, void *stream);

/* iters x { W = exp(kappa Z X^T); Z = normalize(W X) } — seed_hill_climbing_ball,
 * mean_shift.py:79-109 with ball_kernel :26.  d_Z [batch][m][64] is updated in place. */
int uoc_ms_hill_climb(const float *d_X, int batch, int n, float *d_Z, int m, float kappa, int iters,
                      void *d_ws, size_t ws_bytes, void *stream);

/* Sequential epsilon-ball labelling of the seeds — connected_components, mean_shift.py:41-76.
 * d_seed_labels [batch][m] int32; d_num_unique [batch] int32 = len(unique(seed labels)). */
int uoc_ms_seed_components(const float *d_Z, int batch, int m, float epsilon, int32_t *d_seed_labels,
                           int32_t *d_num_unique, void *stream);

/* Nearest-seed assignment + "largest cluster becomes label 0" — mean_shift.py:211-227.
 * d_labels [batch][n] int32; d_closest (nullable) [batch][n] int32 = argmin seed. */
int uoc_ms_assign(const float *d_X, int batch, int n, const float *d_Z, const int32_t *d_seed_labels,
                  const int32_t *d_num_unique, int m, int32_t *d_labels, int32_t *d_closest,
                  void *d_ws, size_t ws_bytes, void *stream);

/* The four stages above back to back — mean_shift_smart_init, mean_shift.py:192-229.
 * d_Z_out (nullable) receives the converged seeds, d_seed_labels_out (nullable) their labels. */
int uoc_ms_cluster(const float *d_X, int batch, int n, int m, float kappa, int iters, float epsilon,
                   const int32_t *d_first_index, int32_t *d_labels, int32_t *d_indices,
                   float *d_Z_out, int32_t *d_seed_labels_out, void *d_ws, size_t ws_bytes, void *stream);

/* 128-d embeddings (cfg.TRAIN.FUSION_TYPE = 'cat', SEG.py:109-110): the field is stored as `halves` = 2 planes of 64
 * channels, d_X [batch][halves][n][64] (plane 0 = channels 0..63), and so are the converged seeds d_Z_out
 * [batch][halves][m][64]; every dot product runs over both planes.  halves = 1 is exactly uoc_ms_cluster. */
size_t uoc_ms_workspace_bytes_wide(int batch, int n, int m, int halves);
int uoc_ms_cluster_wide(const float *d_X, int halves, int batch, int n, int m, float kappa, int iters, float epsilon,
                        const int32_t *d_first_index, int32_t *d_labels, int32_t *d_indices, float *d_Z_out,
                        int32_t *d_seed_labels_out, void *d_ws, size_t ws_bytes, void *stream);


/* ------------------------------------------------------------------------------------------
 * RGB-D ResNet34-8s embedding network — replaces SEGNET.forward (lib/networks/SEG.py:88-119,
 * RGBD/'add' branch) with its two Resnet34_8s backbones (resnet_dilated.py:287-327,
 * resnet.py:116-270).  Activations NHWC fp32, convolutions on fp32 MFMA, BatchNorm folded.
 * ---------------------------------------------------------------------------------------- */
typedef struct uoc_net uoc_net;

/* Input modality / fusion of SEGNET (SEG.py:69-71 construction, :97-110 forward):
 *   RGBD_ADD   cfg.INPUT='RGBD', FUSION_TYPE='add'   : fcn(img) + fcn_depth(xyz)          (two backbones)
 *   COLOR      cfg.INPUT='COLOR'                     : fcn(img)
 *   DEPTH      cfg.INPUT='DEPTH'                     : fcn(xyz)   (the weights still live under "fcn.")
 *   RGBD_EARLY cfg.INPUT='RGBD', FUSION_TYPE='early' : fcn(cat(img, xyz)), a 6-channel stem (SEG.py:178-181)
 *   RGBD_CAT   cfg.INPUT='RGBD', FUSION_TYPE='cat'   : cat(fcn(img), fcn_depth(xyz)) -> 128-d embeddings (:109-110) */
enum { UOC_NET_RGBD_ADD = 0, UOC_NET_COLOR = 1, UOC_NET_DEPTH = 2, UOC_NET_RGBD_EARLY = 3, UOC_NET_RGBD_CAT = 4 };

int uoc_net_create(uoc_net **out);                   /* = uoc_net_create_mode(out, UOC_NET_RGBD_ADD) */
int uoc_net_create_mode(uoc_net **out, int mode);
int uoc_net_embed_dim(const uoc_net *net);           /* 64, or 128 for RGBD_CAT */
int uoc_net_destroy(uoc_net *net);
/* One state-dict entry by its reference key ("fcn.resnet34_8s.layer1.0.conv1.weight", ...;
 * SEG.py:130-159 contract), HOST fp32 memory, copied. */
int uoc_net_load_param(uoc_net *net, const char *name, const float *host_data, size_t numel);
/* Checks that every parameter is present, folds BN (eps 1e-5), re-lays weights out as
 * [tap][cout][cin] and uploads them to the CURRENT device. */
int uoc_net_finalize(uoc_net *net);
size_t uoc_net_workspace_bytes(const uoc_net *net, int B, int H, int W);
/* d_rgb, d_xyz: [B][3][H][W] fp32 NCHW (what test_sample hands the network, test_dataset.py:247);
 * the one the mode does not read (d_xyz for COLOR, d_rgb for DEPTH) may be NULL.
 * d_embed: [B][H*W][64] pixel-major unit-norm embeddings; RGBD_CAT: [B][2][H*W][64], plane 0 = the image
 * branch's 64 channels, plane 1 = the XYZ branch's, normalised over all 128 (the layout uoc_ms_cluster_wide reads). */
/* EXPERIMENT (round 6), off by default and never part of the headline measurement: the plane GEMMs of the Winograd layers in
 * split precision — every fp32 operand as three bf16 terms (3 x 8 = 24 significand bits), six bf16 MFMA products with fp32
 * accumulation, 2.67x the fp32 matrix rate (csrc/wino4_split.hip).  Embeddings stay within the fp32 path's distance of an
 * fp64 evaluation, but they are NOT bit-identical to it.  Call after uoc_net_finalize; on = 1 splits the transformed
 * weights once (hipMalloc + a synchronisation). */
int uoc_net_set_split_precision(uoc_net *net, int on);
int uoc_net_forward(uoc_net *net, const float *d_rgb, const float *d_xyz, int B, int H, int W, float *d_embed,
                    void *d_ws, size_t ws_bytes, void *stream);

/* Single fused conv (+bias +residual +ReLU), NHWC, over G independent groups stacked on the leading
 * dimension (in [G][B][H][W][Cin], weights [G][K*K][Cout][Cin], bias [G][Cout], ...); K in {1,3}.
 * Cin % 32 == 0, Cout % 64 == 0.  Exposed for unit tests / micro-benchmarks of the conv kernels.
 * uoc_conv2d_nhwc runs the direct implicit-GEMM kernel.  uoc_conv2d_nhwc_algo names the algorithm explicitly (no
 * environment variable decides it): UOC_CONV_DIRECT, or UOC_CONV_WINOGRAD4 = F(4x4,3x3) as the network runs its 3x3
 * stride-1 layers from 64 channels up (UOC_EINVAL if the shape is not eligible).  The Winograd path keeps its transformed weights and
 * scratch in buffers owned by this entry: one caller thread at a time. */
#define UOC_CONV_DIRECT 0
#define UOC_CONV_WINOGRAD4 4
#define UOC_CONV_WINOGRAD4_BF16X3 5   /* EXPERIMENT: the same with the split-precision plane GEMM (see uoc_net_set_split_precision) */
int uoc_conv2d_nhwc(const float *d_in, const float *d_w, const float *d_bias, const float *d_res, float *d_out,
                    int G, int B, int H, int W, int Cin, int Cout, int K, int stride, int dil, int pad, int relu,
                    void *stream);
int uoc_conv2d_nhwc_algo(const float *d_in, const float *d_w, const float *d_bias, const float *d_res, float *d_out,
                         int G, int B, int H, int W, int Cin, int Cout, int K, int stride, int dil, int pad, int relu,
                         int algo, void *stream);


/* ------------------------------------------------------------------------------------------
 * Two-stage glue — replaces lib/fcn/test_dataset.py:62-198 (filter_labels_depth, crop_rois,
 * match_label_crop) with batched device kernels.  Label maps are int32 with ids < 128.
 * ---------------------------------------------------------------------------------------- */
typedef struct uoc_roi_table {
  int32_t K;            /* number of ROIs = non-zero labels present, ascending label order    */
  int32_t label[128];   /* stage-1 label id of ROI k                                          */
  int32_t box[128][4];  /* x0, y0, x1, y1 inclusive, padded 25 % and clamped (:83-93)         */
} uoc_roi_table;

size_t uoc_roi_workspace_bytes(void);

/* Input preparation (read_sample / compute_xyz, tools/test_images.py:96-133) on the device:
 * d_bgr [H][W][3] uint8 (cv2.imread order), d_depth_mm [H][W] uint16 millimetres ->
 * d_image [3][H][W] = bgr/255 - mean (mean_* = PIXEL_MEANS/255 as float32), d_xyz [3][H][W] metres. */
int uoc_prep_rgbd(const uint8_t *d_bgr, const uint16_t *d_depth_mm, int H, int W, float fx, float fy, float px,
                  float py, float mean_b, float mean_g, float mean_r, float *d_image, float *d_xyz, void *stream);

/* filter_labels_depth (:183-198): per batch item, a non-zero label whose fraction of pixels with
 * z > 0 is < threshold becomes 0.  d_z: the Z plane of item 0; items are z_batch_stride floats apart. */
int uoc_filter_labels_depth(int32_t *d_labels, const float *d_z, long z_batch_stride, int B, int H, int W,
                            float threshold, void *d_ws, size_t ws_bytes, void *stream);

/* One pass that (optionally, d_z != NULL) applies the depth filter in place and builds the ROI
 * table of crop_rois (:68-93; mask.py:180-187 tight box, round-half-even padding, clamping). */
int uoc_roi_build(int32_t *d_labels, const float *d_z, int H, int W, float threshold, float pad_fraction,
                  uoc_roi_table *d_table, void *d_ws, size_t ws_bytes, void *stream);

/* crop_rois (:95-110): crops of the [3][H][W] image / XYZ planes resized to SxS with bilinear
 * align_corners=True, object mask with nearest.  Outputs NCHW [K][3][S][S] and [K][S][S].
 * COLOR input (depth is None, :73-76): d_xyz = d_xyz_crops = NULL. */
int uoc_roi_crop(const float *d_rgb, const float *d_xyz, const int32_t *d_labels, int H, int W,
                 const uoc_roi_table *d_table, int K, int S, float *d_rgb_crops, float *d_xyz_crops,
                 float *d_mask_crops, void *stream);

/* match_label_crop part 1 (:118-136): d_keep[k][c] = 1 iff crop cluster c of ROI k overlaps the
 * stage-1 mask by >= 50 %; d_meanz[k] = mean z (>0) over kept pixels (all pixels if none kept).
 * Without depth (d_xyz_crops = NULL) d_meanz is not written: ROIs are ordered by box area (:138-146). */
int uoc_roi_match_stats(const int32_t *d_labels_crop, const float *d_mask_crops, const float *d_xyz_crops, int K,
                        int S, int32_t *d_keep, float *d_meanz, void *d_ws, size_t ws_bytes, void *stream);

/* match_label_crop part 2 (:156-177): d_map[k][c] = global id of kept cluster c (0 = dropped),
 * d_order = ROI paint order (far to near); nearest-resize each crop back to its box and paste
 * non-zeros, later ROIs overwrite earlier ones.  d_refined [H*W] is fully written. */
int uoc_roi_paste(const int32_t *d_labels_crop, const uoc_roi_table *d_table, const int32_t *d_map,
                  const int32_t *d_order, int K, int S, int H, int W, int32_t *d_refined, void *stream);

/* match_label_crop (:116-179) entirely on the device (round 6): the statistics of uoc_roi_match_stats, then the ROI paint
 * order — sorted(key = mean depth | box area, reverse=True) with Python's stable-sort semantics, NaN keys included for
 * K < 64 (csrc/roi.hip restates CPython's list.sort for short lists) — the running renumbering of kept clusters, and the
 * paste of uoc_roi_paste.  No host round trip.  d_keep (nullable) [K][128] receives the keep table, d_plan (nullable)
 * [K + K*128] the paint order followed by the id map.  d_status (nullable): bit 0 is OR-ed in when some key is NaN and
 * K >= 64 — the one case not restated here; d_refined is then painted in index order and the caller re-does the ordering
 * on the host (uoc_roi_match_stats + uoc_roi_paste). */
int uoc_roi_match(const int32_t *d_labels_crop, const float *d_mask_crops, const float *d_xyz_crops,
                  const uoc_roi_table *d_table, int K, int S, int H, int W, int32_t *d_refined, int32_t *d_keep,
                  int32_t *d_plan, int32_t *d_status, void *d_ws, size_t ws_bytes, void *stream);

/* Frame-parallel runner (no reference counterpart; SURVEY 8(e): the gathered block is uint8 like the ROS consumer's cast,
 * ros/test_images_segmentation.py:165): d_out[i] = (uint8) d_labels[i] for i < n, *d_top = max(*d_top, max_i d_labels[i])
 * (the caller checks once per block that no id exceeded 255). */
int uoc_labels_to_u8(const int32_t *d_labels, long n, uint8_t *d_out, int32_t *d_top, void *stream);


/* ------------------------------------------------------------------------------------------
 * Evaluation — the integer statistics of multilabel_metrics (lib/utils/evaluation.py:109-257; overlap and
 * boundary precision / recall of a predicted against a ground-truth label map).  The host mirror
 * unseenobjectclustering_amd/utils/evaluation.py turns them into the reference's metric dictionary.
 * ---------------------------------------------------------------------------------------- */
typedef struct uoc_eval_tables {
  int32_t cont[128 * 128];     /* [gt][pred] pixels with that label pair (:188-190)                          */
  int32_t prec_tp[128 * 128];  /* [gt][pred] boundary pixels of pred inside the dilated boundary of gt (:103)  */
  int32_t rec_tp[128 * 128];   /* [gt][pred] boundary pixels of gt inside the dilated boundary of pred (:104)  */
  int32_t bnd_pred[128];       /* boundary pixels (seg2bmap, :15-73) of every predicted mask (:212-215)        */
  int32_t bnd_gt[128];         /* ... of every ground-truth mask (:216-219)                                    */
  int32_t bad_label;           /* != 0: a label id outside [0, 128) was seen (such pixels are skipped)         */
} uoc_eval_tables;

size_t uoc_eval_workspace_bytes(int H, int W);
/* d_pred, d_gt: [H][W] int32 label maps (0 = background); radius = bound_pix of boundary_overlap (:88-89),
 * the dilation structuring element is the disk x^2 + y^2 <= radius^2 (:94-98). */
int uoc_eval_pair_stats(const int32_t *d_pred, const int32_t *d_gt, int H, int W, int radius, uoc_eval_tables *d_tables,
                        void *d_ws, size_t ws_bytes, void *stream);


/* ------------------------------------------------------------------------------------------
 * Host-side data formats (no device work) — what the dataset loaders need in place of python-pcl
 * (lib/datasets/ocid_object.py:105, osd_object.py:92): LZF decoder for `DATA binary_compressed` PCD files.
 * `in`/`out` are HOST pointers.  Returns the number of bytes written or a negative code.
 * ---------------------------------------------------------------------------------------- */
long uoc_lzf_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap);

/* ------------------------------------------------------------------------------------------
 * Opt-in per-kernel timing (HIP events on the launch stream).  Single-threaded use.
 * uoc_prof_report writes a JSON array [{kernel, launches, total_ms, flops, bytes}, ...] where
 * flops/bytes are the ALGORITHMIC totals of the recorded launches.
 * ---------------------------------------------------------------------------------------- */
int uoc_prof_enable(int on);
int uoc_prof_reset(void);
int uoc_prof_report(char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* UOC_HIP_H */
