/*
 * synergy_hip.h -- C ABI of the MI355X (gfx950) implementation of SynergyNet's
 * inference hot path:
 *
 *   crops [B,3,120,120] -> MobileNetV2 -> 62-d 3DMM params -> 68 landmarks,
 *   53215-vertex mesh, head pose.
 *
 * The reference has no FFI layer for this path; its boundary is the Python
 * class synergy3DMM.SynergyNet (reference synergy3DMM.py:70-207).  Each entry
 * point below names the reference method(s) it replaces.  The Python class
 * synergynet_amd.synergy3DMM.SynergyNet binds these symbols with ctypes and
 * keeps the reference's method set (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative
 *     syn_status, and syn_last_error() returns a thread-local message;
 *   - all tensor arguments of compute calls are DEVICE pointers owned by the
 *     caller (e.g. torch-ROCm tensor.data_ptr()), fp32, contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     are asynchronous on that stream;
 *   - the library owns only its packed constants and its activation workspace;
 *   - one handle per device; calls on one handle are not thread-safe, different
 *     handles are independent.
 */
#ifndef SYNERGY_HIP_H
#define SYNERGY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct syn_handle syn_handle;

typedef enum {
    SYN_OK = 0,
    SYN_ERR_INVALID = -1,      /* bad argument (NULL pointer, B <= 0, wrong size ...) */
    SYN_ERR_HIP = -2,          /* a HIP runtime call failed; see syn_last_error()     */
    SYN_ERR_NOT_LOADED = -3,   /* compute call before the constants it needs          */
    SYN_ERR_PARAM_LEN = -4     /* 'length of params mismatch' (synergy3DMM.py:126-129) */
} syn_status;

#define SYN_PARAM_DIM 62        /* 12 pose + 40 shape + 10 expression (synergy3DMM.py:30-37) */
#define SYN_POOL_DIM 1280       /* pooled feature (mobilenetv2_backbone.py:179-182)          */
#define SYN_STD_SIZE 120        /* utils/params.py:34                                        */

const char *syn_last_error(void);
int syn_abi_version(void);

/* SynergyNet.__init__ (synergy3DMM.py:70-114): one handle per GPU. */
int syn_create(int device, syn_handle **out);
int syn_destroy(syn_handle *h);

/* ---- constants ------------------------------------------------------------------- */

/* Number of floats syn_load_backbone expects. */
size_t syn_backbone_flat_count(void);

/* load_weights (synergy3DMM.py:156-164) for the I2P.backbone.* part of the state_dict.
 * `flat` is a HOST array: for every conv layer in network order
 *   conv.weight (PyTorch [Cout,Cin/groups,kh,kw] order), bn.weight, bn.bias,
 *   bn.running_mean, bn.running_var
 * then classifier_ori.1.{weight,bias}, classifier_shape.1.{...}, classifier_exp.1.{...}
 * (mobilenetv2_backbone.py:33-74,107-158).  BatchNorm (eval, eps 1e-5) is turned into a
 * per-channel scale/shift and the weights are repacked for the kernels. */
int syn_load_backbone(syn_handle *h, const float *flat, size_t n_floats);

/* Range safety of the fp16 arithmetic.  Every GEMM of the default schedule carries its fp32 operands as two fp16 pieces (22-bit
 * significand, fp32 accumulation); fp16's exponent range is 2^-14 .. 65504, and the reference loads ARBITRARY checkpoints
 * (synergy3DMM.py:156-164).  syn_load_backbone therefore bounds, from the folded constants alone, every tensor that is split at run
 * time (interval arithmetic: hidden activations are in [0,6] after ReLU6, project outputs and residual sums follow from the weights)
 * and checks every packed weight row; a block whose bound does not fit runs the tiled fp16 kernel at input scale 1 or the exact
 * fp32-MFMA kernel instead (csrc/synergy_abi.hip analyze_mbv2_ranges).  The verdict travels with exported constants.
 * syn_numerics_report: one text line per .features[] index into buf (NUL terminated, truncated to n); returns the number of
 * blocks that do NOT run the default kernel (0 = the whole network is inside the fp16 window), or a negative syn_status. */
int syn_numerics_report(syn_handle *h, char *buf, size_t n);

/* Which arithmetic the compute calls of this handle use from now on: 2 (default) fused kernels with fp16x2 operands, 1 fused kernels on
 * the fp32-input MFMA (exact fp32 products: the cross-check and fallback schedule), 0 one fp32 kernel per layer.  Returns the previous
 * value (>= 0) or a negative syn_status.  Lets a caller compare the default schedule with the exact one ON ITS OWN CHECKPOINT AND
 * IMAGES without an oracle (SynergyNet.check_numerics).  Not to be changed while calls are in flight. */
int syn_set_schedule(syn_handle *h, int fusion);

/* Calibration of that verdict on the caller's own crops (uint8 [B,120,120,3] on the device, B <= 256): every block output of the default
 * schedule is compared with the exact fp32-MFMA schedule on this data; a block that differs by more than `tol` (relative to the
 * tensor's maximum; 2e-5 is the natural choice, the schedules agree to ~1e-6) runs the exact kernel from then on, in this handle and
 * in every replica that imports its constants.  Closes the one assumption of the load-time analysis that is not a proof: that the
 * interval bound of a tensor is within 2^8 of its true activations (underflow side).  Returns the number of blocks switched
 * (0 for ResNet-50, which is guarded at run time).  Synchronises; not for use while forwards of this handle are in flight.
 * Scope of the check: the comparison runs block by block at B <= 256, i.e. on the kernels THAT batch size selects (tiled /
 * band-marching early blocks, features.8-13 one block per launch).  The whole-face row-marching kernels of B >= 352 ... 513 and the
 * chain launches use the same operand pieces, scales and range proofs but another summation order; the verdict is taken as valid for
 * them because what it measures -- operand underflow of a block's input and weights -- does not depend on the order of the sum. */
int syn_backbone_calibrate(syn_handle *h, const uint8_t *crops_u8, int B, float tol, void *stream);

/* ResNet-50 (ReLU, no static activation bound): the fp16 convolutions are guarded at RUN time.  Every tensor they split reports
 * max |x| into a per-forward status array; when one leaves [2^-10, 6e4] the head kernel returns NaN for that forward (loud, no host
 * synchronisation).  syn_backbone_range_status synchronises the device, copies the maxima of the last forward (slot 0 = max-pool
 * output, 1 + i = convolution i in state_dict order; unguarded slots read 1) and returns the number of tensors outside the window;
 * with fallback != 0 a violation switches the handle to the exact fp32-MFMA convolutions for all later forwards.  0 for MobileNetV2. */
int syn_backbone_range_status(syn_handle *h, float *layer_max, int max_layers, int fallback);
/* The same switch happens by itself: the head kernel of a poisoned forward also writes a page-locked word of the handle, which the
 * NEXT syn_backbone_forward* reads at its entry (no synchronisation) -- from then on the handle runs the exact convolutions.  A
 * caller that never asks sees one NaN batch (as many as were already in flight), not NaN for ever (reference behaviour being
 * replaced: synergy3DMM.py:156-164 loads any checkpoint and always answers).  syn_backbone_range_events returns how many such
 * automatic or requested switches happened since the weights were loaded (0 | 1); after a switch syn_backbone_range_status
 * reports 0 violations and maxima of 1 (the guard is no longer armed). */
int syn_backbone_range_events(syn_handle *h);

/* BASELINE config 5: the ResNet-50 backbone (reference backbone_nets/resnet_backbone.py:139-254, resnet50 :304-312).
 * `flat` = conv1.weight, bn1.{weight,bias,running_mean,running_var}, then per block of layer1..layer4:
 * conv1, bn1, conv2, bn2, conv3, bn3, [downsample.0, downsample.1], then fc_tex, fc_ori, fc_shape, fc_exp
 * ({weight,bias} each) -- the reference's state_dict order.  After this call syn_backbone_forward*() run ResNet-50
 * and return the first 62 of its 102 outputs (ori 12 | shape 40 | exp 10; the texture head is dropped), which is the
 * adapter the reference's own wrapper would need.  pool = the 2048-d pooled feature. */
size_t syn_resnet50_flat_count(void);
int syn_load_backbone_resnet50(syn_handle *h, const float *flat, size_t n_floats);
double syn_resnet50_flops_per_face(void);

/* ParamsPack (utils/params.py:10-35) + the register_buffer block (synergy3DMM.py:95-105).
 * HOST arrays: w_shp [3*n_vert,40], w_exp [3*n_vert,10], u [3*n_vert] (= u_shp+u_exp),
 * param_mean/param_std [>=62] (first 62 used), keypoints [3*n_lmk] flat indices into the
 * 3*n_vert axis ordered [3k,3k+1,3k+2] (utils/io.py:78-81). */
int syn_load_basis(syn_handle *h, const float *w_shp, const float *w_exp, const float *u,
                   const float *param_mean, const float *param_std,
                   const int64_t *keypoints, int n_lmk, int n_vert);

/* Multi-GPU: rank 0 loads from host arrays, exports its packed constants into a caller
 * device buffer, the caller broadcasts that buffer (RCCL, e.g. torch.distributed.broadcast)
 * and every other rank imports it.  No cross-GPU traffic after that. */
size_t syn_constants_bytes(syn_handle *h);
int syn_export_constants(syn_handle *h, void *dev_dst, size_t bytes, void *stream);
int syn_import_constants(syn_handle *h, const void *dev_src, size_t bytes, void *stream);
/* The same hand-off in ONE call for a caller that holds an RCCL communicator instead of torch.distributed (what replaces the
 * reference's nn.DataParallel replication, benchmark.py:112 / main_train.py:176): collective over `nccl_comm` (an ncclComm_t; every
 * rank calls it with its own handle, on the device the communicator was created for) -- rank `root` exports, ncclBroadcast ships the
 * size and then the blob, every other rank imports; returns after the stream has drained.  The library does not link RCCL: it
 * calls the instance the process already holds (the caller's, or torch's librccl.so.1), so the communicator and the code that
 * uses it always belong together (a process that holds more than one RCCL names the owner of the communicator with the environment
 * variable SYNERGY_HIP_RCCL_LIB = path of that librccl).  SYN_ERR_INVALID for a NULL communicator, SYN_ERR_NOT_LOADED when the process
 * holds no RCCL.
 * A COLLECTIVE THAT FAILS, FAILS ON EVERY RANK (csrc/bcast_protocol.h): a root handle without constants, or one that cannot stage its
 * blob, reports that inside the first broadcast (size word 0 + its code) and EVERY rank returns that code (SYN_ERR_NOT_LOADED /
 * SYN_ERR_HIP); a rank that cannot stage or import the blob reports it in an ncclAllReduce(max) of the status and every rank returns
 * it -- no rank is left waiting in a collective another rank never enters, and the communicator stays usable.  (The one local exit:
 * a rank that cannot allocate the 256-byte word buffer has no device memory to join a collective with.) */
int syn_bcast_constants(syn_handle *h, void *nccl_comm, int root, void *stream);
/* The 256-byte header syn_export_constants would write for this handle NOW (what it holds: arch, backbone / basis present, vertex and
 * landmark counts, total bytes), into HOST memory, no device call -- how a host mirror learns what a syn_bcast_constants import gave it. */
int syn_describe_constants(syn_handle *h, void *host_header, size_t bytes);

/* Device-free twins of the hand-off: neither makes a HIP call, so a loader process (or a test) without a GPU can produce and
 * vet the blob.  syn_pack_constants_host writes, from HOST arrays (same meaning as syn_load_backbone / _resnet50 / syn_load_basis;
 * backbone_flat or the whole basis group may be NULL), exactly the bytes syn_export_constants produces after the same loads;
 * arch 0 = mobilenet_v2, 1 = resnet50.  syn_check_constants_host applies syn_import_constants' acceptance checks (magic, version,
 * sizes vs this library's network tables, payload within the buffer) to a host copy of a blob. */
size_t syn_pack_constants_host_bytes(int arch, int have_backbone, int n_vert, int n_lmk);
int syn_pack_constants_host(int arch, const float *backbone_flat, size_t n_floats, const float *w_shp, const float *w_exp,
                            const float *u, const float *param_mean, const float *param_std, const int64_t *keypoints,
                            int n_lmk, int n_vert, void *host_dst, size_t bytes);
int syn_check_constants_host(const void *host_blob, size_t bytes);

/* ---- compute ------------------------------------------------------------------------ */

/* Bytes of activation workspace the library keeps for a batch of B faces. */
size_t syn_workspace_bytes(syn_handle *h, int B);

/* forward_test (synergy3DMM.py:151-154 -> mobilenetv2_backbone.py:173-189).
 * img   [B,3,120,120] fp32 NCHW, already (x-127.5)/128 (synergy3DMM.py:189-192)
 * param [B,62] (ori 0:12, shape 12:52, exp 52:62);  pool [B,1280] or NULL. */
int syn_backbone_forward(syn_handle *h, const float *img, int B, float *param, float *pool,
                         void *stream);

/* Same, from uint8 crops [B,120,120,3] (HWC, BGR as cv2 delivers them); the
 * (x-127.5)/128 normalisation and the HWC->CHW permute (synergy3DMM.py:189-192) are fused
 * into the first convolution. */
int syn_backbone_forward_u8(syn_handle *h, const uint8_t *img_hwc, int B, float *param,
                            float *pool, void *stream);

/* Pre-processing of get_all_outputs (synergy3DMM.py:187-188): crop_img (utils/inference.py:95-125, zero padding outside
 * the frame) + cv2.resize(..., (120,120), INTER_LANCZOS4) for B detections of ONE frame, on device.
 * frame [H,W,3] uint8 HWC; box [B,4] = int(round()) of the enlarged square box (sx,sy,ex,ey);
 * xofs/yofs [B,120] = first of the 8 source taps in crop coordinates, xcoef/ycoef [B,120,8] = OpenCV's 11-bit fixed
 * point Lanczos-4 weights (the host computes these tables: synergynet_amd/inference.py lanczos4_tables);
 * out [B,120,120,3] uint8, the input of syn_backbone_forward_u8.  All pointers are device pointers. */
int syn_crop_resize(syn_handle *h, const uint8_t *frame, int H, int W, const int *box, const int *xofs,
                    const int16_t *xcoef, const int *yofs, const int16_t *ycoef, uint8_t *out, int B, void *stream);

/* The same for the faces of SEVERAL frames in ONE launch (the reference loops over faces inside get_all_outputs, synergy3DMM.py:176-191,
 * and over images outside it; a caller with a list of frames -- synergynet_amd get_all_outputs_batch -- stages all of them in one device
 * block): frames = base of that block, frame_off [n_frames] byte offset of each frame in it, frame_dim [n_frames][2] = H, W of each,
 * face_frame [B] = frame index of face b; box / tables / out per face as above.  Same arithmetic per face: same bytes. */
int syn_crop_resize_frames(syn_handle *h, const uint8_t *frames, const long long *frame_off, const int *frame_dim,
                           const int *face_frame, const int *box, const int *xofs, const int16_t *xcoef, const int *yofs,
                           const int16_t *ycoef, uint8_t *out, int B, void *stream);

/* reconstruct_vertex_62 (synergy3DMM.py:116-149) fused with the ROI affine of
 * _predict_vertices (utils/inference.py:127-138).
 * param [B,param_len] whitened; param_len must be 62 (else SYN_ERR_PARAM_LEN).
 * dense: 0 -> out [B,3,n_lmk], 1 -> out [B,3,n_vert].
 * transform: y -> 121 - y (synergy3DMM.py:139,147).
 * roi [B,5] (sx,sy,ex,ey,score) or NULL for no ROI affine (the batched torch method). */
int syn_reconstruct(syn_handle *h, const float *param, int B, int param_len, int dense,
                    int transform, const float *roi, float *out, void *stream);

/* Same contraction into a PITCHED output: the rows of out (one per face and coordinate) start row_pitch floats apart,
 * row_pitch >= n (n = n_vert or n_lmk); syn_reconstruct is row_pitch = n.
 * pad_writable != 0: the caller owns columns [n, row_pitch) of every row too (out is B*3*row_pitch floats long, the last
 * row included) and allows them to be overwritten with unspecified finite values; 0: they are left untouched.
 * Why: the reference's result is a [B,3,53215] tensor (synergy3DMM.py:131-147) whose rows are 212860 bytes, so packed rows
 * are only 4-byte aligned and every store run shares HBM lines with its neighbours (measured 3.3 TB/s of writes).  With
 * row_pitch = n rounded up to a multiple of 128 floats (53248), a 128-byte aligned out and writable pad columns every run
 * is whole lines (5.2 TB/s) and the kernel takes its branch-free store path.  A torch view `storage[:, :, :n]` of a
 * [B,3,row_pitch] allocation has the reference's shape and values. */
int syn_reconstruct_pitched(syn_handle *h, const float *param, int B, int param_len, int dense,
                            int transform, const float *roi, float *out, int row_pitch, int pad_writable, void *stream);

/* ---- mesh consumers: what the reference's demo does with the meshes (utils/render.py:31-50) ----
 * syn_load_triangles: the mesh topology (param_pack `tri`, 0-based, [ntri,3] int32 host pointer), once per handle;
 * replaces the `triangles` argument of Sim3DR.get_normal / rasterize (Sim3DR/Sim3DR.py:8,14). */
int syn_load_triangles(syn_handle *h, const int32_t *tri, int ntri, int nver);

/* Sim3DR.get_normal (Sim3DR/Sim3DR.py:8-11 -> lib/rasterize_kernel.cpp:158-215) and, when light != NULL, the vertex
 * colours of RenderPipeline.__call__ (Sim3DR/lighting.py:37-64) for F meshes in one launch chain.
 * vertices: device, planar = 1 -> [F,3,nver] (what syn_reconstruct writes), 0 -> [F,nver,3] (the reference's layout),
 *           planar = p >= nver (p > 1) -> [F,3,p][:, :, :nver], rows p floats apart: the pitched tensor syn_reconstruct_pitched
 *           writes, consumed in place (no packed copy anywhere between reconstruction and rendering);
 * normal, light: device [F,nver,3] (light may be NULL);
 * cfg16: HOST pointer to 16 floats: intensity_ambient, color_ambient[3], intensity_directional, color_directional[3],
 *        intensity_specular, specular_exp, light_pos[3], view_pos[3] (lighting.py:24-32). */
int syn_mesh_shade(syn_handle *h, const float *vertices, int F, int planar, const float *cfg16, float *normal,
                   float *light, void *stream);

/* Sim3DR.rasterize (Sim3DR/Sim3DR.py:14-29 -> lib/rasterize_kernel.cpp:219-287, alpha = 1 as the binding defaults):
 * draws F meshes, in order, into image [H,W,channels] uint8 (device, in place), each with a fresh z-buffer, i.e. the
 * result of calling the reference once per mesh on the same background.  colors: device [F,nver,channels] float32 in [0,1].
 * F <= 254, channels <= 4.  reverse: flip y on write (rasterize_kernel.cpp:270). */
int syn_rasterize(syn_handle *h, const float *vertices, const float *colors, int F, int planar, int channels,
                  uint8_t *image, int H, int W, int reverse, void *stream);

/* cv2.addWeighted(a, alpha, b, beta, 0) for uint8 arrays of n bytes (utils/render.py:45), device pointers. */
int syn_add_weighted(syn_handle *h, const uint8_t *a, float alpha, const uint8_t *b, float beta, uint8_t *out,
                     size_t n, void *stream);

/* ---- FaceBoxes face detector: the boxes get_all_outputs crops (synergy3DMM.py:169-171) ----
 * syn_detector_flat_count / syn_load_detector: FaceBoxesNet's state_dict (FaceBoxes/models/faceboxes.py:64-114) flattened in
 * forward order -- conv1, conv2, inception{1,2,3}.{branch1x1, branch1x1_2, branch3x3_reduce, branch3x3, branch3x3_reduce_2,
 * branch3x3_2, branch3x3_3}, conv3_1, conv3_2, conv4_1, conv4_2: weight [cout,cin,k,k] | bn weight | bias | running_mean |
 * running_var;  then loc.i, conf.i (i = 0..2): weight | bias.  Host pointer. */
size_t syn_detector_flat_count(void);
int syn_load_detector(syn_handle *h, const float *flat, size_t count);
/* number of priors for a network input of Hs x Ws pixels (FaceBoxes/utils/prior_box.py:20) */
int syn_detector_prior_count(int Hs, int Ws);
/* FaceBoxes.__call__ (FaceBoxes/FaceBoxes.py:60-127) on one uint8 BGR frame [H,W,3] (device): optional bilinear down-scaling
 * to Hs x Ws = int(scale*H) x int(scale*W), which the CALLER evaluates in double precision exactly like :63-75 (a float32
 * product lands one pixel off on ~8 % of frame sizes); Hs = H, Ws = W, scale = 1 for no scaling.  Then mean subtraction,
 * network, priors, decoding (boxes * (Ws,Hs,Ws,Hs) / scale, :101-104), score > conf_thr, top_k by score (any number of
 * candidates; top_k <= 8192), NMS (cpu_nms.pyx semantics, IoU >= nms_thr suppresses), first keep_top_k rows.
 * dets: device [keep_top_k,5] (x1, y1, x2, y2, score) in original-frame pixels, score-descending; n_dets: HOST int, rows
 * valid (the call synchronises `stream`).  The vis_thres filter (:133-140) is the caller's. */
int syn_detect(syn_handle *h, const uint8_t *frame, int H, int W, int Hs, int Ws, float scale, float conf_thr, float nms_thr,
               int top_k, int keep_top_k, float *dets, int *n_dets, void *stream);

/* calc_nme (benchmark_aflw2000.py:107-139): fit [N,2,68] fitted landmarks in 120x120 crop coordinates, gt [N,3,68] ground truth
 * in image coordinates, roi [N,4] crop boxes (sx, sy, ex, ey) -> nme [N] float32.  All device pointers. */
int syn_nme(syn_handle *h, const float *fit, const float *gt, const float *roi, float *nme, int N, void *stream);

/* predict_pose (utils/inference.py:146-157 -> parse_pose :86-92 -> P2sRt :33-43 ->
 * matrix2angle_corr :45-62).  angles [B,3] degrees (double, like the reference's python
 * floats), t3d [B,3] fp32 with the ROI affine on x,y; roi may be NULL. */
int syn_pose(syn_handle *h, const float *param, int B, const float *roi, double *angles,
             float *t3d, void *stream);

/* Landmarks AND pose of a batch in one launch: lmk [B,3,n_lmk] (packed rows) = syn_reconstruct(dense = 0, roi) and angles / t3d =
 * syn_pose(roi) -- what the reference's get_all_outputs computes per face besides the mesh (synergy3DMM.py:194-201: predict_sparseVert
 * + predict_pose; utils/inference.py:127-157).  The two calls above remain the boundary; this one exists because as separate calls
 * the three dependent launches (prologue, contraction, pose) are ~19 us of a 128-face landmarks-only step, one launch is ~6.  The
 * contraction runs as plain fp32 multiply-adds on the exact fp32 landmark basis (no fp16 pieces): equal to syn_reconstruct's landmarks
 * to fp32 rounding (~1e-6 of the largest coordinate); translation bit-identical to syn_pose, angles within 1e-4 degree of it.  SYN_ERR_PARAM_LEN for param_len != 62. */
int syn_landmarks_pose(syn_handle *h, const float *param, int B, int param_len, int transform, const float *roi /*nullable [B,5]*/,
                       float *lmk, double *angles /*[B,3] degrees*/, float *t3d /*[B,3]*/, void *stream);

/* predict_pose(..., ret_mat=True) (utils/inference.py:146-157): parse_pose's P = [R | t3d] "without scale" (:86-92),
 * pmat [B,3,4] fp32 row-major; R = normalised rows 0,1 of the de-whitened camera matrix and their cross product, column 3 =
 * the de-whitened translation WITHOUT the ROI affine (the reference concatenates P before predict_pose rescales t3d). */
int syn_pose_matrix(syn_handle *h, const float *param, int B, float *pmat, void *stream);

/* ---- introspection (bench / profiling) --------------------------------------------- */

/* Number of kernel launches one per-layer syn_backbone_forward issues. */
int syn_backbone_launch_count(syn_handle *h);

/* Runs one forward with a HIP event recorded on the library's stream after every kernel launch.
 * Returns the number of launches n (<= max_launches) or a negative syn_status; for launch i:
 * feature_of_launch[i] = index into .features the launch completes (1 = fused stem + features.1,
 * 19 = pool + heads, 100 * a + b = one launch covering .features[a] .. [b]), ms_of_launch[i] = event-to-event time, flops_of_launch[i] = algorithmic FLOPs
 * of the layers it covers for the whole batch (no halo / padding work counted).  Synchronises. */
int syn_backbone_profile(syn_handle *h, const uint8_t *img_hwc, int B, int max_launches,
                         int *feature_of_launch, float *ms_of_launch, double *flops_of_launch);

/* The dense reconstruction (transform = 1) into a pitched output with HIP events around its kernels, on the default stream;
 * synchronises.  ms2[0] = per-face prologue, ms2[1] = the contraction + pose epilogue + mesh stores -- the HBM-write-bound kernel
 * of the path (B * 3 * n_vert * 4 bytes).  Arguments as syn_reconstruct_pitched. */
int syn_reconstruct_profile(syn_handle *h, const float *param, int B, const float *roi, float *out, int row_pitch, int pad_writable,
                            float *ms2);

/* Algorithmic FLOPs per face of the whole backbone / of its pointwise (MFMA) convolutions. */
double syn_backbone_flops_per_face(void);
double syn_pointwise_flops_per_face(void);

#ifdef __cplusplus
}
#endif
#endif /* SYNERGY_HIP_H */
