/*
 * dinov2_hip.h -- C-ABI of the MI355X-native DINOv2 forward (libdinov2_hip.so).
 *
 * Drop-in boundary for the hot path of lavaman131/dinov2.cpp: everything the reference does between
 * `dino_model_load` and the return of `dino_predict`.  Plain pointers and sizes only: no ggml, OpenCV,
 * torch or C++ types cross this boundary.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference repo).  Status codes, never abort/assert/throw (the reference mixes
 * bool / empty unique_ptr / assert / exceptions: dinov2.cpp:58,269-272,945-948).
 *
 * Threading: a model is immutable after load and may be shared by any number of sessions; one session
 * = one HIP stream + one workspace and is used by one host thread at a time (the role the caller-owned
 * `ggml_gallocr_t allocr` plays in dinov2.h:111-112).
 */
#ifndef DINOV2_HIP_H
#define DINOV2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINOV2_HIP_ABI_VERSION 1

typedef struct dinov2_hip_model dinov2_hip_model;     /* replaces `struct dino_model`     dinov2.h:49-55   */
typedef struct dinov2_hip_session dinov2_hip_session; /* replaces `ggml_gallocr_t allocr` dinov2.h:111-112 */

enum dinov2_hip_status {
    DINOV2_HIP_OK = 0,
    DINOV2_HIP_ERR_IO = 1,          /* file cannot be opened / short read   (dinov2.cpp:269-272 returns false) */
    DINOV2_HIP_ERR_FORMAT = 2,      /* not GGUF, missing KV or tensor       (dinov2.cpp:58 asserts)            */
    DINOV2_HIP_ERR_UNSUPPORTED = 3, /* tensor type / shape outside the DINOv2 family                            */
    DINOV2_HIP_ERR_INVALID = 4,     /* bad argument                                                             */
    DINOV2_HIP_ERR_HIP = 5,         /* HIP runtime error (message carries hipGetErrorString)                    */
    DINOV2_HIP_ERR_NO_HEAD = 6      /* classify requested on a model loaded without a classifier                */
};

enum dinov2_hip_dtype { DINOV2_HIP_F16 = 0, DINOV2_HIP_BF16 = 1 };

/* Input image memory layouts accepted by dinov2_hip_predict. */
enum dinov2_hip_layout {
    DINOV2_HIP_BGR_HWC = 0, /* continuous CV_32FC3 cv::Mat as handed to dino_predict (dinov2.cpp:900, 914-931) */
    DINOV2_HIP_RGB_CHW = 1, /* the planar "input" tensor dino_predict uploads        (dinov2.cpp:629-631, 933) */
    DINOV2_HIP_U8_BGR_HWC = 2 /* RAW 8-bit BGR images [B, h, w, 3] as cv::imread returns them (inference.cpp:36): the library
                                 runs dino_preprocess (or dino_classify_preprocess with DINOV2_HIP_CLASSIFY) on the device
                                 first (dinov2.cpp:106-156); `data` points to uint8_t, height/width are the RAW size */
};

/* predict flags */
#define DINOV2_HIP_CLASSIFY 1u /* dino_params.classify (dinov2.h:63): run forward_head, patch view includes registers */

typedef struct dinov2_hip_load_opts {
    int32_t device;        /* HIP device ordinal (the reference picks its backend by #ifdef, dinov2.cpp:241-261)   */
    int32_t compute_dtype; /* enum dinov2_hip_dtype: MFMA input type of every weight GEMM and of attention          */
    int32_t classify;      /* dino_params.classify at load: read labels + classifier (dinov2.cpp:297-305)           */
    int32_t skip_tensor_data; /* 1: parse metadata and allocate the arena but leave it unfilled -- for ranks that
                                 receive the arena by RCCL broadcast (dinov2_hip_model_arena)                        */
    int32_t quirk_pool_const_divisor;      /* 1 (default): pooled = sum / (img_size/patch)^2  (dinov2.cpp:794,800-803) */
    int32_t quirk_pool_includes_registers; /* 1 (default): register tokens are pooled too      (dinov2.cpp:772-776)     */
    int32_t batch_invariant; /* kept for ABI compatibility; the value is ignored.  EVERY plan is batch-invariant: all kernels sum K in
                                one order, so B images == B independent forwards BIT FOR BIT, whatever the batch size, the chunking
                                of an over-long batch or the number of devices a dinov2_hip_group shards it over.  (Round 2 / early
                                round 3 had an opt-in mode, 0, that split K at tiny batches; the small-tile plans that replaced it are
                                faster than it was and keep the bits: profiles/r03_small_m_gemm.md.)                              */
    int32_t ln_fold; /* LayerNorm folded into the GEMMs on either side of it: 0 = the library's choice, 1 = on
                        (where the model allows it: hidden % 128 == 0 and <= 1 536), -1 = off: separate LayerNorm launches that round
                        f16(LN(x)) exactly where ggml does.  On: the residual epilogues emit f16(gamma x) + row statistics and the QKV /
                        FFN-in epilogues apply mean, rstd and beta -- 2 of 7 launches per layer less; the activation is rounded BEFORE the
                        normalisation instead of after it (same 2^-11 relative rounding per element; parity numbers: profiles/r06_*).      */
    int32_t reserved[8];
} dinov2_hip_load_opts;

/* dino_hparams (dinov2.h:25-47) plus what the loader derives from the tensor list. */
typedef struct dinov2_hip_hparams {
    uint32_t hidden_size, num_hidden_layers, num_attention_heads, num_classes;
    uint32_t num_register_tokens, patch_size, img_size, ftype;
    float eps;             /* 1e-6 (dinov2.h:33; dino_params.eps is unused by the reference) */
    uint32_t ffn_hidden;   /* fc1 rows, or weights_out columns for SwiGLU                                */
    uint32_t swiglu;       /* reference selects by num_hidden_layers == 40 (dinov2.cpp:740); here: tensor presence */
    uint32_t has_classifier;
    uint32_t weight_type;  /* ggml type id of the 2-D weights as stored in the file (1 f16, 8 q8_0, ...)  */
    uint32_t compute_dtype;
} dinov2_hip_hparams;

typedef struct dinov2_hip_input {
    const float *data; /* [batch] images, f32, already preprocessed (dino_preprocess output)               */
    int32_t batch;     /* the reference is batch 1 (dinov2.cpp:630); B images = B independent forwards     */
    int32_t height;    /* multiples of patch_size, like img.size() in dinov2.cpp:908                       */
    int32_t width;
    int32_t layout;    /* enum dinov2_hip_layout                                                           */
    int32_t on_device; /* 0: host memory (copied H2D each call); 1: device memory on the model's device.
                          Device inputs must be COMPLETE before the call: a session created with stream = NULL runs on
                          its own non-blocking stream, which is not ordered after work the caller queued on another
                          stream (e.g. the kernel that produced the images) -- synchronise that stream first, or hand
                          the producer's stream to dinov2_hip_session_create.  Same for on_device outputs: read them
                          after dinov2_hip_session_sync (or on the session's stream).                              */
} dinov2_hip_input;

/* Caller-allocated outputs; any pointer may be NULL.  Printing top-k stays in the caller. */
typedef struct dinov2_hip_output {
    float *cls;          /* [B, H]            "cls_token"    dinov2.cpp:764-768                                   */
    float *patch_tokens; /* [B, P, H] features; [B, R+P, H] with CLASSIFY ("patch_tokens", dinov2.cpp:770-789);
                            row = patch index y*w0+x like the cv::Mat of dinov2.cpp:979-992                        */
    float *logits;       /* [B, C]            unnamed tensor of dinov2.cpp:811-812 (CLASSIFY only)                */
    float *probs;        /* [B, C]            "probs"        dinov2.cpp:815-820  (CLASSIFY only)                  */
    int32_t *topk_ids;   /* [B, topk]         sorted descending like dinov2.cpp:961-965 (CLASSIFY only)           */
    float *topk_probs;   /* [B, topk]         (the reference's preds[] holds uint32(prob) == 0, dinov2.cpp:975)   */
    int32_t topk;        /* dino_params.topk (dinov2.h:59)                                                        */
    int32_t on_device;   /* 0: host pointers (call returns after the copy-out); 1: device pointers, async         */
} dinov2_hip_output;

/* -- load (replaces dino_model_load, dinov2.h:98-99 / dinov2.cpp:239-352) ---------------------------- */
void dinov2_hip_default_load_opts(dinov2_hip_load_opts *opts);
int dinov2_hip_model_load(const char *gguf_path, const dinov2_hip_load_opts *opts, dinov2_hip_model **out,
                          char *err, size_t errlen);
/* replaces the caller-side frees of inference.cpp:70-73 */
void dinov2_hip_model_free(dinov2_hip_model *model);
/* replaces reads of model.hparams (dinov2.cpp:276-299) */
int dinov2_hip_model_hparams(const dinov2_hip_model *model, dinov2_hip_hparams *out);
/* replaces model.hparams.id2label.at(id) (dinov2.cpp:301-305, 972); NULL when out of range */
const char *dinov2_hip_model_label(const dinov2_hip_model *model, int32_t id);
/* The packed device weight arena (one allocation, like model.buffer of dinov2.cpp:341): lets a multi-GPU
 * host broadcast rank 0's converted weights over RCCL/xGMI instead of re-reading the GGUF 8 times. */
int dinov2_hip_model_arena(dinov2_hip_model *model, void **device_ptr, size_t *bytes);

/* -- session (replaces ggml_gallocr_new / reuse across calls, inference.cpp:63, realtime.cpp:62) ---- */
/* `stream`: a hipStream_t to run on, or NULL to let the session create its own. */
int dinov2_hip_session_create(dinov2_hip_model *model, void *stream, dinov2_hip_session **out, char *err,
                              size_t errlen);
void dinov2_hip_session_free(dinov2_hip_session *session);
/* Bytes of device workspace one predict of this shape needs (cf. ggml_gallocr_alloc_graph, dinov2.cpp:910). */
size_t dinov2_hip_workspace_bytes(const dinov2_hip_model *model, int32_t batch, int32_t height, int32_t width);
/* Blocks until everything enqueued on the session's stream has finished (ggml_backend_synchronize,
 * inference.cpp:66). */
int dinov2_hip_session_sync(dinov2_hip_session *session);
void *dinov2_hip_session_stream(dinov2_hip_session *session);

/* -- predict (replaces dino_predict, dinov2.h:111-112 / dinov2.cpp:900-999) -------------------------- */
int dinov2_hip_predict(dinov2_hip_session *session, const dinov2_hip_input *in, dinov2_hip_output *out,
                       uint32_t flags, char *err, size_t errlen);

/* Copy-out half of dinov2_hip_predict on its own: the outputs of the session's LAST predict (which may have been called with
 * out = NULL, i.e. forward only) into the caller's buffers.  Lets a host overlap the device -> host copy of batch k with the
 * forward of batch k + 1 on another session (this is what the group's lanes do).  Not available after a predict that had to
 * split an over-long batch into passes. */
int dinov2_hip_fetch(dinov2_hip_session *session, dinov2_hip_output *out, char *err, size_t errlen);

/* -- multi-device group (no reference counterpart: the reference is one backend, batch 1) -----------------
 *    Host threads + sessions per device inside the library; dinov2_hip_group_predict splits the caller's global batch
 *    contiguously (device g owns images [g*B/G, (g+1)*B/G), remainder to the low ranks) and every device writes its results
 *    into the caller's HOST buffers at its shard offset.  Images are independent forwards: no data-path collective.  With
 *    `broadcast` = 1 (default) only device 0 parses / dequantises the GGUF; the packed weight arena reaches the others by ONE
 *    RCCL broadcast over xGMI (single-process ncclCommInitAll + ncclBroadcast; librccl is dlopen'ed on first use).  A device
 *    list that names a device twice (two sessions on one GPU) makes every entry read the file itself. */
typedef struct dinov2_hip_group dinov2_hip_group;
typedef struct dinov2_hip_group_opts {
    dinov2_hip_load_opts load; /* compute dtype, classify, quirks; `device` and `skip_tensor_data` are set per rank          */
    int32_t n_devices;         /* 0: every visible device                                                                     */
    const int32_t *devices;    /* [n_devices] HIP ordinals, or NULL for 0 .. n_devices-1                                      */
    int32_t broadcast;         /* 1: rank 0 loads, RCCL broadcast of the arena; 0: every rank loads the file                  */
    int32_t streams_per_device; /* lanes (host thread + stream + workspace) per device, 1..4; default 2 = the number of jobs
                                   dinov2_hip_group_submit accepts before one must be waited for.  Results do not depend on it.
                                   A lane allocates its workspace (dinov2_hip_workspace_bytes of its shard) and input staging buffer
                                   when it first runs a job; a job goes to the lowest lane with nothing in flight, so callers of
                                   dinov2_hip_group_predict alone (one job in flight) only ever pay for lane 0.                    */
    int32_t reserved[7];
} dinov2_hip_group_opts;
void dinov2_hip_default_group_opts(dinov2_hip_group_opts *opts);
int dinov2_hip_group_create(const char *gguf_path, const dinov2_hip_group_opts *opts, dinov2_hip_group **out, char *err,
                            size_t errlen);
void dinov2_hip_group_free(dinov2_hip_group *group);
int dinov2_hip_group_size(const dinov2_hip_group *group);
/* the model of one rank (hparams / labels are the same on every rank); owned by the group */
dinov2_hip_model *dinov2_hip_group_model(dinov2_hip_group *group, int32_t rank);
/* wall time of the load-time arena broadcast in ms; negative when every rank read the file itself */
double dinov2_hip_group_broadcast_ms(const dinov2_hip_group *group);
/* Where the group's devices sit: one text line per device -- ordinal, PCI bus id, NUMA node, local CPUs (its worker threads are bound
 * to them; DINOV2_HIP_GROUP_NO_AFFINITY=1 turns that off), peer-to-peer reachability of the group's other devices (y / n), and whether
 * its weights came from the file or from the RCCL broadcast (a broadcast that cannot be set up degrades to file reads with a line on
 * stderr; DINOV2_HIP_GROUP_REQUIRE_RCCL=1 makes it an error instead). */
int dinov2_hip_group_describe(const dinov2_hip_group *group, char *out, size_t cap);
/* dino_predict over the whole group: host input [B, ...] (any dinov2_hip_layout), host outputs [B, ...]; returns when every
 * shard has landed.  B < G leaves the high ranks idle.  One call at a time per group; refused (DINOV2_HIP_ERR_INVALID, nothing
 * queued) while a ticket of dinov2_hip_group_submit has not been waited for.  Layout / height / width are checked before
 * anything is copied. */
int dinov2_hip_group_predict(dinov2_hip_group *group, const dinov2_hip_input *in, dinov2_hip_output *out, uint32_t flags,
                             char *err, size_t errlen);
/* The same call in two halves, so that ONE host thread can keep up to `streams_per_device` batches in flight: while a device
 * computes batch k, its other lane copies batch k + 1 in and batch k - 1 out (host -> device copy, forward and device -> host
 * copy are separate turnstiles per device, each passed in submission order).  `in` / `out` are copied; the buffers they point
 * to must stay valid (and unread) until the ticket has been waited for.  Tickets are waited for in submission order.  Results
 * are bit-identical to dinov2_hip_group_predict's.  Page-locked buffers (dinov2_hip_host_alloc) make the copies faster but are
 * not required. */
int dinov2_hip_group_submit(dinov2_hip_group *group, const dinov2_hip_input *in, const dinov2_hip_output *out, uint32_t flags,
                            int64_t *ticket, char *err, size_t errlen);
int dinov2_hip_group_wait(dinov2_hip_group *group, int64_t ticket, char *err, size_t errlen);

/* Page-locked host memory for images / results handed to dinov2_hip_predict or dinov2_hip_group_predict: copies to and from
 * such buffers run at the full PCIe rate and truly asynchronously (pageable memory is staged through bounce buffers by the
 * runtime, at roughly half the rate).  Plain malloc'ed buffers remain valid inputs.  NULL on failure. */
void *dinov2_hip_host_alloc(size_t bytes);
void dinov2_hip_host_free(void *ptr);

/* -- preprocessing (replaces dino_preprocess / dino_classify_preprocess, dinov2.h:93-96,
 *    dinov2.cpp:106-156, without OpenCV).  mode 0: resize to ((w/p)+1)*p x ((h/p)+1)*p; mode 1: resize to 256x256 ignoring
 *    aspect, centre-crop 224.  Input 8-bit BGR interleaved [h, w, 3]; output continuous f32 BGR [out_h, out_w, 3], /255,
 *    bicubic (cv::INTER_CUBIC), (c - mean) / std with the reference's BGR<->mean indexing.  Host implementation; the same
 *    arithmetic runs on the device for DINOV2_HIP_U8_BGR_HWC inputs. */
int dinov2_hip_preprocess_size(int32_t mode, int32_t height, int32_t width, int32_t patch, int32_t *out_h, int32_t *out_w);
int dinov2_hip_preprocess(int32_t mode, const uint8_t *bgr, int32_t height, int32_t width, int32_t patch, float *out);

/* -- feature post-processing (replaces cv::PCA(tokens, noArray(), DATA_AS_ROW, 3) + project of
 *    inference.cpp:76-81).  tokens: [P, H] f32 (P >= 4, 8 <= H <= 4096, |x| within the f16 range), a host pointer, a device
 *    pointer (on_device = 1), or NULL = the patch tokens of image 0 that the session's last dinov2_hip_predict left on the
 *    device (P and H must be theirs) -- the realtime loop's case: nothing but the [P, 3] projection crosses PCIe.
 *    On the device: column means, the H x H covariance (one MFMA GEMM of the centred, transposed f16 tokens with themselves),
 *    a block iteration (8 vectors, CholeskyQR, one launch per step) for the leading eigenvectors, and the projection; on the
 *    host only the 8 x 8 Rayleigh-Ritz problem.  Deterministic.  Outputs (host, any may be NULL): components [3, H] unit
 *    vectors sorted by variance, each oriented so that its largest loading is positive; mean [H]; projection [P, 3] =
 *    (tokens - mean) components^T. */
int dinov2_hip_pca3(dinov2_hip_session *session, const float *tokens, int32_t P, int32_t H, int32_t on_device,
                    float *components, float *mean, float *projection, char *err, size_t errlen);

/* -- quantise a GGUF (replaces dino_model_quantize, dinov2.h:118 / dinov2.cpp:355-453).  Host only.
 *    itype: ggml type id 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0.  2-D tensors named `*weight` are re-encoded, the rest copied. */
int dinov2_hip_quantize(const char *fname_inp, const char *fname_out, int32_t itype, char *err, size_t errlen);

/* Host helper, exposed for parity tests: interpolate_pos_embed (dinov2.h:101-103 / dinov2.cpp:159-225).
 * out: [(1 + h_new*w_new), H] f32. */
int dinov2_hip_interpolate_pos_embed(const dinov2_hip_model *model, int32_t h_new, int32_t w_new, float *out);

/* -- measurement hooks (no reference counterpart; the reference times the whole call, inference.cpp:64-68) */
/* Per-kernel-kind HIP-event timing of subsequent predicts on this session (adds two events per launch). */
int dinov2_hip_session_profile(dinov2_hip_session *session, int32_t enable);
/* Accumulated since enable: for kind k in [0, n): name, total ms, launches.  Returns n (<= max). */
int dinov2_hip_session_profile_read(dinov2_hip_session *session, int32_t max, const char **names, float *total_ms,
                                    int32_t *launches);
/* Debug/parity: copy the f32 token stream [B, T, H] as it stands after `layer` layers (0 = embeddings) of the
 * LAST predict with the same shape re-run up to that point.  Host pointer. */
int dinov2_hip_debug_hidden(dinov2_hip_session *session, const dinov2_hip_input *in, int32_t layer, float *out,
                            char *err, size_t errlen);

int dinov2_hip_abi_version(void);
/* the commit the library was built from ("<12 hex digits>[+dirty]", "unknown" outside a git checkout): measurements taken where there is
 * no repository next to the library (bench.py on a GPU box) stamp themselves with it */
const char *dinov2_hip_build_id(void);

/* -- Environment ----------------------------------------------------------------------------------------------------------
 * The library reads exactly seven environment variables; none is needed in normal use.
 *   DINOV2_HIP_GRAPHS=1      replay a captured hipGraph for a forward that repeats with the same session, input pointer, shape
 *                            and flags (second sighting is captured).  Off by default: the forward is kernel-bound and the
 *                            replay measured no faster on an idle host; it is there for hosts whose launch thread is contended.
 *   DINOV2_HIP_MAX_CHUNK=n   testing aid: split a predict call into passes of at most n images (the split that otherwise only
 *                            happens past 2^31 bytes of activations), to exercise that path at small sizes.
 *   The next four are testing aids that pick a kernel the library would otherwise choose by shape.  They are read ONCE, on first use
 *   (round 5; they used to be getenv() calls on every launch); a test that flips one inside a process uses dinov2_hip_op_set_tuning
 *   (include/dinov2_hip_ops.h), and dinov2_hip_op_gemm_plan reports which GEMM kernels a shape gets under the current setting.
 *   DINOV2_HIP_ATTN_V=1|2|3|4  force the throughput / the software-pipelined attention kernel (normally chosen by workgroup count; 3 and 4 =
 *                            the measured, never auto-selected 64-queries-per-wave variants: two waves per SIMD, and software-pipelined
 *                            with one wave per SIMD).  All give the same bits.
 *   DINOV2_HIP_ATTN_NWV=2|3|4  waves (32-query blocks) per workgroup of the software-pipelined attention kernel (normally 4; 2 for short
 *                            sequences); every size gives the same bits.
 *   DINOV2_HIP_GEMM_TILE=128|256   the small-tile kernel only / 256-row persistent tiles only.
 *   DINOV2_HIP_GEMM_GEN=2|4  which generation of the persistent GEMM runs the 256-row / mixed / one-tile-per-workgroup plans -- 2 = gemm2.hip
 *                            (eight waves, barrier-separated sections), 4 = gemm4.hip (four waves, hand-ordered K loop; the default for
 *                            K >= 1 024).  Both give every row the same bits.  (The LN-fold epilogues exist in gemm4.hip and the small-tile kernel only.)
 *   DINOV2_HIP_LN_FOLD=0|1   what dinov2_hip_load_opts.ln_fold = 0 ("the library's choice") resolves to; unset: off.
 *   DINOV2_HIP_GROUP_NO_AFFINITY=1  the group's worker threads are not bound to the CPUs local to their device.
 *   DINOV2_HIP_GROUP_REQUIRE_RCCL=1  dinov2_hip_group_create fails when librccl cannot be loaded instead of letting every device
 *                            read the GGUF itself.
 * (DINOV2_HIP_LIB, read by the Python binding only, points it at another build of this library.) */

#ifdef __cplusplus
}
#endif
#endif /* DINOV2_HIP_H */
