/* dta_hip.h - C ABI of the MI355X-native (gfx950) Hang2020 hot path.
 *
 * This is the drop-in boundary.  The reference has no native layer: its boundary is the
 * torch.nn.Module contract `cls(bands, classes)` / `forward(x)` of /root/reference/src/models/Hang2020.py
 * plus autograd + torch.optim.Adam.  Each entry point below names the reference interface it replaces; the
 * Python binding a reference maintainer adds is shown in INTEGRATION.md (ctypes, raw device pointers).
 *
 * Conventions
 *  - plain C, no C++/torch types; every pointer is a DEVICE pointer unless said otherwise
 *  - every buffer is allocated by the caller (PyTorch caching allocator) and only borrowed for the call;
 *    the library allocates nothing on the device (one documented exception: the peer-exchange object dta_xchg_*)
 *  - host-side state the library keeps (all of it per process): the thread-local text of dta_last_error(); the
 *    developer switches of the DEVELOPER library only (dta_dev_switches_enabled; the product library reads nothing from
 *    the environment); per launch site, a bit per
 *    device ordinal saying that the function's dynamic-LDS attribute has been set on that device; the optional
 *    profiling event pool of dta_profile_* (one device at a time); peer-exchange objects the caller created.  Nothing
 *    else survives a call: no caches keyed on shapes or pointers, no device allocations
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no internal syncs
 *  - return 0 on success; otherwise non-zero and dta_last_error() describes the failure (thread-local text)
 *  - parameters/gradients use the reference's torch layouts and state_dict shapes
 */
#ifndef DTA_HIP_H
#define DTA_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DTA_ABI_VERSION 2   /* 2: DTA_MAX_YEARS 16, dta_adam_segment::lr, dta_multistage_*, 16 Adam segments per launch */

enum { DTA_F32 = 0, DTA_BF16 = 1 };               /* arithmetic type of the conv contractions */
enum { DTA_NET_HANG2020 = 0, DTA_NET_SPECTRAL = 1, DTA_NET_SPATIAL = 2, DTA_NET_VANILLA = 3 };

typedef struct dta_net_desc {
  int batch, bands, height, width, classes;
  int kind;          /* DTA_NET_* */
  int dtype;         /* DTA_F32 (exact fp32 MFMA) or DTA_BF16 (bf16 MFMA inputs, fp32 accumulate) */
  int training;      /* 1: BatchNorm batch statistics + running-stat update; 0: running statistics */
  int heads_mask;    /* bit L-1 set: compute classifier head L (Hang2020.forward only needs head 3 = 4);
                        | DTA_FORWARD_ONLY: no dta_net_backward will follow on this workspace (inference)
                        | DTA_SKIP_BLEND: Hang2020 forward leaves the sigmoid(alpha) blend to dta_net_loss (joint unused)
                        | DTA_REUSE_PACKED: inference with frozen weights (below) */
  float bn_momentum, bn_eps;
} dta_net_desc;

/* One spectral_network / spatial_network (Hang2020.py:170-240) or the vanilla_CNN conv stack (:33-53).
 * att[L][*]: spectral_attention -> {attention_conv1.weight (C,C,K), .bias, attention_conv2.weight, .bias, 0, 0}
 *            spatial_attention  -> {channel_pool.weight (1,C,1,1), .bias, attention_conv1.weight (1,1,k,k), .bias,
 *                                   attention_conv2.weight, .bias};  vanilla: all null.
 * fc_w/fc_b : classifier{L}.fc1 (vanilla: only index 2 = fc1). */
typedef struct dta_subnet_params {
  const float* conv_w[3]; const float* conv_b[3];
  const float* bn_w[3]; const float* bn_b[3];
  float* bn_rm[3]; float* bn_rv[3]; long long* bn_nbt[3];
  const float* att[3][6];
  const float* fc_w[3]; const float* fc_b[3];
} dta_subnet_params;

/* Gradient destinations, same shapes as the parameters; a null pointer skips that gradient. */
typedef struct dta_subnet_grads {
  float* conv_w[3]; float* conv_b[3];
  float* bn_w[3]; float* bn_b[3];
  float* att[3][6];
  float* fc_w[3]; float* fc_b[3];
} dta_subnet_grads;

/* heads_mask flag: the forward skips what only a backward would read (saved attention state, the bf16 input tiles) */
#define DTA_FORWARD_ONLY 8
#define DTA_SKIP_BLEND 16
/* heads_mask flag, inference only (training == 0 and DTA_FORWARD_ONLY; ignored otherwise): the conv / spectral-attention
 * weight re-layouts and the conv row tables that an earlier forward with the SAME descriptor, parameter table and workspace
 * left in the workspace are used as they are -- the caller guarantees that those weights have not changed since that
 * call and that nothing else wrote to the workspace.  The first forward on a workspace must not carry the flag.
 * (Tile prediction, reference predict.py:140-151 / main.py:165-205: one trained model, thousands of 64-crop batches --
 * the re-layout of 15 networks' weights is 19 us of a 145 us MultiStage.predict_step.)  BatchNorm running statistics,
 * biases and the classifier heads are read from the parameter table in every call either way. */
#define DTA_REUSE_PACKED 32

int dta_abi_version(void);
/* Hash of the sources this library was built from (python -m deeptreeattention_amd.build): measurement artifacts record
 * it, bench.py refuses to quote counters taken on another build. */
const char* dta_build_id(void);
const char* dta_last_error(void);

/* Bytes of scratch + saved-activation workspace dta_net_forward/backward need for `d` (same blob for both;
 * keep it alive and untouched between a training forward and its backward). */
size_t dta_net_workspace_bytes(const dta_net_desc* d);

/* Replaces Hang2020.forward (Hang2020.py:251-263), spectral_network.forward (:226-240), spatial_network.forward
 * (:190-204), vanilla_CNN.forward (:45-53) incl. every conv_module / attention / Classifier inside.
 *  nets   : 2 entries {spectral, spatial} for DTA_NET_HANG2020, otherwise 1
 *  alpha  : float64 scalar (Hang2020.alpha), HANG2020 only
 *  x      : float32 NCHW contiguous [batch][bands][height][width]
 *  scores : [net][L] -> float32 [batch][classes] outputs of the classifier heads selected by heads_mask
 *  joint  : HANG2020: sigmoid(alpha)-blended scores; VANILLA: fc1 output; else unused (may be null)
 * DTA_NET_VANILLA: the head is Linear(512, classes) as in the reference (:43), so only patches whose twice-pooled map
 * flattens to 128 * (H/4) * (W/4) = 512 features are accepted (the descriptor is rejected otherwise). */
int dta_net_forward(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const float* x,
                    void* workspace, float* const scores[2][3], float* joint, void* stream);

/* Replaces autograd's backward through the module above.  `workspace` is the blob the matching training (or eval)
 * forward filled.  dscores[net][L] / djoint: gradients wrt the forward outputs (null = that output unused; for
 * HANG2020 pass djoint, for the others dscores; HANG2020 with djoint == NULL and dscores set is the all-heads mode of
 * the Hang et al. multi-head loss: every listed head of both branches back-propagates, the blend and alpha do not).
 * Every non-null gradient buffer in `grads`, and `dalpha`, must arrive
 * ZERO-FILLED (split-K partial sums are accumulated with atomics); on return it holds the gradient.
 * phases: bit 0 = everything except the first conv's weight gradient, bit 1 = the first conv's weight gradient
 * (the largest and last piece); 3 = all.  Two calls (1, then 2) let the caller start the gradient all-reduce of
 * the rest (RCCL on a side stream) while the first conv's weight gradient is still being computed. */
int dta_net_backward(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, void* workspace,
                     const float* const dscores[2][3], const float* djoint, const dta_subnet_grads* grads,
                     double* dalpha, int phases, void* stream);

/* The same two calls for a batch whose input is ALREADY the first conv's bf16 tiles (dta_preprocess_crops_tiles below:
 * raw crops -> tiles in one launch): bf16 mode, one input tensor (every kind but the ensemble), patches whose tile is a
 * single band (11x11-class).  x_tiles: bf16 [batch][ceil(bands / 16)][height * width][16], bands past `bands` zero; it
 * must stay valid and unchanged until the matching backward, which reads it again for the first conv's weight gradient. */
int dta_net_forward_tiles(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                          void* workspace, float* const scores[2][3], float* joint, void* stream);
int dta_net_backward_tiles(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                           void* workspace, const float* const dscores[2][3], const float* djoint,
                           const dta_subnet_grads* grads, double* dalpha, int phases, void* stream);

/* Data-parallel form of dta_net_backward / dta_net_backward_tiles (x_tiles NULL: the workspace's own tiles).  One extra
 * destination: dalpha_f32 (may be NULL) receives d(alpha) rounded to float32 -- ONE rounding of the finished float64 sum,
 * written by the launch that ends the call (any call with phases & 1), no float atomics: reruns give the same bits.  A
 * data-parallel caller points it at a slot of its flat float32 gradient buffer, so that alpha's gradient takes
 * part in the buffer's all-reduce without copy kernels around the collective (reference train.py:89-98 lets Lightning's
 * DDP all-reduce every parameter; here that is at most two collectives per step). */
int dta_net_backward_dp(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                        void* workspace, const float* const dscores[2][3], const float* djoint,
                        const dta_subnet_grads* grads, double* dalpha, float* dalpha_f32, int phases, void* stream);

/* dta_net_backward_dp for a caller that exchanges gradients through a peer exchange object (dta_xchg_* below) whose
 * buffer has a head / tail split (dta_xchg_set_split): the whole backward in one call, ordered so that the exchange's HEAD
 * segment -- every gradient except the first conv's weights, 76 % of the bytes -- is complete before the first conv's
 * weight-gradient launch, whose spare workgroups (the 16 CUs that launch leaves idle) then sum this rank's shard of the
 * head over the ranks through peer memory WHILE the matrix cores compute the last gradient (reference train.py:89-98:
 * DDP overlaps the gradient all-reduce with the backward).  The dta_xchg_adam_step / dta_xchg_allreduce that follows on
 * the same stream finds the head summed and only has the tail left.  grads must point into dta_xchg_grad_buffer(xchg);
 * alpha_slot: index of alpha's float32 exchange slot inside the head (or -1).  When the plan has no combined kernel
 * (fp32 mode, other shapes) this is exactly dta_net_backward_dp(phases = 3) and the exchange sums both segments itself. */
struct dta_xchg;
int dta_net_backward_xchg(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                          void* workspace, const float* const dscores[2][3], const float* djoint,
                          const dta_subnet_grads* grads, double* dalpha, struct dta_xchg* xchg, long long alpha_slot,
                          void* stream);

/* Loss head of the network whose forward just ran on `workspace`: the Hang2020 blend (Hang2020.py:260-261; the forward
 * was built with DTA_SKIP_BLEND), F.cross_entropy(scores, labels, weight) (src/main.py:78), d(loss)/d(scores) and the
 * scalar loss in ONE launch (dta_net_forward + dta_weighted_ce take three for the same).
 *  joint   : Hang2020: receives the blended scores [batch][classes] (may be NULL); other kinds: the scores to use (input)
 *  dlogits : d(loss)/d(scores) [batch][classes], may be NULL (validation)
 *  scratch : batch + 2 floats; the LAST 32-bit word must be zero on entry and is zero again on return (block counter) */
int dta_net_loss(const dta_net_desc* d, const double* alpha, void* workspace, const long long* labels, const float* weight,
                 float* joint, float* loss, float* dlogits, float* scratch, void* stream);
/* Forward AND loss of a single-score network (Hang2020, vanilla_CNN) in one call: what TreeModel.training_step does in
 * `y_hat = self.model.forward(images); loss = F.cross_entropy(y_hat, y, weight=self.loss_weight)` (reference src/main.py:77-78;
 * validation_step :88-89 with dlogits = NULL).  Same results as dta_net_forward (built with DTA_SKIP_BLEND) followed by
 * dta_net_loss.  For a training-mode Hang2020 on 11x11 patches the third stage of both branches, the two last heads, the
 * blend and the loss are ONE launch (the forward ends two launches earlier than dta_net_forward + dta_net_loss).
 *  x / x_tiles : the input as float32 NCHW [batch][bands][H][W], or (bf16 mode, x = NULL) as the first conv's tiles
 *                (dta_preprocess_crops_tiles) -- exactly one of the two
 *  labels, weight, joint, loss, dlogits, scratch : as for dta_net_loss (vanilla_CNN: joint receives the scores, required) */
int dta_net_forward_loss(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const float* x,
                         const void* x_tiles, void* workspace, const long long* labels, const float* weight, float* joint,
                         float* loss, float* dlogits, float* scratch, void* stream);

/* ---- Year ensemble (reference src/models/year.py:9-33): `years` (1..DTA_MAX_YEARS) spectral_networks, each on its own
 * input, run as the groups of ONE set of launches (a third of the launches of `years` separate dta_net_* calls);
 * the returned scores are the mean over the years of each year's last-head scores (year.py:30,33).
 * The reference skips a year whose whole batch tensor sums to zero (year.py:27): the caller passes only the years it
 * keeps (their params / inputs / grads, in any order), so a skipped year's BatchNorm state and gradients stay untouched.
 *  d      : kind DTA_NET_SPECTRAL; heads_mask is ignored (only the last head is evaluated)
 *  nets   : `years` entries;  x : `years` device pointers, each float32 NCHW [batch][bands][height][width]
 *  mean_scores : float32 [batch][classes] */
#define DTA_MAX_YEARS 16
size_t dta_ensemble_workspace_bytes(const dta_net_desc* d, int years);
int dta_ensemble_forward(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                         void* workspace, float* mean_scores, void* stream);
/* The reference's missing-year test on the device (year.py:27 `x.sum() == 0`): flags[y] = 1 when year y's tensor has a
 * non-zero element (NaN counts), else 0.  For the non-negative crops the reference's loader produces (min-max scaled to
 * [0, 1], missing years zero-filled, src/data.py:295-296) that is the same decision -- a sum of non-negative floats is
 * zero exactly when all of them are; tensors with negative entries that cancel to an exact zero sum would be skipped by
 * the reference and are kept here.  x: HOST array of `years` device pointers (16-byte aligned), n_per_year floats each.
 * clear_next (device, float[years], may be NULL): a second flag bank this call zeroes, so that a caller alternating two
 * banks never needs a clearing launch; with NULL the call clears `flags` itself first (one more tiny launch). */
int dta_year_flags(const float* const* x, int years, size_t n_per_year, float* flags, float* clear_next, void* stream);
/* dta_ensemble_forward over ALL years with the skip decided on the device: gate (device, float[years], e.g. from
 * dta_year_flags) <= 0 leaves that year out of the mean and leaves its BatchNorm running statistics / counter untouched,
 * exactly as a year the reference skips -- no host round trip.  (The skipped year's launches still run; their results are
 * never used.)  kept (device, float[2], may be NULL) receives {years kept, 1 / years kept}: the second word is the
 * gradient scale dta_weighted_ce_scaled_dev takes.  No year kept: mean_scores are NaN (the reference raises there). */
int dta_ensemble_forward_gated(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                               const float* gate, void* workspace, float* mean_scores, float* kept, void* stream);
/* Forward of the year ensemble AND the level's loss in one call (reference multi_stage.py:283-285: `y_hat = self.models[..]
 * .forward(images); loss = F.cross_entropy(y_hat, y, weight=self.loss_weight[..])`): dta_ensemble_forward(_gated) with the mean
 * over the kept years formed inside the loss launch instead of a launch of its own -- same values, bit for bit.
 *  gate    : float[years] year flags (dta_year_flags) or NULL = every year is kept
 *  mean_scores [batch][classes] (may be NULL), kept {kept years, 1 / kept} (may be NULL): outputs as in the gated forward
 *  dscore  : d(loss)/d(ONE year's scores) = d(loss)/d(mean) / kept years -- what dta_ensemble_backward* take; may be NULL
 *  labels / weight / loss / scratch : as for dta_net_loss.  Nothing kept: NaN loss, exact-zero dscore. */
int dta_ensemble_forward_loss(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                              const float* gate, void* workspace, const long long* labels, const float* weight,
                              float* mean_scores, float* kept, float* loss, float* dscore, float* scratch, void* stream);
/* Backward of the above.  dscore: d(loss)/d(one year's scores) = d(loss)/d(mean_scores) / years, float32
 * [batch][classes], shared by all years.  grads: `years` entries, every non-null buffer ZERO-FILLED on entry;
 * classifier1/2 gradients are not produced (those heads never reach the loss). */
int dta_ensemble_backward(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                          const float* dscore, const dta_subnet_grads* grads, void* stream);
/* Same with the `phases` split of dta_net_backward (bit 0: everything but the years' first-conv weight gradients,
 * bit 1: those), so that a data-parallel caller can overlap the first gradient exchange with them. */
int dta_ensemble_backward_phased(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                                 const float* dscore, const dta_subnet_grads* grads, int phases, void* stream);
/* Backward of dta_ensemble_forward_gated: gate (device, float[years], the flags the forward was gated by; NULL = all on)
 * <= 0 makes that year's gradients EXACT ZEROS (its score gradient is taken as zero whatever dscore holds; its launches
 * still run), as a year the reference skips has grad None (year.py:27-28).  A data-parallel rank whose batch lacks a year
 * that another rank kept thus contributes zeros to that year's gradient sum. */
int dta_ensemble_backward_gated(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                                const float* dscore, const dta_subnet_grads* grads, const float* gate, int phases,
                                void* stream);
/* dta_ensemble_backward_gated (phases = 3) for a data-parallel caller whose gradients live in a peer exchange object with a
 * head / tail split (dta_xchg_set_split; declared below): the head -- every gradient of every year except the years'
 * first-conv weights, plus whatever the caller keeps in it (its year flags) -- is complete before the years' grouped
 * first-conv weight-gradient launch, whose spare workgroups sum this rank's shard of the head over the ranks through
 * peer memory while the matrix cores work (as dta_net_backward_xchg does for one network; reference multi_stage.py:277-288
 * under train.py:89-98's DDP).  The dta_xchg_allreduce that follows finds the head summed.  Plans without a combined
 * kernel (fp32 mode, launches that fill every CU): exactly dta_ensemble_backward_gated, the exchange sums both segments. */
struct dta_xchg;
int dta_ensemble_backward_xchg(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                               const float* dscore, const dta_subnet_grads* grads, const float* gate, struct dta_xchg* xchg,
                               void* stream);

/* ---- Crop preprocessing on the device: replaces load_image / preprocess_image (src/utils.py:36-79: drop the first and
 * last `clip` bands when there are more than 3, float32, per-pixel min-max over the bands as
 * sklearn.preprocessing.minmax_scale(axis=1) computes it, NEAREST resize to size x size) plus the training flips
 * (src/augmentation.py:13-14: horizontal then vertical, both p=1) for a whole batch of ragged crops in one launch.
 * Output is bit-identical to the reference's CPU path.
 *  raw     : all crops of the batch back to back (device memory), element type `dtype`
 *  offsets : [batch] int64 element offset of each crop inside raw;  heights/widths : [batch] int32 (device)
 *            a crop with height or width 0 is a missing year: its output is all zeros (src/data.py:295-296)
 *  layout  : DTA_CROP_CHW = band-first [bands_raw][h][w] (what rasterio's read() / np.load return),
 *            DTA_CROP_HWC = pixel-interleaved [h][w][bands_raw] (how the crops lie on disk: TIFF PlanarConfiguration 1)
 *  out     : float32 [batch][dta_preprocess_out_bands(bands_raw, clip)][size][size] */
enum { DTA_CROP_F32 = 0, DTA_CROP_I16 = 1, DTA_CROP_U8 = 2 };
enum { DTA_CROP_CHW = 0, DTA_CROP_HWC = 1 };
typedef struct {
  int batch, bands_raw, clip, size, flip, layout, dtype;
} dta_crop_desc;
int dta_preprocess_out_bands(int bands_raw, int clip);
int dta_preprocess_crops(const dta_crop_desc* d, const void* raw, const long long* offsets, const int* heights,
                         const int* widths, float* out, void* stream);
/* Same preprocessing, written straight as the bf16 conv tiles dta_net_forward_tiles takes (the float32 batch never
 * exists): tiles bf16 [batch][ceil(out_bands / 16)][size * size][16] = the float32 results above rounded to bf16
 * (nearest even), bands past out_bands zero. */
int dta_preprocess_crops_tiles(const dta_crop_desc* d, const void* raw, const long long* offsets, const int* heights,
                               const int* widths, void* tiles, void* stream);

/* Replaces F.cross_entropy(logits, y, weight=w) forward+backward (src/main.py:78, multi_stage.py:285).
 * weight may be null (= ones, metadata.py:61).  scratch: batch+1 floats.  dlogits may be null.
 * Labels: -100 is ignored as torch's default ignore_index; any other label outside [0, classes) is a caller bug
 * (torch raises or device-asserts) and makes the loss and that row of dlogits NaN instead of being skipped silently. */
int dta_weighted_ce(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                    float* loss, float* dlogits, float* scratch, void* stream);

/* The same loss in ONE launch (the last block to arrive finalises it) with a factor on the gradient only: dlogits =
 * grad_scale * d(loss)/d(logits).  The year ensemble's step uses it with grad_scale = 1 / kept years, the derivative of
 * the mean over the kept years' scores (src/models/year.py:33), so no elementwise pass follows the loss.
 * scratch: batch + 2 floats whose LAST 32-bit word is zero on entry (block counter; left zero). */
int dta_weighted_ce_scaled(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                           float grad_scale, float* loss, float* dlogits, float* scratch, void* stream);
/* Same with the factor read from the device (grad_scale_dev[0]): 1 / kept years as dta_ensemble_forward_gated left it. */
int dta_weighted_ce_scaled_dev(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                               const float* grad_scale_dev, float* loss, float* dlogits, float* scratch, void* stream);

/* Inference epilogue: replaces F.softmax(pred, dim=1) (src/models/multi_stage.py:302,315; src/main.py:190) and the
 * top-1/top-2 label+score extraction of src/main.py:192-205.  probs [batch][classes] may be null.
 * top_idx [batch][2] int64, top_score [batch][2] float32. */
int dta_softmax_top2(const float* logits, int batch, int classes, float* probs, long long* top_idx, float* top_score,
                     void* stream);

/* Replaces torch.optim.Adam(params, lr).step() (src/main.py:136) over one flat fp32 buffer plus the float64
 * alpha (alpha_* may be null).  step = 1-based step count; grads are multiplied by grad_scale first. */
int dta_adam_step(float* p, const float* g, float* m, float* v, size_t n, double* alpha_p, const double* alpha_g,
                  double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2, float eps,
                  float grad_scale, void* stream);

/* optimizer.step() followed by optimizer.zero_grad() (the Lightning loop of src/main.py does both every batch) in one
 * pass: same update as dta_adam_step, then g and alpha_g are cleared, which is the state dta_net_backward expects. */
int dta_adam_step_zero_grad(float* p, float* g, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                            double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2, float eps,
                            float grad_scale, void* stream);

/* Data-parallel form: alpha's gradient is read from alpha_g_f32 (the exchange slot dta_net_backward_dp filled and the
 * all-reduce summed) when that is non-NULL, from alpha_g otherwise; zero_grad != 0 clears g and alpha_g after use. */
int dta_adam_step_dp(float* p, float* g, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                     const float* alpha_g_f32, double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2,
                     float eps, float grad_scale, int zero_grad, void* stream);

/* optimizer.step() + zero_grad() gated ON THE DEVICE (year ensembles under data parallelism, where whether a year is
 * stepped -- "some rank kept it", src/models/year.py:27 -- is only known on the device after the gradient exchange):
 * active[0] > 0: Adam step number dev_step[0] + 1 (dev_step[0] = steps this parameter group has taken so far, a device
 * counter); otherwise only the gradient buffer is cleared -- always: a group that is not stepped has no gradient (torch
 * leaves grad None), whatever zero_grad says -- and nothing else happens (no moment decay, as torch's Adam passes over
 * parameters whose grad is None).  dev_step_next (may be NULL; must not alias dev_step): receives the count after
 * this step, dev_step[0] + (active ? 1 : 0) -- callers ping-pong two counter words, so the count advances without any
 * launch of their own; pass it in ONE of the launches that share a counter.  No float64 alpha here. */
int dta_adam_step_gated(float* p, float* g, float* m, float* v, size_t n, const float* active, const int* dev_step,
                        int* dev_step_next, float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                        void* stream);
/* Several parameter groups in ONE launch: the per-year optimizers of a year ensemble (reference multi_stage.py:258-275: one
 * Adam per level over all years' parameters; years the step skipped are passed over) instead of one launch per year and
 * segment.  Each segment is stepped exactly as dta_adam_step (active == NULL: host step count `step` >= 1) or as
 * dta_adam_step_gated (active != NULL: device gate + device step counter; `step` ignored) would step it; zero_grad and the
 * hyper-parameters are common.  1 <= nseg <= DTA_ADAM_MAX_SEGMENTS. */
#define DTA_ADAM_MAX_SEGMENTS 16
typedef struct dta_adam_segment {
  float* p; float* g; float* m; float* v; size_t n;
  const float* active; const int* dev_step; int* dev_step_next;   /* gated form, or all NULL */
  int step;                                                        /* host step count (ungated form) */
  float lr;                                                        /* > 0: this segment's learning rate instead of the call's (one Adam per
                                                                      level, reference multi_stage.py:258-262); 0: the call's */
} dta_adam_segment;
int dta_adam_step_multi(int nseg, const dta_adam_segment* segs, float lr, float beta1, float beta2, float eps, float grad_scale,
                        int zero_grad, void* stream);

/* ---- Multi-stage step (reference src/models/multi_stage.py:41-66: one learned_ensemble per level of the hierarchy, each with
 * its own class count; :277-288: per level weighted cross-entropy of the mean over the years; train.py:75-100 trains all
 * levels on every batch).  The levels x kept-years spectral networks run as the groups of ONE set of launches -- one forward
 * chain, ONE loss launch over the levels, one backward chain -- instead of one chain per level.  Every level's batch has
 * the same size (d->batch) and crop shape; d->classes is ignored (each level carries its own).
 *  lv[l].first / count : the level's networks are nets[first .. first + count) (likewise x, grads, gate); levels in order,
 *                        adjacent, the total at most DTA_MAX_YEARS networks
 *  gate (device, float[total], may be NULL): as for dta_ensemble_forward_gated -- a network whose flag is <= 0 is left out
 *                        of its level's mean, keeps its BatchNorm statistics and gets exact-zero gradients */
#define DTA_MAX_LEVELS 8
typedef struct dta_level {
  int classes, first, count;
  const long long* labels;   /* device int64 [batch] */
  const float* weight;       /* device float32 [classes], NULL = ones (reference multi_stage.py:67-79 loss_weight_{level}) */
  float* mean_scores;        /* out, device [batch][classes]: mean over the level's kept years (may be NULL) */
  float* kept;               /* out, device {kept years, 1 / kept years} (may be NULL) */
  float* loss;               /* out, device scalar */
  float* dscore;             /* out, device [batch][classes] = d(loss) / d(ONE kept year's scores); NULL: no gradient */
  float* scratch;            /* device, batch + 2 floats; the last word must be zero on entry and is left zero */
} dta_level;
size_t dta_multistage_workspace_bytes(const dta_net_desc* d, int levels, const dta_level* lv);
int dta_multistage_forward_loss(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                                const float* const* x, const float* gate, void* workspace, void* stream);
/* The forward alone (reference MultiStage.predict_step, multi_stage.py:306-318: every level's model on the SAME crops -- the
 * x entries of the levels may point at the same tensors; and validation): lv[l].mean_scores receives each level's mean over its
 * kept years, lv[l].kept (may be NULL) {kept, 1 / kept}; labels / loss / dscore / scratch are not read. */
int dta_multistage_forward(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                           const float* const* x, const float* gate, void* workspace, void* stream);
/* ... and with the inference epilogue of every level in the SAME call: each level's mean over its kept years, softmax over
 * its classes and the top-2 labels / scores (reference multi_stage.py:306-318 + main.py:190-205) in ONE launch behind the
 * forward.  probs (may be NULL, entries may be NULL) / top_idx / top_score: HOST arrays of `levels` device pointers
 * ([batch][classes] float32, [batch][2] int64, [batch][2] float32); lv[l].mean_scores (may be NULL) receives the scores. */
int dta_multistage_predict(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                           const float* const* x, const float* gate, void* workspace, float* const* probs,
                           long long* const* top_idx, float* const* top_score, void* stream);
int dta_multistage_backward(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                            void* workspace, const dta_subnet_grads* grads, const float* gate, void* stream);

/* ---- Peer gradient exchange: data-parallel training with one process per GPU of ONE node (reference train.py:89-98:
 * Lightning DDP all-reduces every parameter's gradient between loss.backward() and optimizer.step()).  Here the sum over
 * ranks and the Adam step are ONE launch on the caller's compute stream: every rank pulls its shard of all ranks'
 * gradient buffers through IPC-mapped peer memory (xGMI), publishes the sums in its staging area, and every rank then
 * pulls all summed shards and steps its full (replicated) optimizer state -- no collective library, no side stream, no
 * stream-event hops; sums are taken in rank order, so replicas stay bit-identical.  Waits inside the kernel are bounded:
 * a missing peer sets a status word (dta_xchg_status) instead of hanging the GPU.
 * EXCEPTION to the "allocates nothing" rule above: the exchange object owns device memory that peers map -- the flat
 * float32 gradient buffer (dta_xchg_grad_buffer; the backward entry points write the step's gradients there) and an
 * uncached signal + staging allocation.  Create it once per trainer, destroy it after a barrier over the ranks.
 *  handles: DTA_XCHG_HANDLE_BYTES per rank (dta_xchg_export writes this rank's; dta_xchg_connect takes all ranks' in
 *  rank order, gathered by the caller over any side channel, e.g. torch.distributed.all_gather_object). */
typedef struct dta_xchg dta_xchg;
#define DTA_XCHG_HANDLE_BYTES 128
#define DTA_XCHG_MAX_WORLD 8
int dta_xchg_create(int rank, int world, size_t n_floats, dta_xchg** out);
float* dta_xchg_grad_buffer(dta_xchg* x);          /* device pointer, dta_xchg_grad_capacity(x) floats, zero-filled */
size_t dta_xchg_grad_capacity(dta_xchg* x);        /* n_floats rounded up to a multiple of 4 */
int dta_xchg_export(dta_xchg* x, void* handles);
int dta_xchg_connect(dta_xchg* x, const void* all_handles);
/* Bound of every in-kernel wait (default 1800 s, a collective library's watchdog scale: rank skew of seconds is routine).
 * A wait that expires is FATAL for the exchange: the launch writes the status word and returns without applying anything,
 * and every later launch of this exchange returns at once (sticky), so a replica cannot train on past a failed exchange. */
void dta_xchg_set_timeout(dta_xchg* x, double seconds);
/* Exchange the buffer as two segments, head [0, split_floats) and tail [split_floats, n): the head can then be summed
 * ahead of the exchange launch by dta_net_backward_xchg.  Only before the first step; every rank sets the same split
 * (a multiple of 4 floats; 0 = one segment, the default). */
int dta_xchg_set_split(dta_xchg* x, size_t split_floats);
/* Grid bound of the exchange launch (default 256, one workgroup per CU); only before the first step.  Ranks that share
 * one GPU (tests) must keep world x workgroups co-resident: every rank's launch waits in-kernel for the others. */
void dta_xchg_set_max_workgroups(dta_xchg* x, int workgroups);
/* Overlapped form without a combined kernel: the reduce-scatter of the HEAD segment [0, split) of the exchange's NEXT launch
 * as a launch of its own (16 workgroups) on `stream` -- e.g. a side stream beside the caller's last gradient kernel, or the
 * compute stream itself.  The head must be complete on `stream` (kernel boundary / event); the dta_xchg_allreduce /
 * dta_xchg_adam_step that follows (ordered after this launch by the caller) then sums only the tail and gathers both.
 * One rank or no split: a no-op.  alpha_g / alpha_slot as below (the slot must lie inside the head).  This is the device
 * code dta_net_backward_xchg / dta_ensemble_backward_xchg run in the spare workgroups of the first conv's weight gradient.
 * Replaces: DDP's first gradient bucket being all-reduced while the backward still runs (reference train.py:89-98). */
int dta_xchg_reduce_head(dta_xchg* x, const double* alpha_g, long long alpha_slot, void* stream);
/* g := sum over ranks of g (all ranks end with the same bits).  alpha_g (may be NULL): this rank's float64 d(alpha); the
 * launch first stores it, rounded to float32, in slot alpha_slot of the gradient buffer, so that it takes part in the sum. */
int dta_xchg_allreduce(dta_xchg* x, const double* alpha_g, long long alpha_slot, void* stream);
/* Sum over ranks + dta_adam_step_dp's update in one launch.  p / m / v: this rank's flat buffers of
 * dta_xchg_grad_capacity(x) floats; alpha_slot: index (in floats) of alpha's exchange slot inside the gradient buffer, or
 * -1 with alpha_p NULL; alpha_g: this rank's float64 d(alpha) as the backward left it -- the launch rounds it to float32
 * into the slot before the sum (no float atomics anywhere: replicas and reruns are bit-identical) and afterwards stores
 * the summed gradient there (zero_grad = 0) or clears it.  zero_grad != 0: the gradient buffer is cleared for the next backward; else it holds the sum. */
int dta_xchg_adam_step(dta_xchg* x, float* p, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                       long long alpha_slot, double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2,
                       float eps, float grad_scale, int zero_grad, void* stream);
/* Self-test aid (peer_probe.py): fill the gradient buffer with a known pattern of (rank, step) BY A KERNEL on `stream`, so
 * that an exchange enqueued right behind it tests exactly what a train step relies on -- gradients written by the kernel
 * before the exchange on the same stream are visible to the peers' system-scope loads (kernel-boundary write-back). */
int dta_xchg_selftest_fill(dta_xchg* x, int step, void* stream);
/* ... and its counterpart behind the exchange: compares the buffer with the sum over ranks of step `step`'s patterns ON THE
 * DEVICE and adds the number of differing elements to a counter (dta_xchg_selftest_mismatches reads it after a
 * synchronisation).  fill -> all-reduce -> verify triples can so be enqueued back to back, with no host synchronisation
 * between steps -- what a training loop does, and what a lazily written-back buffer would not survive. */
int dta_xchg_selftest_verify(dta_xchg* x, int step, void* stream);
int dta_xchg_selftest_mismatches(dta_xchg* x);
/* 0 = every step so far completed; otherwise (phase << 8 | rank waited for) of the first timed-out wait (host-side read of
 * a pinned word, no synchronisation: a launch that timed out has written it by the time it ends; trainers poll it at the
 * start of every step). */
int dta_xchg_status(dta_xchg* x);
/* Development aid: how long workgroup 0 of the LAST exchange launch waited for the ranks to arrive and how long the
 * exchange proper took afterwards, in microseconds (pinned host words: synchronise the stream first). */
int dta_xchg_last_timing(dta_xchg* x, float* wait_us, float* exchange_us);
/* Orderly teardown in two collective steps: every rank UNMAPS its peers' buffers (dta_xchg_disconnect), the ranks meet at a
 * barrier, and only then does anybody free what the others had mapped (dta_xchg_destroy, which also disconnects when that
 * was not done).  Freeing while a peer still holds a mapping is safe for the peer, but a buffer re-created at the same address
 * right away (probe -> trainer, one trainer after another) then races with the peer's close of the old mapping: seen as
 * hipIpcGetMemHandle / stale-mapping failures with eight processes on one test GPU. */
int dta_xchg_disconnect(dta_xchg* x);
int dta_xchg_destroy(dta_xchg* x);

/* ---- stand-alone building blocks (same kernels as the network-level path) ------------------------------------ */

typedef struct dta_conv_module_desc {
  int batch, in_channels, filters, height, width;   /* filters in {32, 64, 128} */
  int pool;        /* 1: MaxPool2d(2) after ReLU (conv_module.forward(x, pool=True)) */
  int training, dtype;
  float bn_momentum, bn_eps;
} dta_conv_module_desc;

/* Replaces conv_module.forward (Hang2020.py:24-31): x NCHW float32 -> out NCHW float32 [batch][filters][H'][W'].
 * Workspace (dta_conv_module_workspace_bytes) is kept for the backward call. */
size_t dta_conv_module_workspace_bytes(const dta_conv_module_desc* d);
int dta_conv_module_forward(const dta_conv_module_desc* d, const float* conv_w, const float* conv_b, const float* bn_w,
                            const float* bn_b, float* bn_rm, float* bn_rv, long long* bn_nbt, const float* x,
                            void* workspace, float* out, void* stream);
/* dout: NCHW gradient of `out`.  dx_nhwc (optional, needs in_channels in {32,64,128}): gradient wrt x as
 * [batch][H*W][in_channels].  g_* arrive zero-filled. */
int dta_conv_module_backward(const dta_conv_module_desc* d, const float* conv_w, const float* bn_w, void* workspace,
                             const float* dout, float* dx_nhwc, float* g_conv_w, float* g_conv_b, float* g_bn_w,
                             float* g_bn_b, void* stream);

typedef struct dta_attention_desc {
  int batch, filters, height, width;   /* filters in {32, 64, 128} */
  int kind;                            /* 0 spectral_attention, 1 spatial_attention */
} dta_attention_desc;

/* Replaces spectral_attention.forward (Hang2020.py:149-168) / spatial_attention.forward (:105-124).
 * params: the module's tensors in the order of dta_subnet_params.att.  x_nhwc: [batch][H*W][filters] float32.
 * out_nchw: gated map [batch][filters][H][W]; feat: pooled features [batch][F]. */
size_t dta_attention_workspace_bytes(const dta_attention_desc* d);
int dta_attention_forward(const dta_attention_desc* d, const float* const params[6], const float* x_nhwc, void* workspace,
                          float* out_nchw, float* feat, void* stream);
/* dout_nchw / dfeat: gradients of the two outputs (either may be null).  dx_nhwc: [batch][H*W][filters].
 * grads[6] arrive zero-filled (null entries are skipped). */
int dta_attention_backward(const dta_attention_desc* d, const float* const params[6], const float* x_nhwc, void* workspace,
                           const float* dout_nchw, const float* dfeat, float* dx_nhwc, float* const grads[6], void* stream);

/* Replaces Classifier.forward / nn.Linear (Hang2020.py:63-66) and its backward.  gw / gb arrive zero-filled. */
int dta_linear_forward(const float* x, const float* w, const float* b, int batch, int in_features, int out_features,
                       float* out, void* stream);
int dta_linear_backward(const float* x, const float* w, const float* dout, int batch, int in_features, int out_features,
                        float* dx, float* gw, float* gb, void* stream);

/* Measurement aid (host-side state only): record a HIP-event pair around every launch of a kernel site, on the
 * stream the kernel is launched on.  site = DTA_SITE_* + layer (0..2); each call adds a site (up to four are timed at
 * once), -1 stops and forgets them all.  dta_profile_collect_site waits for the recorded events of one site, writes up
 * to `max` durations in milliseconds (HOST pointer) and returns the count; dta_profile_collect = the first site. */
enum { DTA_SITE_CONV_FWD = 0, DTA_SITE_CONV_WGRAD = 3, DTA_SITE_CONV_DGRAD = 6, DTA_SITE_STAGE_FWD = 9,
       DTA_SITE_STAGE_BWD = 12,
       DTA_SITE_GEMM = 15 /* +0 classifier heads forward, +1 head input gradients, +2 parameter-gradient group */ };
int dta_profile_enable(int site);
/* Time only every stride-th launch of an enabled site (default 1): an event pair costs ~5 us of stream time per launch,
 * which a throughput measurement running beside the timing should not pay on every step. */
int dta_profile_set_stride(int stride);
int dta_profile_collect(float* ms, int max);
int dta_profile_collect_site(int site, float* ms, int max);
/* Development aid.  The PRODUCT library (libdta_hip.so) reads nothing from the environment: dta_dev_switches_enabled()
 * returns 0 and dta_dev_reload_switches() fails.  The developer library (libdta_hip_dev.so: the same sources compiled with
 * -DDTA_DEV_SWITCHES) reads its switches (DTA_NO_FUSED_INPUT, DTA_NO_TAIL_MERGE, DTA_BN_INKERNEL, DTA_NO_LEAN, DTA_FANIN,
 * ...: same-box A/B runs of alternative launch plans) once at load time, and again on dta_dev_reload_switches(). */
int dta_dev_switches_enabled(void);
int dta_dev_reload_switches(void);

/* ---- site-metadata head of the fusion model (reference src/models/metadata.py:9-44) --------------------------------
 * meta = ReLU(Linear(Dropout(BatchNorm1d(Embedding(site)))))  (16 wide), out = ReLU(Linear(cat[meta, hsi_scores])).
 * Replaces the reference's torch modules `metadata` (:9-24) and the `fc1` fusion layer of `metadata_sensor_fusion` (:26-44)
 * for the train / validation step of MetadataModel (:52-83); the HSI scores come from dta_net_forward.
 * All tensors float32, row-major; site: int64 indices in [0, sites).  drop: [batch][16] dropout factors (0 or 1/(1-p)),
 * or NULL (no dropout / eval) -- drawn by the caller (torch's generator), applied here.  training != 0: batch statistics
 * (and the running statistics are updated with `momentum`), else running statistics.  The workspace
 * (dta_meta_head_workspace_bytes) carries the forward's state to the backward. */
typedef struct dta_meta_params {
  const float* emb;            /* [sites][16]   metadata_model.embedding.weight */
  const float* bn_w; const float* bn_b;   /* [16]  metadata_model.batch_norm.weight / bias */
  float* bn_rm; float* bn_rv; long long* bn_nbt;   /* running_mean / running_var [16], num_batches_tracked (may be NULL) */
  const float* mlp_w; const float* mlp_b;  /* [classes][16], [classes]   metadata_model.mlp */
  const float* fc_w; const float* fc_b;    /* [classes][2*classes], [classes]   fc1 (columns: [site scores | hsi scores]) */
} dta_meta_params;
typedef struct dta_meta_grads {   /* same shapes; zero-filled on entry (bias gradients are accumulated); NULL = not wanted */
  float* emb; float* bn_w; float* bn_b; float* mlp_w; float* mlp_b; float* fc_w; float* fc_b;
} dta_meta_grads;
size_t dta_meta_head_workspace_bytes(int batch, int classes, int sites);
/* out[batch][classes] = the fused scores (after the last ReLU) */
int dta_meta_head_forward(int batch, int classes, int sites, int training, float momentum, float eps, const dta_meta_params* p,
                          const long long* site, const float* scores, const float* drop, void* workspace, float* out,
                          void* stream);
/* unweighted cross-entropy of the fused scores (MetadataModel.training_step, metadata.py:52-63) in one launch: loss (0-d) and
 * dout[batch][classes] = d(loss)/d(the last ReLU's INPUT), i.e. the ReLU's backward is applied; scratch: batch + 2 floats,
 * word batch + 1 zero on entry (left zero).  dout may be NULL (validation). */
int dta_meta_head_loss(int batch, int classes, const float* out, const long long* labels, float* loss, float* dout, float* scratch,
                       void* stream);
/* dout = what dta_meta_head_loss wrote; writes the parameter gradients and dscores[batch][classes] = d(loss)/d(hsi scores) */
int dta_meta_head_backward(int batch, int classes, int sites, int training, const dta_meta_params* p, const long long* site,
                           const float* drop, void* workspace, const float* out, const float* dout, const dta_meta_grads* grads,
                           float* dscores, void* stream);

#ifdef __cplusplus
}
#endif
#endif
