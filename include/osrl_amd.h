/*
 * osrl_amd.h -- C ABI of libosrl_amd.so: the MI355X (gfx950) kernels behind the
 * OSRL Trainer.train_one_step() hot path (BC / CPQ / BCQ-Lag / CDT).
 *
 * The reference (liuzuxin/OSRL) is pure Python on PyTorch aten and has NO FFI seam
 * (SURVEY.md section 8b); the seam this library fills is "what aten did for the
 * reference": each entry point below names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (row-major) unless stated otherwise;
 *    nn.Linear weights are [out,in] exactly as in the reference state_dict;
 *  - caller allocates every output and workspace; nothing is allocated or freed here;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), makes no
 *    host synchronisation and is hipGraph-capturable;
 *  - return value: 0 on success, otherwise a hipError_t (or -1 for a bad argument);
 *    nothing throws across the ABI.  State: none that persists between calls, with ONE opt-in exception --
 *    osrl_args_begin / osrl_args_end bracket a THREAD-LOCAL argument arena (csrc/argmem.h, a static thread_local in
 *    csrc/optim.hip): between the two calls the fused-MLP / optimizer / prologue launches of the calling thread record
 *    or look up their descriptors there; outside such a bracket (the default) every call is stateless and re-entrant
 *    across streams, devices and threads.  A few entry points own caller-provided device scratch that must be zero
 *    before the first call and is re-armed by the call itself (arrival counters: osrl_mlp_backward_dw_tiles_adam,
 *    osrl_mlp_backward_dz_seed, osrl_mlp_regress_step, the *_ws loss / quantile calls).
 */
#ifndef OSRL_AMD_H
#define OSRL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSRL_MAX_LAYERS 4 /* Linear layers per MLP */
#define OSRL_MAX_NETS 8   /* ensemble members sharing one input tile */
#define OSRL_MAX_WIDTH 448 /* widest hidden layer supported by the fused kernels */

enum { OSRL_ACT_ID = 0, OSRL_ACT_RELU = 1, OSRL_ACT_TANH = 2 };
/* row r of the virtual input maps to source row: r | r % div | r / div
 * (MOD = torch.tile(obs[None],(N,1,1)) cpq.py:170-174 ; DIV = repeat_interleave bcql.py:138) */
enum { OSRL_MAP_ID = 0, OSRL_MAP_MOD = 1, OSRL_MAP_DIV = 2 };

/* A stack of n_layers Linear(+activation) layers for n_nets ensemble members:
 * osrl/common/net.py:12-30 `mlp()`; ensembles net.py:208-287. */
typedef struct {
  int32_t n_layers, n_nets;
  int32_t dims[OSRL_MAX_LAYERS + 1]; /* in, h1, ..., out */
  int32_t acts[OSRL_MAX_LAYERS];     /* activation after each layer */
  float out_scale;                   /* net output = out_scale * act(z)  (act_limit of net.py:62,85,339) */
  int32_t tile_rows;                 /* tuning hint: rows per workgroup tile (0 = auto, else 16 / 32 / 64);
                                      * 80 = forward-only launches take the one-workgroup-per-CU kernel when the
                                      * network's shape allows it */
  int32_t wg_cap;                    /* forward launches: at most this many workgroups in the grid (0 = one per
                                      * tile); a capped launch walks its tiles, leaving CU slots to concurrent
                                      * latency-critical launches */
  /* PACKED weights (osrl_pack_weights): Wf feeds forward, Wb (packed W^T) feeds backward-dz.
   * Wf[e][l]: PF[k/4][n][k%4], n < round16(out), k < round16(in), zero padded.
   * Wb[e][l]: PB[o/4][i][o%4], i < round16(in)+16, o < round16(out), zero padded (NULL if unused). */
  const float* Wf[OSRL_MAX_NETS][OSRL_MAX_LAYERS];
  const float* Wb[OSRL_MAX_NETS][OSRL_MAX_LAYERS];
  const float* b[OSRL_MAX_NETS][OSRL_MAX_LAYERS]; /* [dims[l+1]] canonical bias */
} osrl_mlp_t;

/* One canonical nn.Linear weight [out,in] (row-major, at src_flat + src_off) and where its packed
 * copies go (float offsets into pf / pb; negative = skip). */
typedef struct {
  int64_t src_off, f_off, b_off;
  int32_t out, in;
} osrl_pack_entry_t;

/* Virtual input matrix [rows, d0+d1] = cat(src0[map0(r)], src1[map1(r)]) -- replaces the
 * torch.cat / tile / repeat_interleave copies of net.py:232,273,320,337 and cpq.py:170. */
typedef struct {
  int32_t rows;
  int32_t d0, map0, div0;
  int32_t d1, map1, div1;
  const float* src0;
  const float* src1; /* may be NULL when d1 == 0 */
  /* Optional (both NULL = none): a launch over a row SET chosen on the device -- row r of the virtual input is row
   * row_list[r] of the matrix described above, and only the first min(rows, *n_rows_dev) rows exist (rows = the list's
   * capacity: it sizes the grid; workgroups past the count leave at once).  For cpq.py:183-184: of the N*B sampled actions
   * only those whose KL reaches the batch quantile enter qc_ood -- the target cost critics run on those rows alone.
   * Honoured by the 80-row inference forward of <= 256-wide nets (osrl_mlp_forward); every other entry point returns -3. */
  const int32_t* row_list;
  const int32_t* n_rows_dev;
  /* Optional (0 = off): SHARED SRC0 ROWS of an inference launch whose rows are n * div0 + b (map0 = OSRL_MAP_MOD: cpq.py:164-176,
   * N sampled actions per observation).  share0 = 1: tiles are [copies] x [16 src0 rows] and the part of layer 0 that lies in
   * the first 16 * share_k16 input columns (all of them columns of src0; share_k16 >= 1, at least one 16-column k-step left)
   * is computed once per src0 row of a tile instead of once per launch row.  Same products, another order of a row's sum (a few
   * ulp from the plain launch): for no_grad launches.  Needs div0 % 16 == 0 and (rows / div0) a multiple of the tile's row
   * blocks (5; 4 on the 64-row form).  A HINT: taken by the 80-row inference forward of <= 256-wide (4-wave) and 400-wide
   * (8-wave) nets (osrl_mlp_forward[_tail]); every other launch computes the same function on its plain tiles. */
  int32_t share0, share_k16;
} osrl_rows_t;

/* Activations written by forward / read by backward.  h[e][l] = post-activation output of
 * layer l of net e, [rows, dims[l+1]].  h[e][n_layers-1] (the net output) is REQUIRED in
 * forward; every other pointer may be NULL (= not saved). */
typedef struct {
  float* x; /* [rows, dims[0]] the concatenated input (needed by dW of layer 0) */
  float* h[OSRL_MAX_NETS][OSRL_MAX_LAYERS];
} osrl_mlp_acts_t;

typedef struct {
  const float* dy[OSRL_MAX_NETS]; /* [rows, dims[L]] grad wrt the net's (post-activation) output */
  float* dz[OSRL_MAX_NETS][OSRL_MAX_LAYERS]; /* out (NULL = skip): grad wrt pre-activation of layer l */
  float* dx[OSRL_MAX_NETS]; /* out (NULL = skip): [rows, dx_cols] = dX[:, dx_col0 : dx_col0+dx_cols] */
  int32_t dx_col0, dx_cols;
} osrl_mlp_grads_t;

/* Optional row-local tail of a fused-MLP launch, applied by the workgroup that owns the rows while its last tile is
 * still in LDS (one launch and one global round trip less on the step's latency chain).  HOST struct.
 *   OSRL_TAIL_VAE_LATENT      forward of the VAE encoder (net 0, output [rows, 2L] = mean | log_std):
 *                             out[rows, L] = mean + exp(clamp(log_std, -4, 15)) * eps        == osrl_vae_latent
 *   OSRL_TAIL_VAE_LATENT_BWD  backward of the VAE decoder whose dX slice (dx[0], dx_cols == L) is dL/dz:
 *                             out[rows, 2L] = d(recon + beta KL)/d(mean | log_std)          == osrl_vae_latent_bwd
 *                             with head = the encoder output [rows, 2L]; beta, rows_global as in that call
 *   OSRL_TAIL_GAUSS           forward of a squashed-Gaussian actor trunk (net 0, output [rows, 2L] = mu | log_std, L =
 *                             action_dim; net.py:152-205): up to two action draws and the N pre-tanh OOD draws of
 *                             cpq.py:164-176 from the LDS-resident head tile --
 *                               out [rows, L]  = max_action * tanh(mu + exp(clamp(log_std, -20, 2)) * eps)    == osrl_gauss_head
 *                               out2 [rows, L] = the same with eps2 (NULL = none), tanh2 [rows, L] = its tanh(u) (optional)
 *                               out_ood [n_samples * rows, L], row j * rows + r = mu + sd * eps_ood[j, r, :]  == osrl_gauss_ood_sample
 *                             (any of the three may be absent: eps / eps2 / eps_ood NULL) */
/*   OSRL_TAIL_VAE_KL          forward of the VAE encoder (net 0, output [rows, 2L]): out[rows] = mean_k KL(N(mean_k, sd_k) || N(0, 1)),
 *                             the per-row statistic of cpq.py:178-182                                      == osrl_vae_kl_rows
 *                             (fused into the 80-row N*B-row kernel's output pass; a launch of its own behind any other) */
enum { OSRL_TAIL_NONE = 0, OSRL_TAIL_VAE_LATENT = 1, OSRL_TAIL_VAE_LATENT_BWD = 2, OSRL_TAIL_GAUSS = 3, OSRL_TAIL_VAE_KL = 4 };
typedef struct {
  int32_t kind;
  int32_t L;
  const float* eps;  /* [rows, L] */
  const float* head; /* OSRL_TAIL_VAE_LATENT_BWD only */
  float* out;
  float beta;          /* OSRL_TAIL_VAE_LATENT_BWD only */
  int32_t rows_global; /* OSRL_TAIL_VAE_LATENT_BWD only: the loss is a mean over this many rows (<= 0: rows) */
  float inv_rows_;     /* filled in by the library */
  /* OSRL_TAIL_GAUSS only */
  float max_action;
  const float* eps2;
  float* out2;
  float* tanh2;
  const float* eps_ood;
  float* out_ood;
  int32_t n_samples, pad_;
} osrl_mlp_tail_t;

/* One weight-gradient GEMM  dW[out,in] = dz^T a,  db[out] = sum_rows dz  (autograd of addmm). */
typedef struct {
  const float* dz; /* [rows, out] */
  const float* a;  /* [rows, in]  */
  int64_t w_off, b_off; /* float offsets of dW / db inside one gradient slab */
  int32_t out, in;
  int32_t ldz, lda;     /* row strides of dz / a in floats (0 = dense: out / in) */
} osrl_dw_entry_t;

/* Device-resident per-step scalars, advanced once per train step by osrl_step_tick(). */
typedef struct {
  int64_t step;     /* optimizer step count t (1-based after the first tick) */
  float bc1;        /* 1 - beta1^t */
  float bc2_sqrt;   /* sqrt(1 - beta2^t) */
  float lr_scale;   /* LambdaLR factor min(t/warmup,1) (cdt.py:327-330); 1 when warmup == 0 */
  uint32_t arrive_; /* workgroup arrival counter of osrl_step_begin (0 between launches) */
} osrl_step_state_t;

/* One dropout site (HOST struct, read at launch): probability, site id (unique per nn.Dropout call site and
 * layer), generator seed, and the device step state whose `step` decorrelates successive train steps. */
typedef struct {
  float p;
  uint32_t site;
  uint64_t seed;
  const osrl_step_state_t* st;
} osrl_dropout_t;

/* ---- fused MLP (mlp.hip): replaces nn.Sequential(Linear,act,...) forward + its autograd ---- */
/* lds_bytes / tile selection are internal; rows may be any value >= 1. */
int osrl_mlp_forward(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out, void* stream);
/* Two independent forward problems in one launch (e.g. the actor on next_obs and on obs, cpq.py:141,164): same
 * results as two osrl_mlp_forward calls; falls back to exactly that when the two tile shapes differ. */
int osrl_mlp_forward2(const osrl_mlp_t* net0, const osrl_rows_t* in0, const osrl_mlp_acts_t* out0,
                      const osrl_mlp_t* net1, const osrl_rows_t* in1, const osrl_mlp_acts_t* out1, void* stream);
int osrl_mlp_backward_dz(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                         const osrl_mlp_grads_t* g, void* stream);
/* The same launches followed by a row-local tail (osrl_mlp_tail_t; NULL or kind NONE = the plain call): results are
 * those of the plain call plus the named glue call (net.py:319-331 reparameterisation and its autograd, cpq.py:125-131),
 * fused into the launch when its last tile is LDS-resident, two launches otherwise. */
int osrl_mlp_forward_tail(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out,
                          const osrl_mlp_tail_t* tail, void* stream);
/* osrl_mlp_forward2 with a forward tail per problem (either may be NULL): e.g. the actor on next_obs and on obs with
 * every action draw of the CPQ step (cpq.py:141,159,164-176,209) made by the same launch. */
int osrl_mlp_forward2_tail(const osrl_mlp_t* net0, const osrl_rows_t* in0, const osrl_mlp_acts_t* out0,
                           const osrl_mlp_tail_t* tail0, const osrl_mlp_t* net1, const osrl_rows_t* in1,
                           const osrl_mlp_acts_t* out1, const osrl_mlp_tail_t* tail1, void* stream);
int osrl_mlp_backward_dz_tail(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                              const osrl_mlp_grads_t* g, const osrl_mlp_tail_t* tail, void* stream);
/* The backward launch COMPUTES the gradient it starts from (new; the reference's autograd starts from a loss scalar,
 * cpq.py:133,151,198,220): the loss kernels between a forward and a backward launch (osrl_vae_loss,
 * osrl_cpq_critic_loss, osrl_cpq_cost_loss, osrl_cpq_actor_loss, osrl_gauss_head_bwd) only turn forward outputs of the
 * SAME ROWS into dL/d(output) plus one logged batch sum -- here every row tile evaluates that expression for its rows
 * while it stages dY (g->dy is ignored and may be NULL), and the statistic is the fixed-order sum of per-tile partials
 * taken by the last workgroup to finish (wait-free: partials are device-coherent words, one arrival atomic per
 * workgroup at its very end).  dL/d(output) has the bits of the named loss kernel; the statistic is the same sum in a
 * different (fixed) order.  Kinds, with y_e = output of net e of this launch and inv = 1 / rows_global (<= 0: rows):
 *   OSRL_SEED_MSE        dy = 2 (y - x0[r, c]) scale;   stat = sum (y - x0)^2 * stat_scale
 *                        [+ kl_beta * stat_scale2 * sum_{r, k < kl_L} KL(kl_head[r, k], kl_head[r, kl_L + k]) if kl_head]
 *                        (VAE: x0 = actions, scale = stat_scale = inv / ad, stat_scale2 = inv / L; == osrl_vae_loss)
 *   OSRL_SEED_CPQ_CRITIC backup = x0[r] + gamma (1 - x1[r]) 1[min_e b_e[r] <= thres] min_e a_e[r];  dy_e = 2 (y_e - backup) inv
 *                        (a = target critics [n_a, rows], b = target cost critics, x0 = rewards, x1 = done; == osrl_cpq_critic_loss)
 *   OSRL_SEED_CPQ_COST   backup = x0[r] + gamma min_e a_e[r];  dy_e = 2 (y_e - backup) inv       (== osrl_cpq_cost_loss, MSE part)
 *   OSRL_SEED_CPQ_ACTOR  dy_e = -1[min b_e[r] <= thres] inv if e == argmin_e a_e[r] else 0; stat = -sum mask min_e a_e * inv
 *                        (a = the critics' outputs [n_a, rows] = this launch's nets, b = the cost critics'; == osrl_cpq_actor_loss)
 *   OSRL_SEED_GAUSS_HEAD dy[r, :] = d/d(mu | log_std) of a = max_action tanh(mu + sd eps) given dL/da = sum_e a_e[r, :]
 *                        (a = [n_a, rows, ad] input gradients, tanh_u / eps [rows, ad]; no statistic; == osrl_gauss_head_bwd)
 *   OSRL_SEED_BCQ_CRITIC backup = x0[r] + gamma (1 - x1[r]) max_j [ thres min(q1_j, q2_j) + (1 - thres) max(q1_j, q2_j) ] over the
 *                        n_samples target samples j of row r (a = [n_a + n_b, rows * n_samples] target outputs, row-major by
 *                        sample: r * n_samples + j; q1 = min over the first n_a, q2 over the last n_b; thres = lambda of
 *                        bcql.py:144-146; x1 may be NULL: no (1 - done)); dy_e = 2 (y_e - backup) inv      (== osrl_bcq_critic_loss)
 * partials: >= 2 * n_nets * ceil(rows / 16) floats of scratch; counter: one uint32, ZERO before the first launch
 * (re-armed by the launch).  HOST struct. */
enum { OSRL_SEED_NONE = 0, OSRL_SEED_MSE = 1, OSRL_SEED_CPQ_CRITIC = 2, OSRL_SEED_CPQ_COST = 3, OSRL_SEED_CPQ_ACTOR = 4,
       OSRL_SEED_GAUSS_HEAD = 5, OSRL_SEED_BCQ_CRITIC = 6 };
typedef struct {
  int32_t kind;
  int32_t n_a, n_b;
  int32_t rows_global;
  const float *a, *b;
  const float *x0, *x1;
  const float *eps, *tanh_u;
  const float* kl_head;
  int32_t kl_L, n_samples;
  float gamma, thres, scale, max_action;
  float stat_scale, stat_scale2, kl_beta, pad2_;
  float* partials;
  uint32_t* counter;
  float* stat;
} osrl_mlp_seed_t;
int osrl_mlp_backward_dz_seed(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                              const osrl_mlp_grads_t* g, const osrl_mlp_tail_t* tail, const osrl_mlp_seed_t* seed,
                              void* stream);
/* ---- one supervised regression step of one MLP in ONE launch (mlp.hip mlp_step_kernel) -------------------------
 * BCTrainer.train_one_step (osrl/algorithms/bc.py:45-55,103-109 with the minibatch of examples/train/train_bc.py:105-121):
 *   [sample + gather the minibatch] -> forward -> F.mse_loss -> backward -> dW / db -> Adam (+ refresh of the packed
 *   weight copies) -> step tick (+ the previous step's statistics into the ring)
 * i.e. exactly osrl_step_begin + osrl_mlp_forward + osrl_mse_loss + osrl_mlp_backward_dz + osrl_mlp_backward_dw_tiles +
 * osrl_adam_step_packed on the same arguments, with the same parameter bits (same MFMA chains, same summation orders,
 * same element-wise update); the loss statistic is summed per 16-row tile and then in tile order (deterministic, not
 * the single-workgroup order of osrl_mse_loss).  At B = 256 those six launches are ~53 us of mostly launch gaps; here
 * the row tiles (16 rows, 8 waves) run gather / forward / loss / backward back to back out of LDS, signal an
 * arrival counter, and the dW tile workgroups -- which cover ALL rows, so there are no gradient slabs -- apply Adam to
 * their tile from LDS.  One grid-wide dependency instead of five launch boundaries.
 * Requirements (else OSRL_E_UNSUPPORTED and nothing is launched; callers keep the six-launch plan): one net; widest
 * layer 129..448 (the 8-wave 16-row tile of the fused MLP kernels); rows <= 16 * OSRL_STEP_MAX_WG; a dW work list in
 * osrl_mlp_backward_dw_tiles' format with ONE row split per tile and <= OSRL_STEP_MAX_WG items; weight decay 0; no
 * target copy.  ws: OSRL_STEP_WS floats of device scratch, zero before the first call, owned by this step (the kernel
 * re-arms its counters; ws[OSRL_STEP_MAX_WG + 2] != 0 afterwards = a workgroup gave up waiting, results invalid).
 * All workgroups of the launch must be resident at once (<= OSRL_STEP_MAX_WG of 256 CUs: they are). HOST struct. */
#define OSRL_STEP_MAX_WG 128
#define OSRL_STEP_WS (OSRL_STEP_MAX_WG + 8)
#define OSRL_E_UNSUPPORTED (-2)
typedef struct {
  /* osrl_step_begin's state arguments */
  osrl_step_state_t* st;
  float beta1, beta2;
  int32_t warmup, n_stats, ring_len;
  /* osrl_replay_gather's arguments; n_fields = 0: the minibatch is already in `in` / `target` */
  int32_t n_fields;
  const float* stats_cur;
  float* ring;
  const float* src[8];
  float* dst[8];
  int32_t width[8];
  float scale[8];
  int64_t n_rows;
  uint64_t gather_seed;
  uint32_t gather_stream, pad0_;
  /* osrl_mlp_forward / osrl_mlp_backward_dz's arguments (grads.dy[0] = the loss gradient buffer, written here) */
  osrl_mlp_t net;
  osrl_rows_t in;
  osrl_mlp_acts_t acts;
  osrl_mlp_grads_t grads;
  /* osrl_mse_loss's arguments */
  const float* target; /* [rows, dims[L]] */
  int64_t n_global;    /* elements the mean runs over (<= 0: rows * dims[L]) */
  float* stat;         /* optional */
  /* osrl_mlp_backward_dw_tiles's arguments; tile_blocks = 4 (64 x 64 tiles) or 2 (32 x 32: four times the
   * workgroups on a quarter of the MFMA work each -- what a 256-row batch wants; same bits, the sum order of an
   * output element depends on its rows-per-wave only) */
  const osrl_dw_entry_t* entries;
  const int32_t* work;
  int32_t n_work, tile_blocks;
  /* osrl_adam_step_packed's arguments (n_splits = 1, no target, weight decay 0) */
  float *p, *m, *v;
  const int32_t *map_f, *map_b;
  float *pf, *pb;
  float lr, eps;
  float* ws;
} osrl_mlp_step_t;
int osrl_mlp_regress_step(const osrl_mlp_step_t* s, void* stream);

/* ---- the VAE's forward and backward on the training rows as ALL-CU layer launches (vae_ns.hip, round 5) -------------
 * VAE.forward + the reconstruction / KL loss + its autograd through decoder and encoder (osrl/common/net.py:290-339,
 * osrl/algorithms/cpq.py:125-135 == bcql.py:122-133 == bearl.py vae_loss): the same results in the same buffers as
 *   osrl_mlp_forward_tail(enc, VAE_LATENT) -> osrl_mlp_forward(dec) -> osrl_mlp_backward_dz_seed(dec, MSE + KL statistic,
 *   VAE_LATENT_BWD tail) -> osrl_mlp_backward_dz(enc)
 * (statistic and gradients equal up to fp32 summation order), launched differently: at 2048 rows a fused 16-row-tile
 * launch of a 400-wide net occupies 128 of the 256 CUs and spends most of its time walking the 400 x 400 layer; here
 * every H x H layer is one launch of [48 rows x 80 columns] output tiles over ALL CUs (4 waves split K, operands
 * requested k-steps ahead, partial tiles meet in LDS), the narrow first layers are one launch for both nets, and the
 * narrow heads / the latent's dX ride as split-K partial rows ("slabs", one per 80-column group) in the epilogue of the
 * wide launch that produces their input and are summed by the prologue of the launch that consumes them:
 *   forward : l0 (enc layer 0 | the observation part of dec layer 0) -> enc wide (+ head slabs) -> dec wide (prologue:
 *             head = sum slabs, z = mean + sd eps, h0 = relu(P + Wz z + b); + output slabs)
 *   backward: dec wide^T (prologue: u = max_action tanh(sum slabs), dY of the MSE, dZ2, the logged loss; A operand =
 *             (dZ2 W2) * relu'(h1); epilogue: dZ0 = . * relu'(h0), + dL/dz slabs) -> enc wide^T (prologue: dL/d(mean |
 *             log_std) through z and the KL term)
 * Requirements (osrl_vae_ns_supported; else callers keep the fused launches): both nets [in, H, H, out] with relu, relu,
 * (id | tanh); H % 80 == 0, 80 <= H <= 448; obs + act <= 128 and obs + L <= 128 columns; L <= 16; act <= 16.
 * HOST struct; every pointer is device memory.  enc / dec: n_nets 1, n_layers 3 (the descriptors of the fused launches). */
typedef struct {
  const osrl_mlp_t* enc;     /* [od + ad, H, H, 2 L]  relu, relu, id   (mean | log_std) */
  const osrl_mlp_t* dec;     /* [od + L,  H, H, ad]   relu, relu, tanh, out_scale = max_action */
  int32_t rows, od, ad, L;
  int32_t rows_global;       /* the loss is a mean over this many rows (<= 0: rows) */
  float beta;                /* KL weight (cpq.py:129) */
  const float *obs, *act, *eps; /* [rows, od], [rows, ad], [rows, L] */
  osrl_mlp_acts_t enc_acts;  /* out: x [rows, od + ad], h[0][0..1] [rows, H], h[0][2] = head [rows, 2 L] */
  osrl_mlp_acts_t dec_acts;  /* out: x [rows, od + L],  h[0][0..1] [rows, H], h[0][2] = u [rows, ad] */
  float* z;                  /* out [rows, L] */
  osrl_mlp_grads_t enc_g;    /* out: dz[0][0..2] (backward) */
  osrl_mlp_grads_t dec_g;    /* out: dz[0][0..2] (backward) */
  float* P;                  /* scratch [rows, H] */
  float* slabs;              /* scratch 3 * (H / 80) * rows * 32 floats */
  float* partials;           /* scratch 2 * ceil(rows / 48) floats: per-tile partials of the logged loss */
  uint32_t* counter;         /* one word, ZERO before the first launch (re-armed by the launch) */
  float* stat;               /* out: loss_vae = mse + beta * kl (cpq.py:127-129) */
} osrl_vae_ns_t;
int osrl_vae_ns_supported(const osrl_vae_ns_t* v);   /* 1 / 0; no launch */
int osrl_vae_ns_forward(const osrl_vae_ns_t* v, void* stream);
int osrl_vae_ns_backward(const osrl_vae_ns_t* v, void* stream);

/* General linear layer on packed weights: Y[M,N] = A[M,K] * P (+ bias[N]) (+ resid[M,N]).  P is the forward
 * pack of W[N,K] (y = x W^T; Np = round16(N), col0 = 0) or the backward pack of W[N',K'] for dx = dy W
 * (then K = N', N = K' or a column slice starting at col0, Np = round16(K')+16).  K <= 1024; N is
 * unbounded (column groups).  Replaces nn.Linear / addmm + residual adds of the CDT block
 * (osrl/common/net.py:406-415,439-440, osrl/algorithms/cdt.py:96-141) and their input gradients. */
int osrl_linear(const float* A, int64_t lda, int32_t M, int32_t K, const float* P, int32_t Np, int32_t col0,
                int32_t N, const float* bias, const float* resid, int64_t ldr, float* Y, int64_t ldy, void* stream);
/* Refresh the packed copies of `n_entries` weights (entries in DEVICE memory).  Sizes in floats:
 * forward round16(in)*round16(out), backward round16(out)*(round16(in)+16).  max_elems = the largest
 * packed size among the entries (grid sizing).  Must run after every change of the canonical weights. */
int osrl_pack_weights(const float* src_flat, float* pf, float* pb, const osrl_pack_entry_t* d_entries,
                      int32_t n_entries, int32_t max_elems, void* stream);
/* entries/items live in DEVICE memory (static plan): items[i] = {entry, o_tile, i_tile, 0} with
 * 64x64 tiles; slabs = [n_splits][slab_stride] partial gradients (deterministic split-K over rows). */
int osrl_mlp_backward_dw(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items,
                         int32_t rows, int32_t n_splits, float* slabs, int64_t slab_stride, void* stream);
/* The same GEMMs on (16 * tile_blocks)^2 tiles (tile_blocks = 4 | 5) with a FLAT work list: d_work[4 i ..] =
 * (entry, out tile, in tile, split | n_splits << 16); every (tile, split) pair is one workgroup and writes slab `split`.
 * Lets the caller give tiles of unequal work unequal row splits and pick 80 x 80 tiles for 400-wide layers (25 column
 * blocks = 5 x 5: no ragged tiles).  Same per-element arithmetic as osrl_mlp_backward_dw. */
int osrl_mlp_backward_dw_tiles(const osrl_dw_entry_t* d_entries, const int32_t* d_work, int32_t n_work, int32_t rows,
                               int32_t tile_blocks, float* slabs, int64_t slab_stride, void* stream);
/* osrl_mlp_backward_dw_tiles AND the optimizer step of the group the gradients belong to, in ONE launch (new: the
 * reference runs loss.backward() and optimizer.step() as separate passes, cpq.py:131-133,149-151,196-198,218-220):
 * every (tile, split) workgroup stores its slab tile as above and signs in at its tile's arrival counter
 * (d_counters[d_tile_ids[i]], uint32, ZERO before the first launch; the last split to arrive re-arms it); that last
 * workgroup sums the tile's n_splits slabs in slab order and applies osrl_adam_step_packed's update (weight decay 0,
 * no gradient scale) to the tile's parameters: p, m, v, the Polyak target (tgt, may be NULL) and the packed copies
 * through map_f / map_b (may be NULL).  Same bits as the two launches it replaces.  Nobody waits inside the kernel (a
 * workgroup either is the last of its tile or leaves), so there is no residency requirement.  The work list must
 * cover every parameter of the group that has a gradient exactly once per split; d_tile_ids[i] is the same for the
 * n_splits items of a tile and different between tiles.  NOT for data-parallel steps (the all-reduce sits between dW
 * and the update there).  HOST struct. */
typedef struct {
  float *p, *m, *v, *tgt;            /* flat group buffers (tgt may be NULL) */
  const int32_t *map_f, *map_b;      /* osrl_adam_step_packed's maps (may be NULL) */
  float *pf, *pb, *tf;               /* packed copies (tf: packed Polyak target, may be NULL) */
  const osrl_step_state_t* st;       /* bias corrections / warm-up factor of the current step */
  float lr, beta1, beta2, eps, tau;
  int32_t pad_;
} osrl_dw_adam_t;
int osrl_mlp_backward_dw_tiles_adam(const osrl_dw_entry_t* d_entries, const int32_t* d_work, const int32_t* d_tile_ids,
                                    uint32_t* d_counters, int32_t n_work, int32_t rows, int32_t tile_blocks,
                                    float* slabs, int64_t slab_stride, const osrl_dw_adam_t* opt, void* stream);
/* The same contract for items given in units of 128 (out) x 64 (in) tiles that lie fully inside their dW (only
 * full tiles may be listed): one wave per tile and row split, 128-register accumulator tiles, one wave per SIMD -- the big-row-count (token matrix) variant; db of an entry is written by its it == 0 tiles.
 * Splits beyond a plan's own n_splits are never written (the caller keeps them zero).
 * The operands are read with 16-byte loads: dz / a of every listed entry 16-byte aligned, row strides % 4 == 0, and the
 * gradient slab 16-byte aligned with w_off % 4 == 0 (the caller checks: entries are device memory). */
int osrl_mlp_backward_dw_big(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items, int32_t rows,
                             int32_t n_splits, float* slabs, int64_t slab_stride, void* stream);
/* The same contract for items in units of 256 (out) x 256 (in) tiles (out % 256 == 0, in % 256 == 0): one 8-wave
 * workgroup per tile and row split, the operands staged through LDS by DMA and shared by the eight waves.  rows % 16 == 0;
 * alignment as above.  db of an entry is written by its it == 0 tiles. */
int osrl_mlp_backward_dw_coop(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items, int32_t rows,
                              int32_t n_splits, float* slabs, int64_t slab_stride, void* stream);

/* ---- optimizer (optim.hip): torch.optim.Adam/AdamW.step + _soft_update (cpq.py:107-113,232-238) ---- */
/* Advance the device step state (t += 1, bias corrections, LR-warmup factor).  If stats_cur/ring are
 * given, first commits the previous step's statistics into ring[((t-1) % ring_len)][n_stats] so that
 * the host can read logged values lazily instead of .item()-syncing every step. */
int osrl_step_tick(osrl_step_state_t* st, float beta1, float beta2, int32_t warmup, const float* stats_cur,
                   float* ring, int32_t n_stats, int32_t ring_len, void* stream);
/* The whole step prologue in ONE launch: osrl_step_tick + osrl_randn_fill (if noise != NULL) + osrl_replay_gather (if
 * n_fields > 0), with exactly their results -- the draws use step t+1, the value the tick is about to store; the last
 * workgroup to have read the old step performs the tick.  Replaces three dependent launches (and their launch gaps)
 * at the head of every train step. */
int osrl_step_begin(osrl_step_state_t* st, float beta1, float beta2, int32_t warmup, const float* stats_cur,
                    float* ring, int32_t n_stats, int32_t ring_len, float* noise, int64_t noise_n,
                    uint64_t noise_seed, uint32_t noise_stream, int32_t n_fields, const float* const* src,
                    float* const* dst, const int32_t* width, const float* scale, int64_t n_rows, int32_t batch,
                    uint64_t gather_seed, uint32_t gather_stream, void* stream);
/* The two calls above for TWO step states that take turns (software-pipelined train steps: step k+1's prologue runs while
 * step k's last optimizer launches still read step k's bias corrections, so consecutive steps cannot share one state).
 * `st` receives t = max(st->step, peer->step) + 1 and commits ITS OWN previous statistics (those of step st->step) to the
 * ring slot of that step; `peer` is only read.  peer == NULL: exactly osrl_step_tick / osrl_step_begin.  The reference has
 * no counterpart (its optimizer steps are host-serialised, cpq.py:294-313); same arithmetic per step as the plain calls. */
int osrl_step_tick_peer(osrl_step_state_t* st, const osrl_step_state_t* peer, float beta1, float beta2, int32_t warmup,
                        const float* stats_cur, float* ring, int32_t n_stats, int32_t ring_len, void* stream);
int osrl_step_begin_peer(osrl_step_state_t* st, const osrl_step_state_t* peer, float beta1, float beta2, int32_t warmup,
                         const float* stats_cur, float* ring, int32_t n_stats, int32_t ring_len, float* noise,
                         int64_t noise_n, uint64_t noise_seed, uint32_t noise_stream, int32_t n_fields,
                         const float* const* src, float* const* dst, const int32_t* width, const float* scale,
                         int64_t n_rows, int32_t batch, uint64_t gather_seed, uint32_t gather_stream, void* stream);
/* g = sum_s slabs[s][i] (* *gscale if gscale != NULL); AdamW decay if weight_decay != 0;
 * then tgt = tau*p + (1-tau)*tgt if tgt != NULL.  n and slab_stride must be multiples of 4. */
int osrl_adam_step(float* p, float* m, float* v, float* tgt, const float* slabs, int32_t n_splits,
                   int64_t slab_stride, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float tau, const float* gscale, const osrl_step_state_t* st,
                   void* stream);
/* Same step, and the packed weight copies (osrl_pack_weights layouts) are refreshed in the same pass:
 * map_f[i] / map_b[i] = float index of flat parameter i inside pf / pb (-1 = not a packed weight; map_b may be
 * NULL); tf (may be NULL) receives the Polyak target at map_f.  Padding of the packed buffers is not touched. */
int osrl_adam_step_packed(float* p, float* m, float* v, float* tgt, const float* slabs, int32_t n_splits,
                          int64_t slab_stride, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float tau, const float* gscale, const osrl_step_state_t* st,
                          const int32_t* map_f, const int32_t* map_b, float* pf, float* pb, float* tf,
                          void* stream);
/* The Polyak step alone: tgt = tau p + (1 - tau) tgt, and tf[map_f[i]] = the new target where map_f[i] >= 0 (map_f / tf may
 * be NULL) -- the bits osrl_adam_step(_packed) leaves when it carries the step.  For a plan whose last reader of the OLD
 * targets runs after the group's optimizer step (reference order: every loss of cpq.py:294-313 reads the targets of the
 * step's start, sync_weight() comes last).  n a multiple of 4. */
int osrl_polyak(const float* p, float* tgt, int64_t n, float tau, const int32_t* map_f, float* tf, void* stream);
/* flat[i] = sum_s slabs[s][i]  (pre-reduction before an RCCL all-reduce in the data-parallel path) */
int osrl_reduce_slabs(float* flat, const float* slabs, int32_t n_splits, int64_t slab_stride, int64_t n,
                      void* stream);
/* The same sum with a row-split count per 1024-float chunk of the flat gradient (counts: device bytes, ceil(n / 1024) of
 * them, each >= 1): a group whose ranges were written by plans with different split counts (CDT: cdt.py:396-400 needs the
 * summed gradient for clip_grad_norm_) reads only the slabs that hold something.  Same bits as osrl_reduce_slabs with the
 * largest count, since the slabs beyond a range's own count are zero. */
int osrl_reduce_slabs_counts(float* flat, const float* slabs, const uint8_t* counts, int64_t slab_stride, int64_t n,
                             void* stream);

/* ---- device-resident argument blocks for the launches of a captured step (csrc/argmem.h) --------------------
 * New (nothing in the reference to mirror: torch passes kernel arguments through the runtime).  The fused-MLP
 * launches carry 1.2-2.4 KB descriptors by value; where the HIP runtime keeps kernel arguments in HOST memory (no
 * large BAR, or HIP_FORCE_DEV_KERNARG=0) every wave fetches them over PCIe and a train step loses ~20 %.  A step
 * engine's launches are static, so the caller may keep the descriptors in HBM instead:
 *   1. osrl_args_begin(staging, NULL, cap, 0, 1)   record: run the step once (eagerly); every fused-MLP launch also
 *      copies its descriptor into `staging` (host memory, identical blocks once);  osrl_args_end(&used, ...);
 *   2. copy staging[0:used] to device memory `dev` (any hipMemcpy);
 *   3. osrl_args_begin(staging, dev, cap, used, 2)  replay: run the step again (typically under hipGraph capture);
 *      a launch whose descriptor is found in the staging buffer BY CONTENT starts the kernel variant that reads it
 *      from `dev`; anything else is launched by value as always (counted in n_misses);  osrl_args_end(...).
 * The context belongs to the calling thread; without it nothing changes.  `dev` must outlive the captured graph.
 * Returns 0, -1 on bad arguments / no open context, -2 when a context is already open on this thread. */
int osrl_args_begin(void* host_staging, const void* dev_copy, int64_t capacity, int64_t used, int32_t mode);
int osrl_args_end(int64_t* used, int32_t* n_blocks, int32_t* n_hits, int32_t* n_misses);

/* ---- RNG + replay (rng.hip): torch.randn / TransitionDataset sampling (dataset.py:832-847) ---- */
/* out[i] ~ N(0,1): Philox4x32-10 keyed by seed, counter = (i/4, st->step, stream_id), Box-Muller. */
int osrl_randn_fill(float* out, int64_t n, uint64_t seed, uint32_t stream_id, const osrl_step_state_t* st,
                    void* stream);
/* idx[b] ~ U{0..n_rows-1} (with replacement) and gather of `n_fields` (<= 8) row-major fp32 tables:
 * dst[f][b, :width[f]] = src[f][idx[b], :width[f]] * scale[f].  src/dst/width/scale are HOST arrays
 * of device pointers / ints / floats (copied into the launch). idx_out (device int32[batch]) optional. */
int osrl_replay_gather(int32_t n_fields, const float* const* src, float* const* dst, const int32_t* width,
                       const float* scale, int64_t n_rows, int32_t batch, int32_t* idx_out, uint64_t seed,
                       uint32_t stream_id, const osrl_step_state_t* st, void* stream);

/* CDT minibatch source -- SequenceDataset.__iter__/__prepare_sample (dataset.py:749-787) on device: per sample
 * draw a trajectory (inverse CDF of `cdf`, or uniform when NULL) and a start ~ U{0..len-1}, slice seq_len steps
 * of the concatenated trajectory tables (clipped at the trajectory end), zero-pad the tail, emit mask,
 * time_steps = start + arange(T), returns*reward_scale, cost_returns*cost_scale, episode_cost =
 * cost_returns[first step]*cost_scale.  idx_out (optional) receives (trajectory, start) per sample.
 * start_cdf (optional, [total rows]): inclusive cumulative start-index probabilities inside each trajectory
 * (SequenceDataset(start_sampling=True), dataset.py:742-744,781-783; osrl_start_index_prob) -- NULL = uniform starts.
 * idx_in (optional, [B,2]): (trajectory, start) pairs given by the caller instead of drawn (fixed evaluation windows). */
int osrl_seq_window_gather(const float* obs, const float* act, const float* returns, const float* cost_returns,
                           const float* costs, const int64_t* traj_start, const int32_t* traj_len, const float* cdf,
                           const float* start_cdf, const int32_t* idx_in,
                           int32_t n_traj, int32_t B, int32_t T, int32_t od, int32_t ad, float reward_scale,
                           float cost_scale, float* o_states, float* o_actions, float* o_returns,
                           float* o_cost_returns, int64_t* o_time_steps, float* o_mask, float* o_episode_cost,
                           float* o_costs, int32_t* idx_out, uint64_t seed, uint32_t stream_id,
                           const osrl_step_state_t* st, void* stream);

/* ---- glue (glue.hip): the elementwise / reduction tails of the loss functions ----
 * `rows_global` (0 = rows) is the data-parallel global batch used in every 1/B normalisation.
 * `stat` pointers are device floats (logged statistics), may be NULL. */
/* SquashedGaussianMLPActor tail net.py:176-201: head=[rows,2*ad]=(mu|log_std_raw);
 * u = mu + exp(clamp(ls,-20,2))*eps (eps NULL => deterministic), a = max_action*tanh(u).
 * Outputs (each may be NULL): a[rows,ad], tanh_u[rows,ad], logp[rows]. */
int osrl_gauss_head(const float* head, const float* eps, int32_t rows, int32_t ad, float max_action,
                    float* a, float* tanh_u, float* logp, void* stream);
/* d head from d a (autograd of the tail above); da_nets = [n_nets][rows,ad] is summed over nets. */
int osrl_gauss_head_bwd(const float* head, const float* eps, const float* tanh_u, const float* da_nets,
                        int32_t n_nets, int32_t rows, int32_t ad, float max_action, float* dhead,
                        void* stream);
/* pi_dist.sample([N]) cpq.py:166-169: out[j*rows+b,:] = mu[b] + std[b]*eps[j,b,:]  (pre-tanh) */
int osrl_gauss_ood_sample(const float* head, const float* eps, int32_t n_samples, int32_t rows, int32_t ad,
                          float* out, void* stream);
/* VAE latent net.py:323-327: head=[rows,2L]=(mean|log_std_raw) -> z = mean + exp(clamp(ls,-4,15))*eps */
int osrl_vae_latent(const float* head, const float* eps, int32_t rows, int32_t L, float* z, void* stream);
/* vae_loss cpq.py:125-129 == bcql.py:122-126: stat = mse(u, act) + beta*KL(head); du = d loss / d u. */
int osrl_vae_loss(const float* u, const float* act, const float* head, int32_t rows, int32_t ad, int32_t L,
                  float beta, int32_t rows_global, float* du, float* stat, void* stream);
/* d head of the VAE latent given dz = d loss / d z from the decoder backward (+ the KL term). */
int osrl_vae_latent_bwd(const float* head, const float* eps, const float* dz, int32_t rows, int32_t L,
                        float beta, int32_t rows_global, float* dhead, void* stream);
/* per-row mean over the latent of the KL term (cpq.py:181-182): kl[r] */
int osrl_vae_kl_rows(const float* head, int32_t rows, int32_t L, float* kl, void* stream);
/* torch.quantile(x, q) (linear interpolation) by an exact 4-pass radix select; out[0] = quantile. */
int osrl_quantile(const float* x, int64_t n, float q, float* out, void* stream);

/* CPQ critic target + loss gradient (cpq.py:145-148, net.py:240-242).
 * q_old/qc_old/q are [n][rows] net-major outputs. dq[e][rows] = 2(q_e-backup)/B; stat = sum_e mse. */
int osrl_cpq_critic_loss(const float* q_old, int32_t n_q_old, const float* qc_old, int32_t n_qc_old,
                         const float* q, int32_t n_q, const float* rew, const float* done, int32_t rows,
                         float gamma, float q_thres, int32_t rows_global, float* dq, float* stat, void* stream);
/* The same quantile for large n (the all-gathered KL rows of data-parallel CPQ: world * N * B values) as a grid: four
 * multi-workgroup histogram passes + a successor search + a finish launch instead of one streaming workgroup (200 us
 * at n = 163840).  ws: OSRL_QUANTILE_WS uint32 of device scratch, all zero before the first call; every call leaves it
 * zeroed again.  Same result bits as osrl_quantile. */
#define OSRL_QUANTILE_WS 1032
int osrl_quantile_ws(const float* x, int64_t n, float q, uint32_t* ws, float* out, void* stream);
/* out[0] = mean over the (global) batch of qc_ood = ((KL >= quantile) * min_e qc_sampled).mean(0)
 * (cpq.py:184,187); under data parallelism this rank's share, to be all-reduced(SUM). */
int osrl_cpq_ood_mean(const float* qc_sampled, int32_t n_qc_old, const float* kl, const float* quantile,
                      int32_t n_samples, int32_t rows, int32_t rows_global, float* out, void* stream);
/* osrl_quantile(kl, n_samples*rows, q) + osrl_cpq_ood_mean in ONE launch (cpq.py:183-184,187; single-GPU step):
 * quant_out[0] = the quantile, out[0] = the OOD mean; same bits as the two calls.  n_samples*rows <= 32768 (-2). */
int osrl_cpq_ood_stat(const float* qc_sampled, int32_t n_qc_old, const float* kl, float q, int32_t n_samples,
                      int32_t rows, int32_t rows_global, float* quant_out, float* out, void* stream);
/* cpq.py:183-184 as a ROW SET instead of a mask: qc_ood = ((KL >= quantile) * qc_sampled).mean(0) multiplies three
 * quarters of the N*B target-cost-critic outputs by zero; with the rows that count known BEFORE those networks run, they
 * run on that quarter only (osrl_rows_t.row_list / n_rows_dev).
 *   osrl_cpq_ood_select: quantile = *quantile_in if given, else the q-quantile of kl[0..n) (torch.quantile 'linear',
 *     n <= 32768: -2 otherwise); list[0..count) = the indices i, ascending, with kl[i] >= quantile; count[0], quant_out[0].
 *   osrl_cpq_ood_sum: out[0] = scale * sum_{r < min(*count, cap)} min_e qc_sel[e * cap + r]   (scale = 1 / (n_samples *
 *     rows_global): the mean over the batch of the mean over the samples; fixed summation order).
 * Equal to osrl_cpq_ood_stat on the full rows up to the order of the sum (and to a non-finite critic output on a row
 * OUTSIDE the set, which the reference's multiplication by zero would turn into NaN). */
int osrl_cpq_ood_select(const float* kl, const float* quantile_in, float q, int32_t n, float* quant_out, int32_t* list,
                        int32_t* count, void* stream);
int osrl_cpq_ood_sum(const float* qc_sel, int32_t n_qc, int32_t cap, const int32_t* count, float scale, float* out,
                     void* stream);
/* CPQ cost-critic loss (cpq.py:161,186-199): backup = c + gamma*min qc_old; dq = 2(qc-backup)/B;
 * log_alpha (device scalar) ascends with the GLOBAL ood_mean (device scalar) and is clamped to +-5;
 * stat[0] = loss, stat[1] = exp(log_alpha) after the update.  stat_share = 1/world_size scales the
 * batch-global terms of the statistics so that an all-reduce(SUM) of per-rank stats is exact. */
int osrl_cpq_cost_loss(const float* qc_old_next, int32_t n_qc_old, const float* qc, int32_t n_qc,
                       const float* ood_mean, const float* cost, int32_t rows, float gamma, float qc_thres,
                       float alpha_lr, int32_t rows_global, float stat_share, float* log_alpha, float* dq,
                       float* stat, void* stream);
/* ood_mean == NULL in osrl_cpq_cost_loss defers the dual step: stat[0] = the MSE part only and log_alpha is left
 * alone (the gradient dq never depends on ood_mean, cpq.py:186-199 -- qc_ood is under no_grad and log_alpha has no
 * grad); osrl_cpq_alpha_step then applies it once the GLOBAL ood_mean is known: stat[0] -= share*exp(la)*(ood-thres),
 * log_alpha += alpha_lr*exp(la)*(thres - ood), clamp +-5, stat[1] = share*exp(log_alpha).  Data parallelism uses
 * this to fold the ood_mean reduction into the gradient all-reduce of the phase. */
int osrl_cpq_alpha_step(const float* ood_mean, float qc_thres, float alpha_lr, float stat_share, float* log_alpha,
                        float* stat, void* stream);
/* Single-GPU fusion of osrl_cpq_ood_mean + osrl_cpq_cost_loss (cpq.py:184-199): the OOD mean is computed inside
 * the loss launch (and written to ood_mean_out); no batch-global reduction can sit between the two. */
int osrl_cpq_cost_loss_ood(const float* qc_sampled, int32_t n_qc_sampled, const float* kl, const float* quantile,
                           int32_t n_samples, const float* qc_old_next, int32_t n_qc_old, const float* qc,
                           int32_t n_qc, float* ood_mean_out, const float* cost, int32_t rows, float gamma,
                           float qc_thres, float alpha_lr, float* log_alpha, float* dq, float* stat, void* stream);
/* CPQ actor loss (cpq.py:210-212): loss = -mean(1[min qc <= thres] * min q); dq routed to arg-min net. */
int osrl_cpq_actor_loss(const float* q, int32_t n_q, const float* qc, int32_t n_qc, int32_t rows,
                        float q_thres, int32_t rows_global, float* dq, float* stat, void* stream);
/* F.mse_loss (bc.py:46-47): stat = mean((u-target)^2) over n elements; du = 2(u-target)/n_global */
int osrl_mse_loss(const float* u, const float* target, int64_t n, int64_t n_global, float* du, float* stat,
                  void* stream);

/* in-place clamp: the `.clamp(-0.5, 0.5)` of VAE.decode's latent draw (net.py:334-335) */
int osrl_clamp(float* x, int64_t n, float lo, float hi, void* stream);
/* BCQ-Lag perturbation tail net.py:61-62: a = clamp(dec + phi*max_a*t, +-max_a) */
int osrl_bcq_perturb(const float* dec, const float* t, int32_t rows, int32_t ad, float phi, float max_action,
                     float* a, void* stream);
/* d t of the perturbation actor from d a (da_nets [n_nets][rows,ad] summed), through the clamp */
int osrl_bcq_perturb_bwd(const float* dec, const float* t, const float* da_nets, int32_t n_nets, int32_t rows,
                         int32_t ad, float phi, float max_action, float* dt, void* stream);
/* BCQ-Lag target (bcql.py:144-150): q_t[e][rows*N] (first n1 = q1 nets, then n2 = q2 nets), row b*N+j ->
 * backup[b] = base[b] + gamma*(1-done[b])*max_j(lmbda*min(q1,q2)+(1-lmbda)*max(q1,q2)); then
 * dq[e][b] = 2*(q_on[e][b]-backup[b])/B, stat = sum_e mse.  done may be NULL (cost critic, bcql.py:172). */
int osrl_bcq_critic_loss(const float* q_t, int32_t n1, int32_t n2, int32_t n_samples, const float* q_on,
                         int32_t n_on, const float* base, const float* done, int32_t rows, float gamma,
                         float lmbda, int32_t rows_global, float* dq, float* stat, void* stream);
/* The two batch-sum losses on a grid of <= 64 workgroups (for batches beyond ~2048 rows, where one workgroup walking the
 * batch alone costs 35-45 us): same values per row, the logged statistic is the sum of per-workgroup partials added in
 * workgroup order by the last workgroup to arrive (deterministic; no floating-point atomics).  ws: OSRL_LOSS_WS floats
 * of device memory, zeroed once before first use, private to one call site while a launch is in flight. */
#define OSRL_LOSS_WS 132
int osrl_bcq_critic_loss_ws(const float* q_t, int32_t n1, int32_t n2, int32_t n_samples, const float* q_on, int32_t n_on,
                            const float* base, const float* done, int32_t rows, float gamma, float lmbda,
                            int32_t rows_global, float* dq, float* stat, float* ws, void* stream);
int osrl_vae_loss_ws(const float* u, const float* act, const float* head, int32_t rows, int32_t ad, int32_t L, float beta,
                     int32_t rows_global, float* du, float* stat, float* ws, void* stream);
/* BCQ-Lag actor loss (bcql.py:190-198 + PID net.py:376-387).  q/qc = [n1+n2][rows] (q1 nets then q2 nets).
 * pid = device {error_old, error_integral}; stat: [0]=loss [1]=qc_penalty [2]=multiplier.
 * Data parallel: global_means = all-reduced {mean q_pi, mean qc_pi} from osrl_bcq_actor_sums (NULL = compute
 * locally), stat_share = 1/world. */
int osrl_bcq_actor_sums(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                        int32_t rows, int32_t rows_global, float* out, void* stream);
int osrl_bcq_actor_loss(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                        int32_t rows, float qc_thres, float KP, float KI, float KD, int32_t rows_global,
                        const float* global_means, float stat_share, float* pid, float* dq, float* dqc, float* stat,
                        void* stream);

/* ---- CDT (cdt.hip): the non-GEMM pieces of the Constrained Decision Transformer step ---- */
/* Token embeddings + timestep embedding, interleave and emb LayerNorm (cdt.py:178-222).  Tokens per timestep, in
 * order: [return if use_rew] [cost if use_cost] state action (R = 2..4); prefix != 0: one more token leads each
 * sequence, Linear(1,E) of episode_cost[b] (cdt.py:207-218); timestep_emb == NULL: time_emb = False.  S = R*T + prefix.
 * Outputs: seq[B*S,E] (pre-LN), x0[B*S,E], stats[B*S,2] = (mean, rstd), ctg_t[B*T] = the (optionally 50 - x
 * transformed, cdt.py:78-81) cost-to-go fed to cost_emb.  Pointers of absent tokens may be NULL. */
int osrl_cdt_embed_ln(const float* states, const float* actions, const float* returns, const float* costs_to_go,
                      const float* episode_cost, const int64_t* time_steps, const float* Ws, const float* bs,
                      const float* Wa, const float* ba, const float* Wc, const float* bc, const float* Wr,
                      const float* br, const float* Wp, const float* bp, const float* timestep_emb, const float* ln_g,
                      const float* ln_b, int32_t B, int32_t T, int32_t od, int32_t ad, int32_t E,
                      int32_t cost_transform, int32_t use_rew, int32_t use_cost, int32_t prefix, float* seq, float* x0,
                      float* stats, float* ctg_t, void* stream);
/* nn.LayerNorm (eps 1e-5) of x (+ delta): y = LN(x + delta); xout = x + delta if non-NULL; stats = (mean, rstd). */
int osrl_layernorm_fwd(const float* x, const float* delta, const float* gamma, const float* beta, float* xout,
                       float* y, float* stats, int32_t M, int32_t E, void* stream);
/* LayerNorm backward: dx = LN'(dy) (+ dres); dgamma/dbeta summed (fixed order, via partial_ws [n_parts, 2E]) into
 * slab[g_off..], slab[b_off..]. */
int osrl_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* dres,
                       float* dx, float* partial_ws, int32_t n_parts, int32_t M, int32_t E, float* slab,
                       int64_t g_off, int64_t b_off, void* stream);
/* The same two calls with the nn.Dropout of the residual branch folded in (net.py:414,439; round 3): forward applies
 * the keep-multiplier of `drop` to `delta` on the way in (== osrl_dropout(delta) then osrl_layernorm_fwd, bit for bit;
 * drop NULL or p <= 0: the plain call); backward also writes dx_dropped = dx * keep-multiplier of `drop` (the gradient
 * that enters the residual branch whose output fed this LayerNorm's input: == osrl_dropout on dx). */
/* slab == NULL in the two backward calls: the per-workgroup partials of (dgamma | dbeta) stay in partial_ws
 * ([n_parts, 2E]) and the caller sums them later -- osrl_layernorm_param_reduce does it for up to 16 LayerNorms in ONE
 * launch (site k: partials at partial_ws + k * ws_stride, results at slab[g_offs[k] ..], slab[b_offs[k] ..]; g_offs /
 * b_offs are HOST arrays).  Same bits as the per-call reduction (net.py:402-403,427,440 under autograd). */
int osrl_layernorm_param_reduce(const float* partial_ws, int64_t ws_stride, int32_t n_sites, int32_t n_parts, int32_t E,
                                float* slab, const int64_t* g_offs, const int64_t* b_offs, void* stream);
int osrl_layernorm_fwd_drop(const float* x, const float* delta, const osrl_dropout_t* drop, const float* gamma,
                            const float* beta, float* xout, float* y, float* stats, int32_t M, int32_t E,
                            void* stream);
int osrl_layernorm_bwd_drop(const float* dy, const float* x, const float* stats, const float* gamma, const float* dres,
                            float* dx, float* dx_dropped, const osrl_dropout_t* drop, float* partial_ws, int32_t n_parts,
                            int32_t M, int32_t E, float* slab, int64_t g_off, int64_t b_off, void* stream);
/* nn.MultiheadAttention core with the block's causal mask and key padding (net.py:417-435): qkv [B,S,3E] (q|k|v,
 * heads split the E axis contiguously), mask [B, (S-prefix)/rep] (1 = valid timestep, each repeated rep times along S;
 * prefix = 1: token 0 is the cost-prefix token, masked like timestep 0, cdt.py:216-218). */
int osrl_attention_fwd(const float* qkv, const float* mask, int32_t B, int32_t S, int32_t E, int32_t H, int32_t rep,
                       int32_t prefix, const osrl_dropout_t* drop, float* o, void* stream);
int osrl_attention_bwd(const float* qkv, const float* mask, const float* dout, int32_t B, int32_t S, int32_t E,
                       int32_t H, int32_t rep, int32_t prefix, const osrl_dropout_t* drop, float* dqkv, void* stream);
/* The same pair with the attention-probability dropout's keep decisions handed from the forward launch to the backward
 * launch instead of being regenerated there (head widths E / H = 16 or 32; net.py:406-409 -- torch keeps its dropout mask
 * for autograd the same way).  `keep`: caller-allocated, osrl_attention_keep_bytes(B, S, E, H) bytes (0 = shape not
 * supported: pass NULL), one nibble of four key decisions per byte; written by _fwd_keep when drop is active, read by
 * _bwd_keep of the SAME step.  The decisions are the Philox words' tests either way: results are bit-identical to the plain
 * pair, which the NULL form is. */
int64_t osrl_attention_keep_bytes(int32_t B, int32_t S, int32_t E, int32_t H);
int osrl_attention_fwd_keep(const float* qkv, const float* mask, int32_t B, int32_t S, int32_t E, int32_t H, int32_t rep,
                            int32_t prefix, const osrl_dropout_t* drop, float* o, unsigned char* keep, void* stream);
int osrl_attention_bwd_keep(const float* qkv, const float* mask, const float* dout, int32_t B, int32_t S, int32_t E,
                            int32_t H, int32_t rep, int32_t prefix, const osrl_dropout_t* drop, float* dqkv,
                            const unsigned char* keep, void* stream);
/* nn.Dropout in training mode (cdt.py:87,222 embedding; net.py:404,439 residual; net.py:414 MLP tail):
 * y[i] = x[i] * keep_i / (1-p), may run in place.  keep_i is a pure function of (seed, st->step, site, i)
 * (Philox4x32-10), so calling it again on the incoming gradient IS the backward pass; nothing is stored.
 * `drop` of the attention entry points is the attention-probability dropout of nn.MultiheadAttention
 * (net.py:406-409), NULL or p = 0 = off; its logical mask layout is [B*H, S, Sp] (Sp = S rounded up to 16) with element
 * (bh*S + i)*Sp + j <-> P[bh][i][j], so osrl_dropout on a ones tensor of that size exports it. */
int osrl_dropout(const float* x, float* y, int64_t n, const osrl_dropout_t* drop, void* stream);
/* nn.GELU() (exact erf) and its derivative; n a multiple of 4 */
int osrl_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int osrl_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* CDTTrainer losses (cdt.py:357-394): head = (mu|log_std)[BT,2ad] (stochastic) or action prediction [BT,ad];
 * writes d head / d logits / d state_pred and stat[0..8] = nll, ent, ent_reg, all_loss, act_loss, cost_loss,
 * cost_acc, state_loss, train_lr; ent_out[0] = entropy (input of the temperature step).
 * ws (optional, >= 8 * ceil(B*T / 1024) floats) together with `counts`: one workgroup per 1024 tokens + an ordered
 * sum of their partials instead of one workgroup for the whole batch. */
int osrl_cdt_loss(const float* head, const float* logits, const float* state_pred, const float* actions,
                  const float* states, const float* mask, const float* costs, int32_t B, int32_t T, int32_t od,
                  int32_t ad, int32_t stochastic, int32_t no_entropy, const float* log_temperature, float cost_w,
                  float state_w, float lr, int32_t warmup, const osrl_step_state_t* st, const float* counts,
                  int32_t world, float* dhead, float* dlogits, float* dsp, float* stat, float* ent_out, float* ws,
                  void* stream);
/* out = {#(mask > 0), sum(mask)}: the count-normalisers of cdt.py:358-359,386, to be all-reduced under data
 * parallelism and handed to osrl_cdt_loss as `counts` (with `world` = number of equal-sized ranks). */
int osrl_cdt_mask_counts(const float* mask, int32_t BT, float* out, void* stream);
/* d timestep_emb[time_steps[b,t]] += sum of the R token rows of timestep (b,t) in dseq [B, R*T + prefix, E] (atomic
 * scatter) */
int osrl_cdt_timestep_scatter(const float* dseq, const int64_t* time_steps, int32_t B, int32_t T, int32_t R,
                              int32_t prefix, int32_t E, float* dte, void* stream);
/* torch.nn.utils.clip_grad_norm_ (cdt.py:398-399): out[0] = min(1, clip/(||grad||+1e-6)), out[1] = ||grad|| */
int osrl_clip_grad_scale(const float* grad, int64_t n, float clip, float* partial_ws, int32_t n_parts, float* out,
                         void* stream);
/* Adam step on log_temperature with loss exp(logT)*(entropy - target) (cdt.py:402-407); moments = {m, v} */
int osrl_cdt_temperature_step(float* log_temperature, float* moments, const float* entropy, float target_entropy,
                              float lr, float beta1, float beta2, float eps, const osrl_step_state_t* st, void* stream);

/* ---- BEAR-Lagrangian glue (osrl/algorithms/bearl.py) ----
 * Rows are batch-major: row b*M + j = sample j of batch element b (repeat_interleave, bearl.py:224-226).
 * osrl_bear_mmd: u = mu + exp(clamp(log_std)) * eps from the actor head [B*M, 2*ad] (net.py:176-186);
 *   mmd[b] = mmd_loss_gaussian / mmd_loss_laplacian(raw_vae[b], u[b], sigma) (bearl.py:277-312);
 *   du = d mmd[b] / d u (unscaled); tanh_u = tanh(u); a0[b] = tanh(u[b, 0]) (the critics' action, bearl.py:243-245).
 *   M <= 64, ad <= 16. */
enum { OSRL_MMD_GAUSSIAN = 0, OSRL_MMD_LAPLACIAN = 1 };
int osrl_bear_mmd(const float* raw_vae, const float* head, const float* eps, int32_t rows, int32_t n_samples,
                  int32_t ad, float sigma, int32_t kernel, float* mmd, float* du, float* tanh_u, float* a0,
                  void* stream);
/* out = {mean min(q1,q2), mean min(qc1,qc2), mean mmd} over rows / rows_global: the data-parallel pre-pass (the PID
 * controller and the dual step need GLOBAL means); all-reduce it and pass it as global_means below. */
int osrl_bear_actor_sums(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                         const float* mmd, int32_t rows, int32_t rows_global, float* out, void* stream);
/* The scalar side of BEARL.actor_loss (bearl.py:246-275): PID multiplier (net.py:376-387, state pid[2]); loss =
 * mean(-q [only once n_train_steps = st->step - 1 >= start_update_policy_step] + exp(log_alpha)*(mmd - thresh)) +
 * mean((qc - qc_thres) * multiplier); dq / dqc = d loss / d (every Q-net output) with torch.min's routing;
 * coef[0] = exp(log_alpha) / rows_global BEFORE the dual step log_alpha += alpha_lr*exp(log_alpha)*mean(mmd - thresh),
 * clamp [-5, 5]; stat[0..5) = actor_loss, mmd_loss, qc_penalty, lagrangian, alpha_value (x stat_share). */
int osrl_bear_actor_loss(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                         const float* mmd, int32_t rows, float qc_thres, float KP, float KI, float KD,
                         float target_mmd_thresh, float alpha_lr, int64_t start_update_policy_step,
                         int32_t rows_global, const float* global_means, float stat_share,
                         const osrl_step_state_t* st, float* pid, float* log_alpha, float* dq, float* dqc, float* coef,
                         float* stat, void* stream);
/* dhead[B*M, 2*ad] = d loss / d (mu, log_std): du = coef[0]*du_mmd + [sample 0] sum_nets da_nets[n, b] * (1 - tanh_u^2);
 * d mu = du, d log_std = du * eps * std inside the clamp range (net.py:180). */
int osrl_bear_head_bwd(const float* head, const float* eps, const float* tanh_u, const float* du_mmd, const float* coef,
                       const float* da_nets, int32_t n_nets, int32_t rows, int32_t n_samples, int32_t ad, float* dhead,
                       void* stream);

/* ---- COptiDICE glue (osrl/algorithms/coptidice.py) ----
 * nu2 / chi2 = EnsembleQCritic outputs [n_nets, 2*rows] on the stacked rows [obs; next_obs].
 * leaves = float[6] {tau, m, v, lmbda, m, v}: the raw scalar leaves (coptidice.py:96-97) with their Adam moments;
 * work = float[4] {softplus(lmbda), softplus(tau), weighted_c, -} of the current step. */
enum { OSRL_F_CHI2 = 0, OSRL_F_SOFTCHI = 1, OSRL_F_KL = 2 }; /* get_f_div_fn, coptidice.py:15-38 */
/* _optimal_w (coptidice.py:122-131): e = r - lambda' c + gamma (1-d) min nu(s') - min nu(s), w = relu(f'^-1(e/alpha)).
 * use_saved_lambda = 0: lambda' = softplus(leaves[3]) and work[0..1] are (re)written (top of update(), :138);
 * 1: lambda' = work[0] (policy extraction after the lambda step, :211-213).  e may be NULL. */
int osrl_dice_optimal_w(const float* nu2, int32_t n_nu, int32_t rows, const float* rew, const float* cost,
                        const float* done, const float* leaves, float* work, int32_t use_saved_lambda, float alpha,
                        float gamma, int32_t f_type, float* e, float* w, void* stream);
/* coptidice.py:149-185: ell, the softmax over the WHOLE batch, D_kl, weighted_c -> work[2], chi_loss and its gradient
 * dchi [n_chi, 2*rows] (weights are not detached in the reference), Adam step on tau; stat[0..3) = chi_loss, tau_loss,
 * D_kl.  chi2 == NULL (cost_ub_epsilon == 0): weighted_c = mean(w c), zero stats, no update.
 * Data parallel: osrl_dice_chi_ell writes this rank's ell [rows]; the caller all-gathers it into ell_all
 * [rows_global] and passes it with this rank's first row row0: the softmax statistics are then the GLOBAL ones
 * (identical on every rank, written x stat_share because the statistics vector is all-reduced), work[2] is this
 * rank's share of weighted_c (all-reduce it), dchi covers this rank's rows.  ell_all == NULL: one GPU. */
int osrl_dice_chi_ell(const float* chi2, int32_t n_chi, int32_t rows, const float* w, const float* cost,
                      const float* done, const float* is_init, float gamma, float init_state_propotion, float* ell,
                      void* stream);
int osrl_dice_chi_step(const float* chi2, int32_t n_chi, int32_t rows, const float* w, const float* cost,
                       const float* done, const float* is_init, float gamma, float init_state_propotion,
                       float cost_ub_epsilon, float scalar_lr, const osrl_step_state_t* st, float* leaves, float* work,
                       float* ell_ws, float* dchi, const float* ell_all, int32_t rows_global, int32_t row0,
                       float stat_share, float* stat, void* stream);
/* coptidice.py:147,188-201: Df, td_error, nu_loss (this rank's share of the global means: 1/rows_global) and dnu
 * [n_nu, 2*rows]; lmbda_loss and the Adam step on lmbda from the GLOBAL weighted_c in work[2];
 * stat[0..7) = Df, td_error, nu_loss, lmbda_loss, (untouched: actor_loss), tau', lambda' (the last three x stat_share). */
int osrl_dice_nu_step(const float* nu2, int32_t n_nu, int32_t rows, const float* e, const float* w, const float* done,
                      const float* is_init, int32_t f_type, float gamma, float alpha, float init_state_propotion,
                      float qc_thres, float scalar_lr, int32_t rows_global, float stat_share,
                      const osrl_step_state_t* st, float* leaves, const float* work, float* dnu, float* stat,
                      void* stream);
/* out = x + eps * std[col] * scale (coptidice.py:204-205; std is [1, dim]) */
int osrl_dice_perturb(const float* x, const float* eps, const float* std, int32_t rows, int32_t dim, float scale,
                      float* out, void* stream);
/* coptidice.py:207-215: actor_loss = -mean(w * sum Normal(mu, exp(clamp(log_std))).log_prob(act)) (pre-tanh Gaussian,
 * mean over rows_global), dhead [rows, 2*ad] = its gradient w.r.t. (mu, log_std); stat[0] = actor_loss (share). */
int osrl_dice_actor_loss(const float* head, const float* act, const float* w, int32_t rows, int32_t ad,
                         int32_t rows_global, float* dhead, float* stat, void* stream);

/* ---- dataset ingestion on device (osrl/common/dataset.py) ----
 * The flat DSRL arrays (observations, actions, rewards, costs, terminals, timeouts) are uploaded once; these calls
 * produce, in HBM, what the reference computes in host python loops before training.  `ws` is an int32 workspace of
 * osrl_ingest_ws_elems(n) elements; n < 2^31.  Index outputs are exact, fp32 recurrences bit-identical to numpy. */
int64_t osrl_ingest_ws_elems(int64_t n);
/* Episode split at `terminals == 1 || timeouts == 1` (dataset.py:60, :165): ep_end[e] = index of the e-th done
 * flag, ep_start[e] = ep_end[e-1] + 1, ep_len[e]; *n_episodes = number of COMPLETE episodes (transitions after the
 * last done flag belong to none, as in process_sequence_dataset dataset.py:156-175).  ep_* are sized n. */
int osrl_episode_segments(const float* terminals, const float* timeouts, int64_t n, int64_t* ep_end, int64_t* ep_start,
                          int32_t* ep_len, int32_t* n_episodes, int32_t* ws, void* stream);
/* discounted_cumsum (dataset.py:19-27) per episode: out[t] = x[t] + gamma*out[t+1].  reverse: x -> 1 - x first
 * (cost_reverse, dataset.py:161-162; x_out, optional, receives the transformed x).  broadcast_first: every step of
 * the episode gets out[first step] (process_bc_dataset dataset.py:63-70; steps outside episodes are not written). */
int osrl_episode_returns(const float* x, const int64_t* ep_start, const int32_t* ep_len, int32_t n_episodes,
                         float gamma, int32_t reverse, int32_t broadcast_first, float* out, float* x_out, void* stream);
/* compute_cost_sample_prob (dataset.py:439-459): prob[e] = max(T(c_e), 0) / sum, c_e = cost_returns[ep_start[e]],
 * T(c) = a*c + b (OSRL_COST_AFFINE; the reference default 50 - c, train_cdt.py:139 70 - c) or 1/(c + b)
 * (OSRL_COST_RECIPROCAL; train_cdt.py:139 1/(c + 10)); cdf (optional) = its running sum, for
 * osrl_seq_window_gather. */
enum { OSRL_COST_AFFINE = 0, OSRL_COST_RECIPROCAL = 1 };
int osrl_cost_sample_prob(const float* cost_returns, const int64_t* ep_start, int32_t n_episodes, int32_t kind,
                          float a, float b, float* prob, float* cdf, void* stream);
/* compute_start_index_sample_prob (dataset.py:472-494): per trajectory, p[i] ~ (costs (*) gauss_kernel(10,10))[i] + x
 * with x balancing cost / no-cost steps for the target proportion `prob`; p_out / cdf_out are [total rows] laid out
 * like the trajectory tables (either may be NULL); cdf_out feeds osrl_seq_window_gather's start_cdf. */
int osrl_start_index_prob(const float* costs, const int64_t* ep_start, const int32_t* ep_len, int32_t n_episodes,
                          double prob, float* p_out, float* cdf_out, void* stream);
/* process_bc_dataset's selection (dataset.py:108-124) as a stable compaction: idx[0..*n_keep) = kept transition
 * indices in order.  ALL: every one; SAFE: cr <= t0; RISKY: cr >= t0; BOUNDARY: t0 < cr <= t1 (the caller passes
 * the thresholds cost_limit, 2 x cost_limit, (0.5, 1.5) x cost_limit rounded to fp32 as numpy does). */
enum { OSRL_BC_ALL = 0, OSRL_BC_SAFE = 1, OSRL_BC_RISKY = 2, OSRL_BC_BOUNDARY = 3 };
int osrl_bc_select(const float* cost_returns, int64_t n, int32_t mode, float t0, float t1, int64_t* idx,
                   int32_t* n_keep, int32_t* ws, void* stream);
/* dst[j, :width] = src[idx[j], :width]; extra (optional): dst[j, width] = extra[idx[j]] (BC multi-task appends the
 * cost return to the observation, dataset.py:128-130). */
int osrl_gather_rows(const float* src, int32_t width, const int64_t* idx, int64_t n_rows, float* dst, int32_t dst_ld,
                     const float* extra, void* stream);

/* ---- batched on-device evaluation ----
 * The reference's Trainer.rollout (cpq.py:330-347, bcql.py:323-340, bc.py:125-145) steps ONE gym env per policy
 * call and crosses host<->device every env step.  No gym env exists in either container, so the build owns a
 * synthetic safe environment (osrl_amd/common/synthetic_env.py: s' = A s + Bm clip(a), reward = 1 - 0.1|s'-goal|^2,
 * cost = 1[s'.w > threshold]); this is its vectorised device step: one row per episode.
 *   state[E, state_dim]  in/out;  obs[E, obs_ld] receives s' in its first state_dim columns (the policy's input
 *   buffer; extra columns, e.g. BC multi-task's cost_limit, are left alone);
 *   acc[E,4] = {return, cost*cost_scale, length, done}: accumulated here, done latches at episode_len and freezes
 *   the episode (state, obs and totals stop changing). */
typedef struct {
  const float* At;   /* A transposed:  At[j*state_dim + i] = A[i][j]   (coalesced per-output reads) */
  const float* Bt;   /* Bm transposed: Bt[k*state_dim + i] = Bm[i][k] */
  const float* w;    /* [state_dim] cost half-space normal */
  const float* goal; /* [state_dim] */
  int32_t state_dim, action_dim, episode_len, pad_;
  float max_action, cost_threshold, cost_scale, pad2_;
} osrl_env_t;
int osrl_env_step(const osrl_env_t* env, const float* act, float* state, float* obs, int32_t obs_ld, float* acc,
                  float* step_out /* optional [E,2]: this step's (reward, raw cost) */, int32_t episodes, void* stream);

/* CDTTrainer.rollout (cdt.py:436-518) with E episodes as batch rows.  The reference keeps the whole history and
 * slices its last seq_len steps per env step; here the CDT engine's [E, seq_len] batch buffers are the window
 * itself: left-aligned and growing (mask = 1 on the filled prefix) until full, then sliding by one per env step.
 * cursor: device int32 = env steps taken so far (identical for all episodes).
 *   pick: act[e] = clamp(head[e, last filled position, :action_dim], +-max_action)     (cdt.py:489-493)
 *   push (after osrl_env_step): actions[last] = act; append (obs', returns - reward, costs_to_go - cost*cost_scale
 *         [1 - cost under cost_reverse], time step + 1, zero dummy action) (cdt.py:496-508); ++cursor.  The pick at
 *         cursor == episode_len reads the last window; callers stop there. */
int osrl_cdt_rollout_pick(const float* head, int32_t head_width, int32_t action_dim, int32_t episodes,
                          int32_t seq_len, const int32_t* cursor, float max_action, float* act, void* stream);
int osrl_cdt_rollout_push(float* states, float* actions, float* returns, float* costs_to_go, int64_t* time_steps,
                          float* mask, int32_t episodes, int32_t seq_len, int32_t state_dim, int32_t action_dim,
                          const float* act, const float* obs, int32_t obs_ld, const float* step_out, float cost_scale,
                          int32_t cost_reverse, int32_t* cursor, int32_t episode_len /* no-op once cursor reaches it */,
                          void* stream);

/* library identity */
/* ---- B = 1 .. OSRL_POLICY_MAX_ROWS act() latency path (act.hip) -----------------------------------------------
 * Replaces one `model.act(obs)` of the reference's episode loop (XTrainer.rollout, cpq.py:330-347): H2D copy of the
 * observation + 3-6 aten kernels + two D2H syncs (cpq.py:240-252, bcql.py:236-243, bc.py:66-76) become ONE launch:
 * the observation rows are read from a pinned host buffer mapped into the device, the whole policy (1-2 chained MLPs
 * as GEMVs over the packed forward weights + the distribution head) runs in one workgroup, action / log-prob land
 * in pinned memory and the host spins on a sequence number.  THE ONE EXCEPTION to this header's conventions: a policy
 * handle owns a small pinned host allocation (osrl_policy_create / _destroy) and osrl_policy_act RETURNS AFTER the
 * kernel has published its results (it is the synchronous call the episode loop needs).  Wf / b are DEVICE pointers. */
#define OSRL_POLICY_MAX_ROWS 4
enum { OSRL_POLICY_MLP = 0, OSRL_POLICY_GAUSS = 1, OSRL_POLICY_BCQ = 2 };
typedef struct {
  int32_t n_layers;
  int32_t dims[OSRL_MAX_LAYERS + 1];
  int32_t acts[OSRL_MAX_LAYERS];
  float out_scale;                  /* net output = out_scale * act(z) */
  const float* Wf[OSRL_MAX_LAYERS]; /* PACKED forward weights PF[k/4][n][k%4] (osrl_pack_weights / the fused optimizer
                                     * step keep them in step with the canonical parameters) */
  const float* b[OSRL_MAX_LAYERS];  /* canonical bias [dims[l+1]] */
} osrl_gemv_net_t;
typedef struct {
  int32_t kind;       /* MLP:   a = net0(obs)                               (BC, MLPActor net.py:65-85)
                       * GAUSS: (mu | log_std) = net0(obs); a = max_action * tanh(mu + exp(clamp(ls)) * eps), logp
                       *                                                    (SquashedGaussianMLPActor net.py:152-205)
                       * BCQ:   a0 = net0([obs, clamp(z, +-0.5)]) (VAE.decode net.py:331-339), t = net1([obs, a0]),
                       *        a = clamp(a0 + phi * max_action * t, +-max_action)       (net.py:33-62, bcql.py:236-243) */
  int32_t obs_dim, act_dim, latent_dim;
  float max_action, phi;
  osrl_gemv_net_t net[2];
} osrl_policy_t;
int osrl_policy_create(const osrl_policy_t* desc, void** handle);
/* HOST pointers into the handle's pinned block: obs [MAX_ROWS, obs_dim] (caller writes), noise [MAX_ROWS, act_dim |
 * latent_dim] (caller writes when host_noise = 1), act [MAX_ROWS, act_dim] and logp [MAX_ROWS] (kernel writes). */
int osrl_policy_io(void* handle, float** obs, float** noise, float** act, float** logp);
/* rows <= OSRL_POLICY_MAX_ROWS.  deterministic: eps = 0 (GAUSS).  host_noise: read eps / z from the noise block
 * instead of drawing it in the kernel (Philox keyed by `seed` and the handle's call counter). */
int osrl_policy_act(void* handle, int32_t rows, int32_t deterministic, int32_t host_noise, uint64_t seed, void* stream);
int osrl_policy_destroy(void* handle);

/* ---- data-parallel exchanges through IPC-mapped device buffers (ipc.hip, round 6; nothing to mirror in the reference: it
 * has no distributed code).  What the ranks of a data-parallel step exchange (flat
 * gradients per optimizer phase, CPQ's KL values, statistics) is latency-bound; a collective-library launch costs 16-19 us
 * even on one rank.  Here every rank owns a published buffer (2 halves of half_floats) and a 64-byte control block in ITS
 * device memory (osrl_ipc_alloc), hands the two handles to its peers by any means (the engines use torch.distributed's
 * object all-gather, once) and maps theirs (osrl_ipc_open).  An exchange is ONE asynchronous, hipGraph-capturable kernel
 * launch per rank: publish -> flag -> wait for every rank's flag -> sum in RANK ORDER (replicas stay bit-identical) or
 * gather.  Every rank must issue the same sequence of exchanges on ONE stream; a peer that never arrives makes the launch
 * give up after ~2 s and set the error word (osrl_ipc_status), it does not hang the device. */
#define OSRL_IPC_MAX_WORLD 8
#define OSRL_IPC_MAX_SEG 8
#define OSRL_IPC_HANDLE_BYTES 64
typedef struct {
  int32_t world, rank;
  float* pub[OSRL_IPC_MAX_WORLD];    /* published buffers, [rank] = this process's own allocation, the others mapped */
  uint32_t* ctl[OSRL_IPC_MAX_WORLD]; /* control blocks (>= 64 bytes, zeroed): flag | arrivals | error | exchanges done */
  int64_t half_floats;               /* floats per published half (a message must fit; multiple of 4) */
} osrl_ipc_t;
/* hipMalloc + zero + hipIpcGetMemHandle / hipIpcOpenMemHandle / Close / hipFree (setup time, synchronous). */
int osrl_ipc_alloc(int64_t bytes, void** dev_ptr, void* handle64);
int osrl_ipc_open(const void* handle64, void** dev_ptr);
int osrl_ipc_close(void* mapped_ptr);
int osrl_ipc_free(void* dev_ptr);
/* bufs[i][0 .. lens[i]) += the same ranges of every other rank (in place, n_bufs <= OSRL_IPC_MAX_SEG, together <=
 * half_floats: OSRL_E_UNSUPPORTED otherwise) -- dist.all_reduce(SUM) of several tensors as one launch. */
int osrl_ipc_all_reduce(const osrl_ipc_t* x, float* const* bufs, const int64_t* lens, int32_t n_bufs, void* stream);
/* The same where bufs[i] is slab 0 of n_slabs[i] split-K gradient slabs slab_strides[i] floats apart (what the dW kernels
 * leave): the rank's own slab sum -- slab order, the bits of osrl_reduce_slabs -- is formed while publishing, and slab 0
 * receives the sum over slabs AND ranks: osrl_reduce_slabs + osrl_ipc_all_reduce in one launch. */
int osrl_ipc_all_reduce_slabs(const osrl_ipc_t* x, float* const* bufs, const int64_t* lens, const int32_t* n_slabs,
                              const int64_t* slab_strides, int32_t n_bufs, void* stream);
/* dst[r * n + i] = rank r's src[i] -- dist.all_gather_into_tensor. */
int osrl_ipc_all_gather(const osrl_ipc_t* x, const float* src, int64_t n, float* dst, void* stream);
/* words4 = this rank's control words (flag, arrivals, error: 0 = fine, 1 + r = rank r never arrived, exchanges done).
 * Synchronous copy: for the points where the host synchronises anyway. */
int osrl_ipc_status(const osrl_ipc_t* x, uint32_t* words4);

/* Diagnostics (diag.hip): where the HIP runtime of this process keeps kernel arguments -- *where = 1 device memory,
 * 0 host memory (every wave then fetches its launch arguments over PCIe; see osrl_args_begin), -1 unknown.  dev_scratch:
 * 8 bytes of device memory.  Synchronises `stream`; not for the hot path. */
int osrl_kernarg_probe(uint64_t* dev_scratch, int32_t* where, uint64_t* address, void* stream);
/* (new, diagnostics) out[0] = the device's constant 100 MHz real-time counter when this one-lane launch executes.
 * Asynchronous, hipGraph-capturable: two of them around a launch of a captured step give that launch's duration inside
 * the replayed graph (bench.py `roofline.frac`; torch has no counterpart: its events cannot be timed inside a capture). */
int osrl_stamp_realtime(uint64_t* out, void* stream);

const char* osrl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OSRL_AMD_H */
