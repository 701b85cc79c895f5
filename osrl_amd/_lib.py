"""ctypes binding of libosrl_amd.so (the C ABI declared in include/osrl_amd.h).

The library is the product path: if it is missing (and cannot be built because
hipcc is absent) every op raises -- there is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch bundles its own libamdhip64 / libhsa-runtime64; it must be loaded BEFORE libosrl_amd.so so
# that both share ONE HIP runtime (streams created by torch are then valid handles for our launches).
import torch  # noqa: F401  (import order matters)

MAX_LAYERS = 4
MAX_NETS = 8
MAX_WIDTH = 448
MAX_FIELDS = 8

ACT_ID, ACT_RELU, ACT_TANH = 0, 1, 2
MAP_ID, MAP_MOD, MAP_DIV = 0, 1, 2
ACT_CODES = {"id": ACT_ID, "identity": ACT_ID, "relu": ACT_RELU, "tanh": ACT_TANH}

_fp = C.c_void_p  # device float*


class MlpT(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_nets", C.c_int32),
                ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("acts", C.c_int32 * MAX_LAYERS),
                ("out_scale", C.c_float), ("tile_rows", C.c_int32), ("wg_cap", C.c_int32),
                ("Wf", (_fp * MAX_LAYERS) * MAX_NETS), ("Wb", (_fp * MAX_LAYERS) * MAX_NETS),
                ("b", (_fp * MAX_LAYERS) * MAX_NETS)]


class RowsT(C.Structure):
    _fields_ = [("rows", C.c_int32), ("d0", C.c_int32), ("map0", C.c_int32), ("div0", C.c_int32),
                ("d1", C.c_int32), ("map1", C.c_int32), ("div1", C.c_int32),
                ("src0", _fp), ("src1", _fp),
                ("row_list", C.c_void_p), ("n_rows_dev", C.c_void_p),  # (a device-chosen row set: include/osrl_amd.h)
                ("share0", C.c_int32), ("share_k16", C.c_int32)]  # (tiles of shared src0 rows: include/osrl_amd.h)


class ActsT(C.Structure):
    _fields_ = [("x", _fp), ("h", (_fp * MAX_LAYERS) * MAX_NETS)]


class GradsT(C.Structure):
    _fields_ = [("dy", _fp * MAX_NETS), ("dz", (_fp * MAX_LAYERS) * MAX_NETS), ("dx", _fp * MAX_NETS),
                ("dx_col0", C.c_int32), ("dx_cols", C.c_int32)]


class TailT(C.Structure):  # osrl_mlp_tail_t
    _fields_ = [("kind", C.c_int32), ("L", C.c_int32), ("eps", _fp), ("head", _fp), ("out", _fp),
                ("beta", C.c_float), ("rows_global", C.c_int32), ("inv_rows_", C.c_float),
                ("max_action", C.c_float), ("eps2", _fp), ("out2", _fp), ("tanh2", _fp), ("eps_ood", _fp),
                ("out_ood", _fp), ("n_samples", C.c_int32), ("pad_", C.c_int32)]


TAIL_NONE, TAIL_VAE_LATENT, TAIL_VAE_LATENT_BWD, TAIL_GAUSS, TAIL_VAE_KL = 0, 1, 2, 3, 4


class SeedT(C.Structure):  # osrl_mlp_seed_t
    _fields_ = [("kind", C.c_int32), ("n_a", C.c_int32), ("n_b", C.c_int32), ("rows_global", C.c_int32),
                ("a", _fp), ("b", _fp), ("x0", _fp), ("x1", _fp), ("eps", _fp), ("tanh_u", _fp), ("kl_head", _fp),
                ("kl_L", C.c_int32), ("n_samples", C.c_int32), ("gamma", C.c_float), ("thres", C.c_float),
                ("scale", C.c_float), ("max_action", C.c_float), ("stat_scale", C.c_float),
                ("stat_scale2", C.c_float), ("kl_beta", C.c_float), ("pad2_", C.c_float), ("partials", _fp),
                ("counter", C.c_void_p), ("stat", _fp)]


SEED_NONE, SEED_MSE, SEED_CPQ_CRITIC, SEED_CPQ_COST, SEED_CPQ_ACTOR, SEED_GAUSS_HEAD, SEED_BCQ_CRITIC = 0, 1, 2, 3, 4, 5, 6


class VaeNsT(C.Structure):  # osrl_vae_ns_t
    _fields_ = [("enc", C.POINTER(MlpT)), ("dec", C.POINTER(MlpT)), ("rows", C.c_int32), ("od", C.c_int32),
                ("ad", C.c_int32), ("L", C.c_int32), ("rows_global", C.c_int32), ("beta", C.c_float),
                ("obs", _fp), ("act", _fp), ("eps", _fp), ("enc_acts", ActsT), ("dec_acts", ActsT), ("z", _fp),
                ("enc_g", GradsT), ("dec_g", GradsT), ("P", _fp), ("slabs", _fp), ("partials", _fp),
                ("counter", C.c_void_p), ("stat", _fp)]


class DwEntryT(C.Structure):
    _fields_ = [("dz", _fp), ("a", _fp), ("w_off", C.c_int64), ("b_off", C.c_int64),
                ("out", C.c_int32), ("in_", C.c_int32), ("ldz", C.c_int32), ("lda", C.c_int32)]


class DwAdamT(C.Structure):  # osrl_dw_adam_t
    _fields_ = [("p", _fp), ("m", _fp), ("v", _fp), ("tgt", _fp), ("map_f", C.c_void_p), ("map_b", C.c_void_p),
                ("pf", _fp), ("pb", _fp), ("tf", _fp), ("st", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("tau", C.c_float), ("pad_", C.c_int32)]


STEP_MAX_WG = 128
STEP_WS = STEP_MAX_WG + 8  # floats of scratch of osrl_mlp_regress_step (include/osrl_amd.h OSRL_STEP_WS)
E_UNSUPPORTED = -2


class MlpStepT(C.Structure):  # osrl_mlp_step_t
    _fields_ = [("st", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float), ("warmup", C.c_int32),
                ("n_stats", C.c_int32), ("ring_len", C.c_int32), ("n_fields", C.c_int32),
                ("stats_cur", _fp), ("ring", _fp), ("src", _fp * 8), ("dst", _fp * 8), ("width", C.c_int32 * 8),
                ("scale", C.c_float * 8), ("n_rows", C.c_int64), ("gather_seed", C.c_uint64),
                ("gather_stream", C.c_uint32), ("pad0_", C.c_uint32),
                ("net", MlpT), ("in_", RowsT), ("acts", ActsT), ("grads", GradsT),
                ("target", _fp), ("n_global", C.c_int64), ("stat", _fp),
                ("entries", C.c_void_p), ("work", C.c_void_p), ("n_work", C.c_int32), ("tile_blocks", C.c_int32),
                ("p", _fp), ("m", _fp), ("v", _fp), ("map_f", C.c_void_p), ("map_b", C.c_void_p), ("pf", _fp),
                ("pb", _fp), ("lr", C.c_float), ("eps", C.c_float), ("ws", _fp)]


class PackEntryT(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("f_off", C.c_int64), ("b_off", C.c_int64),
                ("out", C.c_int32), ("in_", C.c_int32)]


class DropoutT(C.Structure):
    _fields_ = [("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_uint64), ("st", C.c_void_p)]


IPC_MAX_WORLD, IPC_MAX_SEG, IPC_HANDLE_BYTES = 8, 8, 64  # include/osrl_amd.h OSRL_IPC_*


class IpcT(C.Structure):  # osrl_ipc_t
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("pub", C.c_void_p * IPC_MAX_WORLD),
                ("ctl", C.c_void_p * IPC_MAX_WORLD), ("half_floats", C.c_int64)]


class EnvT(C.Structure):
    _fields_ = [("At", C.c_void_p), ("Bt", C.c_void_p), ("w", C.c_void_p), ("goal", C.c_void_p),
                ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("episode_len", C.c_int32), ("pad_", C.c_int32),
                ("max_action", C.c_float), ("cost_threshold", C.c_float), ("cost_scale", C.c_float),
                ("pad2_", C.c_float)]


class GemvNetT(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("acts", C.c_int32 * MAX_LAYERS),
                ("out_scale", C.c_float), ("Wf", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS)]


class PolicyT(C.Structure):
    _fields_ = [("kind", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("latent_dim", C.c_int32),
                ("max_action", C.c_float), ("phi", C.c_float), ("net", GemvNetT * 2)]


POLICY_MLP, POLICY_GAUSS, POLICY_BCQ = 0, 1, 2
POLICY_MAX_ROWS = 4


class StepStateT(C.Structure):
    _fields_ = [("step", C.c_int64), ("bc1", C.c_float), ("bc2_sqrt", C.c_float),
                ("lr_scale", C.c_float), ("arrive_", C.c_uint32)]


_i32, _i64, _f32, _u32, _u64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_uint64, C.c_void_p
_P = C.POINTER

# name -> argtypes  (all return int except osrl_version); must match include/osrl_amd.h
PROTOTYPES = {
    "osrl_mlp_forward": [_P(MlpT), _P(RowsT), _P(ActsT), _vp],
    "osrl_env_step": [_P(EnvT), _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp],
    "osrl_cdt_rollout_pick": [_vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp],
    "osrl_cdt_rollout_push": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _f32, _i32,
                              _vp, _i32, _vp],
    "osrl_bear_mmd": [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp],
    "osrl_bear_actor_sums": [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "osrl_bear_actor_loss": [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _i64,
                             _i32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "osrl_bear_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "osrl_dice_optimal_w": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp],
    "osrl_dice_chi_ell": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp],
    "osrl_dice_chi_step": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                           _i32, _i32, _f32, _vp, _vp],
    "osrl_dice_nu_step": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp,
                          _vp, _vp, _vp, _vp],
    "osrl_dice_perturb": [_vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp],
    "osrl_dice_actor_loss": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "osrl_ingest_ws_elems": [_i64],
    "osrl_episode_segments": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "osrl_episode_returns": [_vp, _vp, _vp, _i32, _f32, _i32, _i32, _vp, _vp, _vp],
    "osrl_policy_create": [_P(PolicyT), _P(C.c_void_p)],
    "osrl_policy_io": [_vp, _P(_P(C.c_float)), _P(_P(C.c_float)), _P(_P(C.c_float)), _P(_P(C.c_float))],
    "osrl_policy_act": [_vp, _i32, _i32, _i32, _u64, _vp],
    "osrl_policy_destroy": [_vp],
    "osrl_cost_sample_prob": [_vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp],
    "osrl_start_index_prob": [_vp, _vp, _vp, _i32, C.c_double, _vp, _vp, _vp],
    "osrl_bc_select": [_vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp],
    "osrl_gather_rows": [_vp, _i32, _vp, _i64, _vp, _i32, _vp, _vp],
    "osrl_vae_ns_supported": [_P(VaeNsT)],
    "osrl_vae_ns_forward": [_P(VaeNsT), _vp],
    "osrl_vae_ns_backward": [_P(VaeNsT), _vp],
    "osrl_mlp_forward2": [_P(MlpT), _P(RowsT), _P(ActsT), _P(MlpT), _P(RowsT), _P(ActsT), _vp],
    "osrl_mlp_backward_dz": [_P(MlpT), _i32, _P(ActsT), _P(GradsT), _vp],
    "osrl_mlp_forward_tail": [_P(MlpT), _P(RowsT), _P(ActsT), _P(TailT), _vp],
    "osrl_mlp_forward2_tail": [_P(MlpT), _P(RowsT), _P(ActsT), _P(TailT), _P(MlpT), _P(RowsT), _P(ActsT), _P(TailT), _vp],
    "osrl_mlp_backward_dz_tail": [_P(MlpT), _i32, _P(ActsT), _P(GradsT), _P(TailT), _vp],
    "osrl_mlp_backward_dz_seed": [_P(MlpT), _i32, _P(ActsT), _P(GradsT), _P(TailT), _P(SeedT), _vp],
    "osrl_linear": [_fp, _i64, _i32, _i32, _fp, _i32, _i32, _i32, _fp, _fp, _i64, _fp, _i64, _vp],
    "osrl_pack_weights": [_fp, _fp, _fp, _vp, _i32, _i32, _vp],
    "osrl_mlp_backward_dw": [_vp, _vp, _i32, _i32, _i32, _fp, _i64, _vp],
    "osrl_mlp_backward_dw_tiles": [_vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp],
    "osrl_mlp_backward_dw_big": [_vp, _vp, _i32, _i32, _i32, _fp, _i64, _vp],
    "osrl_mlp_backward_dw_coop": [_vp, _vp, _i32, _i32, _i32, _fp, _i64, _vp],
    "osrl_mlp_backward_dw_tiles_adam": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _P(DwAdamT), _vp],
    "osrl_step_tick": [_vp, _f32, _f32, _i32, _fp, _fp, _i32, _i32, _vp],
    "osrl_step_begin": [_vp, _f32, _f32, _i32, _fp, _fp, _i32, _i32, _fp, _i64, _u64, _u32, _i32, _P(_fp), _P(_fp),
                        _P(_i32), _P(_f32), _i64, _i32, _u64, _u32, _vp],
    "osrl_step_tick_peer": [_vp, _vp, _f32, _f32, _i32, _fp, _fp, _i32, _i32, _vp],
    "osrl_step_begin_peer": [_vp, _vp, _f32, _f32, _i32, _fp, _fp, _i32, _i32, _fp, _i64, _u64, _u32, _i32, _P(_fp), _P(_fp),
                             _P(_i32), _P(_f32), _i64, _i32, _u64, _u32, _vp],
    "osrl_adam_step": [_fp, _fp, _fp, _fp, _fp, _i32, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _fp,
                       _vp, _vp],
    "osrl_adam_step_packed": [_fp, _fp, _fp, _fp, _fp, _i32, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _fp,
                              _vp, _vp, _vp, _fp, _fp, _fp, _vp],
    "osrl_polyak": [_fp, _fp, _i64, _f32, _vp, _fp, _vp],
    "osrl_reduce_slabs": [_fp, _fp, _i32, _i64, _i64, _vp],
    "osrl_reduce_slabs_counts": [_fp, _fp, _vp, _i64, _i64, _vp],
    "osrl_layernorm_param_reduce": [_fp, _i64, _i32, _i32, _i32, _fp, _vp, _vp, _vp],
    "osrl_randn_fill": [_fp, _i64, _u64, _u32, _vp, _vp],
    "osrl_replay_gather": [_i32, _P(_fp), _P(_fp), _P(_i32), _P(_f32), _i64, _i32, _vp, _u64, _u32, _vp, _vp],
    "osrl_seq_window_gather": [_fp, _fp, _fp, _fp, _fp, _vp, _vp, _fp, _fp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _fp, _fp,
                               _fp, _fp, _vp, _fp, _fp, _fp, _vp, _u64, _u32, _vp, _vp],
    "osrl_gauss_head": [_fp, _fp, _i32, _i32, _f32, _fp, _fp, _fp, _vp],
    "osrl_gauss_head_bwd": [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _f32, _fp, _vp],
    "osrl_gauss_ood_sample": [_fp, _fp, _i32, _i32, _i32, _fp, _vp],
    "osrl_vae_latent": [_fp, _fp, _i32, _i32, _fp, _vp],
    "osrl_vae_loss": [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _i32, _fp, _fp, _vp],
    "osrl_vae_latent_bwd": [_fp, _fp, _fp, _i32, _i32, _f32, _i32, _fp, _vp],
    "osrl_vae_kl_rows": [_fp, _i32, _i32, _fp, _vp],
    "osrl_quantile": [_fp, _i64, _f32, _fp, _vp],
    "osrl_quantile_ws": [_fp, _i64, _f32, _vp, _fp, _vp],
    "osrl_cpq_critic_loss": [_fp, _i32, _fp, _i32, _fp, _i32, _fp, _fp, _i32, _f32, _f32, _i32, _fp, _fp, _vp],
    "osrl_cpq_ood_mean": [_fp, _i32, _fp, _fp, _i32, _i32, _i32, _fp, _vp],
    "osrl_cpq_ood_stat": [_fp, _i32, _fp, _f32, _i32, _i32, _i32, _fp, _fp, _vp],
    "osrl_cpq_ood_select": [_fp, _fp, _f32, _i32, _fp, _vp, _vp, _vp],
    "osrl_cpq_ood_sum": [_fp, _i32, _i32, _vp, _f32, _fp, _vp],
    "osrl_cpq_cost_loss": [_fp, _i32, _fp, _i32, _fp, _fp, _i32, _f32, _f32, _f32, _i32, _f32, _fp, _fp, _fp, _vp],
    "osrl_cpq_alpha_step": [_fp, _f32, _f32, _f32, _fp, _fp, _vp],
    "osrl_cpq_cost_loss_ood": [_fp, _i32, _fp, _fp, _i32, _fp, _i32, _fp, _i32, _fp, _fp, _i32, _f32, _f32, _f32, _fp, _fp,
                               _fp, _vp],
    "osrl_cpq_actor_loss": [_fp, _i32, _fp, _i32, _i32, _f32, _i32, _fp, _fp, _vp],
    "osrl_mse_loss": [_fp, _fp, _i64, _i64, _fp, _fp, _vp],
    "osrl_clamp": [_fp, _i64, _f32, _f32, _vp],
    "osrl_bcq_perturb": [_fp, _fp, _i32, _i32, _f32, _f32, _fp, _vp],
    "osrl_bcq_perturb_bwd": [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _f32, _fp, _vp],
    "osrl_bcq_critic_loss": [_fp, _i32, _i32, _i32, _fp, _i32, _fp, _fp, _i32, _f32, _f32, _i32, _fp, _fp, _vp],
    "osrl_bcq_critic_loss_ws": [_fp, _i32, _i32, _i32, _fp, _i32, _fp, _fp, _i32, _f32, _f32, _i32, _fp, _fp, _fp, _vp],
    "osrl_vae_loss_ws": [_fp, _fp, _fp, _i32, _i32, _i32, _f32, _i32, _fp, _fp, _fp, _vp],
    "osrl_bcq_actor_sums": [_fp, _i32, _i32, _fp, _i32, _i32, _i32, _i32, _fp, _vp],
    "osrl_bcq_actor_loss": [_fp, _i32, _i32, _fp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _i32, _fp, _f32, _fp, _fp,
                            _fp, _fp, _vp],
    "osrl_cdt_embed_ln": [_fp, _fp, _fp, _fp, _fp, _vp] + [_fp] * 13 + [_i32] * 9 + [_fp, _fp, _fp, _fp, _vp],
    "osrl_layernorm_fwd": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _vp],
    "osrl_layernorm_bwd": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp, _i64, _i64, _vp],
    "osrl_layernorm_fwd_drop": [_fp, _fp, _P(DropoutT), _fp, _fp, _fp, _fp, _fp, _i32, _i32, _vp],
    "osrl_layernorm_bwd_drop": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _P(DropoutT), _fp, _i32, _i32, _i32, _fp, _i64, _i64, _vp],
    "osrl_attention_fwd": [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _P(DropoutT), _fp, _vp],
    "osrl_attention_bwd": [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _P(DropoutT), _fp, _vp],
    "osrl_attention_keep_bytes": [_i32, _i32, _i32, _i32],
    "osrl_attention_fwd_keep": [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _P(DropoutT), _fp, _vp, _vp],
    "osrl_attention_bwd_keep": [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _P(DropoutT), _fp, _vp, _vp],
    "osrl_dropout": [_fp, _fp, _i64, _P(DropoutT), _vp],
    "osrl_gelu_fwd": [_fp, _fp, _i64, _vp],
    "osrl_gelu_bwd": [_fp, _fp, _fp, _i64, _vp],
    "osrl_cdt_loss": [_fp] * 7 + [_i32] * 6 + [_fp, _f32, _f32, _f32, _i32, _vp, _fp, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _vp],
    "osrl_cdt_mask_counts": [_fp, _i32, _fp, _vp],
    "osrl_cdt_timestep_scatter": [_fp, _vp, _i32, _i32, _i32, _i32, _i32, _fp, _vp],
    "osrl_clip_grad_scale": [_fp, _i64, _f32, _fp, _i32, _fp, _vp],
    "osrl_cdt_temperature_step": [_fp, _fp, _fp, _f32, _f32, _f32, _f32, _f32, _vp, _vp],
    "osrl_kernarg_probe": [_vp, _P(_i32), _P(_u64), _vp],
    "osrl_stamp_realtime": [_vp, _vp],
    "osrl_ipc_alloc": [_i64, _P(C.c_void_p), _vp],
    "osrl_ipc_open": [_vp, _P(C.c_void_p)],
    "osrl_ipc_close": [_vp],
    "osrl_ipc_free": [_vp],
    "osrl_ipc_all_reduce": [_P(IpcT), _P(C.c_void_p), _P(_i64), _i32, _vp],
    "osrl_ipc_all_reduce_slabs": [_P(IpcT), _P(C.c_void_p), _P(_i64), _P(_i32), _P(_i64), _i32, _vp],
    "osrl_ipc_all_gather": [_P(IpcT), _vp, _i64, _vp, _vp],
    "osrl_ipc_status": [_P(IpcT), _P(C.c_uint32)],
    "osrl_mlp_regress_step": [_P(MlpStepT), _vp],
    "osrl_args_begin": [_vp, _vp, _i64, _i64, _i32],
    "osrl_args_end": [_P(_i64), _P(_i32), _P(_i32), _P(_i32)],
}

_LIB: Optional[C.CDLL] = None


LOSS_WS = 132  # floats of scratch for the grid loss kernels (include/osrl_amd.h OSRL_LOSS_WS)
QUANTILE_WS = 1032  # uint32 elements of scratch for osrl_quantile_ws (include/osrl_amd.h OSRL_QUANTILE_WS)
RESTYPES = {"osrl_ingest_ws_elems": C.c_int64, "osrl_attention_keep_bytes": C.c_int64}  # everything else returns int (0 = ok)


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libosrl_amd.so")


def load() -> C.CDLL:
    """Load (building first if the sources are newer and hipcc exists).  Raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    alt = os.environ.get("OSRL_LIB")  # tuning aid: load an alternative build of the same sources (A/B kernel variants)
    try:
        if alt:
            if not os.path.exists(alt):
                raise RuntimeError(f"OSRL_LIB={alt} does not exist")
            path = alt
        else:
            from . import build as _build
            path = _build.build()
    except Exception as e:  # no hipcc on this machine: use the prebuilt library if present
        if not os.path.exists(path):
            raise RuntimeError(
                f"libosrl_amd.so is missing at {path} and could not be built ({e}); "
                "osrl_amd has no CPU fallback -- build it with `python -m osrl_amd.build`") from e
    lib = C.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, C.c_int)
    lib.osrl_version.restype = C.c_char_p
    lib.osrl_version.argtypes = []
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc} "
                           f"({'bad argument' if rc == -1 else 'hipError_t'})")
