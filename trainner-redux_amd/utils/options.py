"""YAML -> options for the hot path, accepting the reference's `options/*.yml` keys.

Mirrors the *API* of /root/reference/traiNNer/utils/redux_options.py (section and key names, defaults,
strict rejection of unknown keys: `forbid_unknown_fields`, :8) and the post-processing of
/root/reference/traiNNer/utils/options.py (:181-182 `num_gpu: auto`, :185-189 dataset phase from the key
prefix, scale copied into each dataset) without msgspec (absent here).  Keys that belong to subsystems
outside SURVEY.md section 8 (OTF degradations, ONNX, automations) are accepted and carried, not interpreted.
"""
from __future__ import annotations

import copy
from typing import Any

import yaml

_REQ = object()


class Section:
    """Attribute-style options section with a closed key set."""

    _fields: dict[str, Any] = {}
    _name = "options"

    def __init__(self, data: dict[str, Any] | None = None) -> None:
        data = dict(data or {})
        unknown = [k for k in data if k not in self._fields]
        if unknown:
            raise ValueError(f"Object contains unknown field `{unknown[0]}` - at `$.{self._name}`")
        for k, default in self._fields.items():
            if k in data:
                setattr(self, k, data[k])
            elif default is _REQ:
                raise ValueError(f"Object missing required field `{k}` - at `$.{self._name}`")
            else:
                setattr(self, k, copy.deepcopy(default))

    def to_dict(self) -> dict[str, Any]:
        out = {}
        for k in self._fields:
            v = getattr(self, k)
            out[k] = v.to_dict() if isinstance(v, Section) else v
        return out

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.to_dict()})"


_KERNELS = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
_KPROB = [0.45, 0.25, 0.12, 0.03, 0.12, 0.03]


class DatasetOptions(Section):
    _name = "datasets"
    _fields = dict(
        name=_REQ, type=_REQ, io_backend={"type": "disk"}, num_worker_per_gpu=None, batch_size_per_gpu=None, accum_iter=1,
        use_hflip=True, use_rot=True, mean=None, std=None, gt_size=None, lq_size=None, color=None, phase=None, scale=None,
        dataset_enlarge_ratio="auto", prefetch_mode=None, pin_memory=True, persistent_workers=True, num_prefetch_queue=1,
        prefetch_factor=2, clip_size=None, dataroot_gt=None, dataroot_lq=None, meta_info=None, filename_tmpl="{}",
        blur_kernel_size=12, kernel_list=_KERNELS, kernel_prob=_KPROB, kernel_range=(5, 17), sinc_prob=0, blur_sigma=(0.2, 2),
        betag_range=(0.5, 4), betap_range=(1, 2), blur_kernel_size2=12, kernel_list2=_KERNELS, kernel_prob2=_KPROB,
        kernel_range2=(5, 17), sinc_prob2=0, blur_sigma2=(0.2, 1), betag_range2=(0.5, 4), betap_range2=(1, 2), final_sinc_prob=0,
        final_kernel_range=(5, 17),
    )


class PathOptions(Section):
    _name = "path"
    _fields = dict(
        experiments_root=None, models=None, resume_models=None, training_states=None, log=None, visualization=None, results_root=None,
        pretrain_network_g=None, pretrain_network_g_path=None, pretrain_network_ae_path=None, param_key_g=None, strict_load_g=True,
        resume_state=None, pretrain_network_g_ema=None, pretrain_network_d=None, param_key_d=None, strict_load_d=True,
        pretrain_network_ae=None, pretrain_network_ae_ema=None, pretrain_network_ae_decoder=None, pretrain_network_ae_decoder_ema=None,
        ignore_resume_networks=None,
    )


_LEGACY_LOSS_OPTS = ["pixel_opt", "mssim_opt", "ms_ssim_l1_opt", "perceptual_opt", "contextual_opt", "dists_opt", "hr_inversion_opt",
                     "dinov2_opt", "topiq_opt", "pd_opt", "fd_opt", "ldl_opt", "hsluv_opt", "gan_opt", "color_opt", "luma_opt", "avg_opt",
                     "bicubic_opt"]


class TrainOptions(Section):
    _name = "train"
    _fields = dict(
        total_iter=_REQ, adaptive_d=False, adaptive_d_ema_decay=0.999, adaptive_d_threshold=1.02, optim_g=None, ema_decay=0,
        ema_switch_iter=0, ema_update_after_step=0, ema_power=10, grad_clip=False, warmup_iter=-1, scheduler=None, optim_d=None,
        optim_ae=None, losses=None, **{k: None for k in _LEGACY_LOSS_OPTS}, use_moa=False, moa_augs=["none", "mixup", "cutmix", "resizemix", "cutblur"],
        moa_probs=[0.4, 0.084, 0.084, 0.084, 0.348], moa_debug=False, moa_debug_limit=100, dynamic_loss_scheduling=None,
        training_automations=None,
    )
    legacy_loss_opts = _LEGACY_LOSS_OPTS


class ValOptions(Section):
    _name = "val"
    _fields = dict(val_enabled=_REQ, save_img=_REQ, tile_size=0, tile_overlap=0, val_freq=None, suffix=None, metrics_enabled=False,
                   metrics=None, pbar=True)


class WandbOptions(Section):
    _name = "logger.wandb"
    _fields = dict(resume_id=None, project=None)


class LogOptions(Section):
    _name = "logger"
    _fields = dict(print_freq=_REQ, save_checkpoint_freq=_REQ, use_tb_logger=_REQ, save_checkpoint_format="safetensors", wandb=None)


# top-level keys of subsystems outside the hot path: accepted verbatim (names from redux_options.py:503-900)
_CARRIED = """high_order_degradation high_order_degradations_debug high_order_degradations_debug_limit p_clean webp_prob webp_range
avif_prob avif_range oversharpen_prob oversharpen_strength chromatic_aberration_prob demosaic_prob aliasing_prob aliasing_scale_range
motion_blur_prob motion_blur_kernel_size motion_blur_angle_range lens_distort_prob lens_distort_strength_range exposure_prob
exposure_factor_range color_temp_prob color_temp_shift_range sensor_noise_prob sensor_noise_std_range rolling_shutter_prob
rolling_shutter_strength_range heif_prob heif_range enable_sequences editing_prob editing_exposure_prob editing_exposure_range
editing_oversharpen_prob editing_oversharpen_strength sequence_probability predefined_sequences dataroot_lq_prob lq_usm
lq_usm_radius_range blur_prob thicklines_prob resize_prob resize_mode_list resize_mode_prob resize_range gaussian_noise_prob noise_range
poisson_scale_range gray_noise_prob jpeg_prob jpeg_range blur_prob2 resize_prob2 resize_mode_list2 resize_mode_prob2 resize_range2
gaussian_noise_prob2 noise_range2 poisson_scale_range2 gray_noise_prob2 jpeg_prob2 jpeg_range2 resize_mode_list3 resize_mode_prob3
compression_formats compression_weights compression_jpeg_range compression_webp_range compression_avif_range compression_heif_range
recompression_prob recompression_formats recompression_weights queue_size onnx auto_vram_management dist_params contents""".split()


class ReduxOptions(Section):
    _name = ""
    _fields = dict(
        name=_REQ, scale=_REQ, num_gpu=_REQ, path=_REQ, input_pixel_format="rgb", output_pixel_format="rgb", network_g=None,
        network_d=None, network_ae=None, manual_seed=None, deterministic=False, dist=None, launcher=None, rank=None, world_size=None,
        auto_resume=None, watch=False, start_iter=0, is_train=None, root_path=None, switch_iter_per_epoch=1, use_amp=False,
        amp_bf16=False, use_channels_last=True, fast_matmul=False, use_compile=False, compile_mode="max-autotune-no-cudagraphs",
        detect_anomaly=False, datasets={}, train=None, val=None, logger=None, find_unused_parameters=False,
        **{k: None for k in _CARRIED},
    )

    def __init__(self, data: dict[str, Any]) -> None:
        super().__init__(data)
        self.path = PathOptions(self.path if isinstance(self.path, dict) else {})
        self.datasets = {k: DatasetOptions(v) for k, v in (self.datasets or {}).items()}
        if self.train is not None:
            self.train = TrainOptions(self.train)
        if self.val is not None:
            self.val = ValOptions(self.val)
        if self.logger is not None:
            lg = LogOptions(self.logger)
            if isinstance(lg.wandb, dict):
                lg.wandb = WandbOptions(lg.wandb)
            self.logger = lg


def parse_options_dict(data: dict[str, Any], is_train: bool = True) -> ReduxOptions:
    opt = ReduxOptions(data)
    opt.is_train = is_train
    if opt.num_gpu == "auto":  # options.py:181-182
        import torch

        opt.num_gpu = torch.cuda.device_count()
    for key, ds in opt.datasets.items():  # options.py:185-189
        ds.phase = key.split("_")[0]
        ds.scale = opt.scale
    return opt


def load_options(path: str, is_train: bool = True) -> ReduxOptions:
    with open(path, encoding="utf-8") as f:
        data = yaml.safe_load(f)
    if not isinstance(data, dict):
        raise ValueError(f"{path}: top level must be a mapping")
    return parse_options_dict(data, is_train)
