"""Built-in copies of the model configurations the DxMI hot path is measured on, as Python data
(the YAML files of the reference are accepted unchanged by dxmi_config.load; these exist so tests,
bench.py and `--config builtin:cifar10_T10` work without the reference checkout).
Values: reference configs/cifar10/T10.yaml:1-59 and configs/cifar10/cifar10.yaml."""
import copy

_UNET_CIFAR = {"_target_": "models.DxMI.unet_small.Model", "resolution": 32, "in_channels": 3, "out_ch": 3, "ch": 128,
               "ch_mult": [1, 2, 2, 2], "num_res_blocks": 2, "attn_resolutions": [16], "dropout": 0.1}
_VALUE_IGEBM = {"_target_": "models.value.TimeIndependentValue",
                "net": {"_target_": "models.modules.IGEBMEncoderV2", "in_chan": 3, "out_chan": 1,
                        "use_spectral_norm": False, "keepdim": False, "out_activation": "linear", "avg_pool_dim": 1,
                        "learn_out_scale": True, "nh": 128}}

CONFIGS = {
    "cifar10_T10": {
        "sampler_net": _UNET_CIFAR,
        "sampler": {"_target_": "models.DxMI.var_sampler.VARSampler", "n_timesteps": 10, "sample_shape": [3, 32, 32],
                    "trainable_beta": "fix_last"},
        "energy": None,
        "value": _VALUE_IGEBM,
        "trainer": {"_target_": "models.DxMI.trainer.DxMI_Trainer", "tau1": 0.1, "tau2": 0.01, "gamma": 1,
                    "use_sampler_beta": True, "time_cost": 0, "adavelreg": 0.99, "entropy_in_value": None,
                    "velocity_in_value": None, "time_cost_sig": True},
        "training": {"sampler_ckpt": "pretrained/cifar10_ddpm/model.ckpt.pth", "value_ckpt": None, "fid_epoch": 1,
                     "n_epochs": 200, "batchsize": 128, "sampling_batchsize": 100, "n_fid_samples": 10000,
                     "n_critic": 1, "n_generator": 1, "lr": 1e-7, "v_lr": 1e-5, "seed": 112233, "log_every": 50,
                     "beta_lr": 1e-5},
        "data": {"name": "cifar10", "data_dir": "datasets"},
    },
}
# BASELINE configs[2] ("DDGAN backbone T=4"): models.ddgan is absent from the reference snapshot
# (SURVEY 2 row 27), so the DDGAN *protocol* (T=4, value_resample) is run on the DDPM backbone.
CONFIGS["cifar10_T4_ddpm_backbone"] = copy.deepcopy(CONFIGS["cifar10_T10"])
CONFIGS["cifar10_T4_ddpm_backbone"]["sampler"]["n_timesteps"] = 4
CONFIGS["cifar10_T4_ddpm_backbone"]["trainer"].update({"value_resample": True})


# EDM backbones (reference configs/imagenet64/T10.yaml, T4.yaml; configs/lsun/T4.yaml): `diffusion:` feeds
# models.cm.script_util.create_model_and_diffusion, `sampler:` feeds models.DxMI.openai_diffusion.OpenAIDiffusion.
_EDM_COMMON = {"sigma_min": 0.002, "sigma_max": 80.0, "num_res_blocks": 3, "num_heads": 4, "num_heads_upsample": -1,
               "num_head_channels": 64, "attention_resolutions": "32,16,8", "channel_mult": "", "dropout": 0.0,
               "use_checkpoint": False, "use_scale_shift_norm": True, "resblock_updown": True, "use_fp16": True,
               "use_new_attention_order": False, "learn_sigma": False, "weight_schedule": "uniform", "distillation": False}
CONFIGS["imagenet64_T10"] = {
    "diffusion": dict(_EDM_COMMON, image_size=64, num_channels=192, class_cond=True),
    "sampler": {"sample_shape": [3, 64, 64], "n_timesteps": 10, "class_cond": True, "num_classes": 1000,
                "trainable_beta": "fix_last", "sigma_min": 0.002, "sigma_max": 80.0},
    "trainer": {"_target_": "models.DxMI.trainer.DxMI_Trainer_Cond", "tau1": 0.1, "tau2": 0.01, "gamma": 1, "n_timesteps": 10,
                "use_sampler_beta": True, "adavelreg": 0.99, "entropy_in_value": None, "velocity_in_value": None,
                "value_grad_clip": True, "time_cost": 0, "skip_sampler_tau": 3, "time_cost_sig": 1},
    "value": _VALUE_IGEBM,
    "training": {"pretrained_path": "pretrained/imagenet64_edm/edm_imagenet64_ema.pt", "value_ckpt": None, "n_iter": 10000,
                 "batchsize": 128, "sampling_batchsize": 100, "n_fid_samples": 5000, "seed": 42, "lr": 1e-8, "v_lr": 1e-5,
                 "beta_lr": 1e-6, "weight_decay": 0.0, "initial_log_loss_scale": 20, "log_every": 20, "fid_every": 100},
    "data": {"name": "imagenet64", "image_size": 64, "class_cond": True, "n_class": 1000},
}
CONFIGS["imagenet64_T4"] = copy.deepcopy(CONFIGS["imagenet64_T10"])
CONFIGS["imagenet64_T4"]["sampler"].update({"n_timesteps": 4, "stochastic_last": True, "rho": 4.0})
CONFIGS["imagenet64_T4"]["trainer"].update({"n_timesteps": 4, "skip_running_last": 1})
del CONFIGS["imagenet64_T4"]["trainer"]["skip_sampler_tau"]        # configs/imagenet64/T4.yaml has no such key (default 0)
CONFIGS["imagenet64_T4"]["training"]["n_iter"] = 100000
CONFIGS["lsun_bedroom_T4"] = {
    # configs/lsun/T4.yaml:1-21: two res blocks per level, additive embedding (no scale-shift norm), unconditional
    "diffusion": dict(_EDM_COMMON, image_size=256, num_channels=256, class_cond=False, num_res_blocks=2,
                      use_scale_shift_norm=False),
    "sampler": {"sample_shape": [3, 256, 256], "n_timesteps": 4, "class_cond": False, "num_classes": None,
                "trainable_beta": "fix_last", "sigma_min": 0.002, "sigma_max": 80.0, "stochastic_last": True, "rho": 4.0},
    # the file's trainer / value name models.GCD classes that the reference snapshot does not hold (SURVEY 8c
    # "parity unpinned" (2)): generation only; the training block is the file's, less its machine-local checkpoint path
    "training": {"pretrained_path": "pretrained/lsun_bedroom_edm/edm_bedroom256_ema.pt", "value_ckpt": None, "n_iter": 100000,
                 "batchsize": 64, "sampling_batchsize": 100, "n_fid_samples": 5000, "seed": 42, "lr": 1e-8, "v_lr": 1e-5,
                 "beta_lr": 1e-6, "weight_decay": 0.0, "initial_log_loss_scale": 13, "log_every": 20, "fid_every": 100},
    "data": {"name": "lsun_bedroom"},
}


def get(name):
    from dxmi_config import Cfg
    return Cfg(copy.deepcopy(CONFIGS[name]))
